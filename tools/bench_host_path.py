"""Developer tool: the host-pointer entry point crispy_rn_process (PCIe-inclusive, never bench.py's `value`):
4096 streams x T frames from pageable host memory, through the device, back to host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crispy_amd import synthetic_weights, synth_audio
from crispy_amd.denoise import DenoiseState
B = int(os.environ.get("B", 4096)); T = int(os.environ.get("T", 100))
ds = DenoiseState(synthetic_weights(0), B, 0)
x = synth_audio.batch_np(B, T) * np.float32(32768.0)
ds.process(x)
t0 = time.perf_counter()
for _ in range(3):
    out, vad = ds.process(x)
dt = (time.perf_counter() - t0) / 3
print(f"host path B={B} T={T}: {dt*1e3:.1f} ms per call -> {B*T/dt/1e6:.2f} M frames/s = {B*T/dt/100:.0f} streams, "
      f"{2*x.nbytes/dt/1e9:.1f} GB/s over PCIe (in + out)")
