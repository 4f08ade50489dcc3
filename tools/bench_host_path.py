"""Developer tool: the host-pointer entry point crispy_rn_process (PCIe-inclusive, never bench.py's `value`):
4096 streams x T frames from host memory, through the device, back to host -- from pageable arrays and from
arrays registered with crispy_host_register, in both layouts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crispy_amd import synthetic_weights, synth_audio
from crispy_amd.denoise import DenoiseState
B = int(os.environ.get("B", 4096)); T = int(os.environ.get("T", 100))
ds = DenoiseState(synthetic_weights(0), B, 0)
x = synth_audio.batch_np(B, T) * np.float32(32768.0)
out = np.empty_like(x); vad = np.empty((T, B), np.float32)
def run(tag, x, out, vad, layout):
    ds.process_into(x, out, vad, layout)
    t0 = time.perf_counter()
    for _ in range(3):
        ds.process_into(x, out, vad, layout)
    dt = (time.perf_counter() - t0) / 3
    print(f"host path {tag:22s} B={B} T={T}: {dt*1e3:6.1f} ms per call -> {B*T/dt/1e6:6.2f} M frames/s = {B*T/dt/100:7.0f} streams, "
          f"{2*x.nbytes/dt/1e9:5.1f} GB/s over PCIe (in + out)", flush=True)
run("pageable tbf", x, out, vad, "tbf")
xb = np.ascontiguousarray(x.transpose(1, 0, 2)); ob = np.empty_like(xb)
run("pageable btf", xb, ob, vad, "btf")
for a in (x, out, vad): DenoiseState.register_host(a)
run("registered tbf", x, out, vad, "tbf")
for a in (x, out, vad): DenoiseState.unregister_host(a)
