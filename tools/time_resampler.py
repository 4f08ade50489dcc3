"""Developer tool (GPU box): the 48 -> 16 kHz resampler on its own -- B streams x SEC seconds, ms per call and the error
against the oracle on two streams.  B=1024 SEC=30."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crispy_amd.pipeline import Resampler48to16
from oracle import resample_oracle as RO
B = int(os.environ.get("B", 1024)); SEC = float(os.environ.get("SEC", 30))
n = int(48000 * SEC)
rs = Resampler48to16()
n16 = rs.out_len(n)
g = torch.Generator(device="cuda").manual_seed(1)
x = (torch.rand(B, n, device="cuda", generator=g) * 2 - 1) * 0.7
y = torch.zeros(B, n16, device="cuda")
torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    rs.process_device(x.data_ptr(), n, n, B, y.data_ptr(), n16)
    rs.synchronize()
    print(f"{B} streams x {SEC} s: call {rep}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
for b in (0, B - 1):
    ref = RO.resample_48k_to_16k(x[b].cpu().numpy())
    print(f"stream {b}: max |err| {np.abs(y[b].cpu().numpy() - ref).max():.3e}  (max |ref| {np.abs(ref).max():.3f})")
