import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from crispy_amd import synthetic_weights
from crispy_amd.denoise import DenoiseState
B, T = int(os.environ.get("B", 4096)), int(os.environ.get("T", 25))
ds = DenoiseState(synthetic_weights(0), B, 0)
dev = torch.device("cuda:0")
feat = torch.randn(T, B, 48, device=dev); sil = torch.zeros(T, B, dtype=torch.uint8, device=dev)
g1 = torch.zeros(T, B, 24, device=dev); g2 = torch.zeros(T, B, 24, device=dev); vad = torch.zeros(T, B, device=dev)
torch.cuda.synchronize()
for _ in range(3):
    ds.stage_rnn_device(feat.data_ptr(), sil.data_ptr(), g1.data_ptr(), g2.data_ptr(), T, vad.data_ptr())
ds.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ds.stage_rnn_device(feat.data_ptr(), sil.data_ptr(), g1.data_ptr(), g2.data_ptr(), T, vad.data_ptr())
ds.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"rn_rnn_kernel B={B} T={T}: {dt*1e3:.3f} ms per launch = {dt/T*1e6:.1f} us per frame of {B} streams; {B*T/dt/1e6:.1f} M stream-frames/s")
