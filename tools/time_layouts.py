"""Developer tool: one long call (B streams x T frames) in both layouts, device-resident."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crispy_amd import synthetic_weights, synth_audio
from crispy_amd.denoise import DenoiseState
B = int(os.environ.get("B", 1024)); T = int(os.environ.get("T", 3000))
dev = torch.device("cuda:0")
x = synth_audio.batch_torch(B, 100, dev)                  # [100, B, 480]
x = x.repeat(T // 100, 1, 1).contiguous()                 # [T, B, 480]
xb = x.permute(1, 0, 2).contiguous()                      # [B, T, 480]
for name, inp, lay in (("tbf", x, "tbf"), ("btf", xb, "btf")):
    ds = DenoiseState(synthetic_weights(0), B, 0)
    out = torch.empty_like(inp)
    ds.process_device(inp.data_ptr(), out.data_ptr(), T, layout=lay); ds.synchronize()
    ds.set_timing(True)
    t0 = time.perf_counter()
    ds.process_device(inp.data_ptr(), out.data_ptr(), T, layout=lay); ds.synchronize()
    dt = time.perf_counter() - t0
    fk, tot = ds.last_kernel_ms()
    print(f"{name}: B={B} T={T}: {dt*1e3:.1f} ms wall, frame kernels {fk:.1f} ms, enqueue-to-end {tot:.1f} ms -> {B*T/dt/1e6:.2f} M frames/s, {dt/T*1e6:.1f} us per frame")
    del ds, out
