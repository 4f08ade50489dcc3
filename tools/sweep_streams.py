"""Developer tool: frame-kernel time vs number of streams (occupancy / tail behaviour)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crispy_amd import synthetic_weights, synth_audio
from crispy_amd.denoise import DenoiseState
T = int(os.environ.get("T", 50))
w = synthetic_weights(0)
for B in [int(v) for v in os.environ.get("BS", "256,1024,2048,2304,4096,4608,8192,16384").split(",")]:
    ds = DenoiseState(w, B, 0)
    x = synth_audio.batch_torch(B, T, torch.device("cuda:0"))
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    ds.process_device(x.data_ptr(), y.data_ptr(), T); ds.synchronize()
    ds.set_timing(True)
    ds.process_device(x.data_ptr(), y.data_ptr(), T); ds.synchronize()
    fk, tot = ds.last_kernel_ms()
    print(f"B={B:6d} T={T} frame_kernel={fk:8.3f} ms total={tot:8.3f} ms  -> {B*T/(tot*1e-3)/1e6:7.2f} Mframes/s  per-frame-per-wave-round={fk/T*1e3:7.1f} us")
    del ds, x, y
