"""Developer tool: where one BASELINE cfg 2 step (4096 streams x 100 frames, device-resident) spends its time -- start of
every sub-chunk's frame kernel relative to the end of the previous one (CRISPY_RN_TIMELINE=1 makes
crispy_rn_last_kernel_ms print it).  Steps are enqueued back to back, as bench.py does."""
import os, sys
os.environ["CRISPY_RN_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crispy_amd import synthetic_weights, synth_audio
from crispy_amd.denoise import DenoiseState
B, T = int(os.environ.get("B", 4096)), int(os.environ.get("T", 100))
ds = DenoiseState(synthetic_weights(0), B, 0)
x = synth_audio.batch_torch(B, T, torch.device("cuda:0")); y = torch.empty_like(x)
torch.cuda.synchronize()
for _ in range(3):
    ds.process_device(x.data_ptr(), y.data_ptr(), T)
ds.synchronize()
ds.set_timing(True)
for _ in range(3):                       # three steps in flight; the timeline of the LAST one is kept
    ds.process_device(x.data_ptr(), y.data_ptr(), T)
ds.synchronize()
print("frame-kernel sum, step (ms):", ds.last_kernel_ms())
