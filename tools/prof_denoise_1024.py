"""Developer tool (GPU box): the denoise stage of cfg 4 on its own -- 1024 streams x 3001 frames through crispy_rn_process_device,
for `rocprofv3 --kernel-trace`: which of the two chains (the high-pass on the helper stream, the three-wave frame kernel) the
call waits for.  STREAMS=1024 FRAMES=3001."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crispy_amd import synthetic_weights, synth_audio
from crispy_amd.denoise import DenoiseState
B, T = int(os.environ.get("STREAMS", 1024)), int(os.environ.get("FRAMES", 3001))
ds = DenoiseState(synthetic_weights(0), B, 0)
x = synth_audio.batch_torch(B, 250, torch.device("cuda:0")).repeat(13, 1, 1)[:T].contiguous()
y = torch.empty_like(x)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    ds.process_device(x.data_ptr(), y.data_ptr(), T); ds.synchronize()
    print(f"{B} streams x {T} frames: {(time.perf_counter() - t0) * 1e3:.1f} ms = {(time.perf_counter() - t0) * 1e6 / T:.1f} us per frame", flush=True)
