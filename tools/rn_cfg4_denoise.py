import os, sys, time, numpy as np, torch
sys.path.insert(0, ".")
from crispy_amd import synthetic_weights, synth_audio
from crispy_amd.denoise import DenoiseState
w = synthetic_weights(0); dev = torch.device("cuda", 0)
B, T = 1024, 3001
x = synth_audio.batch_torch(B, T, dev, first_stream=0, seed=5).transpose(0, 1).contiguous()
for waves in (1, 3):
    os.environ["CRISPY_RN_WAVES"] = str(waves)
    ds = DenoiseState(w, B, 0)
    y = torch.empty_like(x); torch.cuda.synchronize()
    for rep in range(2):
        ds.reset(); t0 = time.perf_counter(); ds.process_device(x.data_ptr(), y.data_ptr(), T, layout="btf"); ds.synchronize(); dt = time.perf_counter() - t0
    ds.set_timing(True); ds.reset(); ds.process_device(x.data_ptr(), y.data_ptr(), T, layout="btf"); ds.synchronize(); k = ds.last_kernel_ms(); ds.set_timing(False)
    print(os.environ.get("CRISPY_RN_HP_SPLIT"), "waves", waves, "ms", round(dt*1e3, 1), "frame kernels", round(k[0], 1), flush=True); ds.close()
