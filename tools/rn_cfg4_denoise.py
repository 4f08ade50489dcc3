"""Developer tool: the denoise stage of BASELINE cfg 4 alone -- 1024 streams x 3001 frames -- in both layouts (TBF: frame-major,
what bench.py's headline uses; BTF: stream-major, what the cfg 4 pipeline hands over) and both frame-kernel forms.
python tools/rn_cfg4_denoise.py  -> one line each: ms per call, ms of frame kernels inside it (hipEvent)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from crispy_amd import synthetic_weights, synth_audio
from crispy_amd.denoise import DenoiseState
w = synthetic_weights(0); dev = torch.device("cuda", 0)
B, T = 1024, 3001
x_tbf = synth_audio.batch_torch(B, T, dev, first_stream=0, seed=5)
x_btf = x_tbf.transpose(0, 1).contiguous()
for layout, x in (("tbf", x_tbf), ("btf", x_btf)):
    for waves in (1, 3):
        os.environ["CRISPY_RN_WAVES"] = str(waves)
        ds = DenoiseState(w, B, 0)
        y = torch.empty_like(x); torch.cuda.synchronize()
        for rep in range(2):
            ds.reset(); t0 = time.perf_counter(); ds.process_device(x.data_ptr(), y.data_ptr(), T, layout=layout); ds.synchronize(); dt = time.perf_counter() - t0
        ds.set_timing(True); ds.reset(); ds.process_device(x.data_ptr(), y.data_ptr(), T, layout=layout); ds.synchronize(); k = ds.last_kernel_ms(); ds.set_timing(False)
        print(f"layout {layout} waves {waves}: {dt * 1e3:.1f} ms per call, frame kernels {k[0]:.1f} ms", flush=True); ds.close()
