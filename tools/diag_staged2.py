"""Diagnostic: staged vs fused pipeline on fresh handles, many repetitions; reports where they part."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crispy_amd import synthetic_weights, synth_audio as SA
from crispy_amd.denoise import DenoiseState
w = synthetic_weights(0)
REPS = int(os.environ.get("DIAG_REPS", "30"))
for (B, T) in ((20, 60), (6, 120), (64, 30)):
    x = SA.batch_np(B, T) * np.float32(32768.0)
    f = DenoiseState(w, B, 0); f.set_pipeline(False)
    of, vf = f.process(x)
    peak = np.abs(of).max()
    nbad = 0
    for rep in range(REPS):
        a = DenoiseState(w, B, 0); a.set_pipeline(True)
        oa, va = a.process(x)
        e = np.abs(oa - of).max(axis=2) / peak          # [T, B]
        if e.max() > 1e-5:
            nbad += 1
            tb = np.argwhere(e > 1e-5)
            first_t = tb[:, 0].min()
            streams = sorted(set(tb[:, 1].tolist()))
            print(f"B={B} T={T} rep {rep}: staged != fused; first frame {first_t}, streams {streams[:12]} ({len(streams)}), worst {e.max():.2e}; "
                  f"vad diff {np.abs(va - vf).max():.2e}; frames bad per stream {[int((e[:, s] > 1e-5).sum()) for s in streams[:6]]}", flush=True)
        a.close()
    print(f"B={B} T={T}: {nbad} of {REPS} repetitions differ", flush=True)
    f.close()
