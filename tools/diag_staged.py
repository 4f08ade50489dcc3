"""Diagnostic: the staged / fused pipelines against the oracle over 1000 frames, per library build.
python tools/diag_staged.py lib1.so lib2.so ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crispy_amd import _native as N, synthetic_weights, synth_audio as SA
from crispy_amd.denoise import DenoiseState
from tests import oracle_lib as O

B, T = 6, 1000
w = synthetic_weights(0)
x = SA.batch_np(B, T, first_stream=40) * np.float32(32768.0)
refs = [O.OracleDenoiseState(w).process(x[:, b])[0] for b in range(B)]
REPS = int(os.environ.get("DIAG_REPS", "2"))
MODES = (True,) if os.environ.get("DIAG_STAGED_ONLY") == "1" else (False, True)
for path in sys.argv[1:]:
    L = N.lib() if path == "default" else N.load_library(os.path.abspath(path))
    for staged in MODES:
        for rep in range(REPS):
            ds = DenoiseState(w, B, 0, lib=L)
            ds.set_pipeline(staged)
            out, vad = ds.process(x)
            msg = []
            for b in range(B):
                peak = max(np.abs(refs[b]).max(), 1.0)
                e = np.abs(out[:, b] - refs[b]).max(axis=1) / peak
                bad = np.flatnonzero(e > 1e-4)
                msg.append(f"{e.max():.1e}@{bad[0] if bad.size else -1}")
            print(os.path.basename(path), "staged" if staged else "fused ", rep, " ".join(msg), flush=True)
            ds.close()
