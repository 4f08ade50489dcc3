"""Generates tests/golden/rnnoise_golden.npz from the C oracle (oracle/rnnoise_oracle.c).

The reference's own arithmetic (nnnoiseless 0.5.2) cannot be run in this environment and the
reference holds no fixture for this path (parity unpinned), so these vectors pin
the ORACLE (regression) and give the GPU tests fixed expected outputs that travel to the GPU box.

    python tests/golden/make_rnnoise_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from crispy_amd import synthetic_weights, synth_audio  # noqa: E402
from tests import oracle_lib as O  # noqa: E402

T = 20
cases = {}
for seed in (0, 1, 2):
    w = synthetic_weights(seed)
    x = (synth_audio.stream_np(100 + seed, T, silent=False) * np.float32(32768.0)).reshape(T, 480)
    out, vad, taps = O.OracleDenoiseState(w).process(x, with_taps=True)
    cases[f"seed{seed}"] = (w, x, out, vad, taps)
# silence and pure tone with seed-0 weights
w = synthetic_weights(0)
x = np.zeros((T, 480), np.float32)
cases["silence"] = (w, x) + O.OracleDenoiseState(w).process(x, with_taps=True)
t = np.arange(T * 480) / 48000.0
x = (0.25 * np.sin(2 * np.pi * 440.0 * t) * 32768.0).astype(np.float32).reshape(T, 480)
cases["tone440"] = (w, x) + O.OracleDenoiseState(w).process(x, with_taps=True)
# near-silence: below the E < 0.04 threshold, pure pass-through of the high-passed input
x = (1e-3 * np.sin(2 * np.pi * 300.0 * t)).astype(np.float32).reshape(T, 480)
cases["whisper_quiet"] = (w, x) + O.OracleDenoiseState(w).process(x, with_taps=True)

blob = {}
for k, (w, x, out, vad, taps) in cases.items():
    blob[f"{k}/x"] = x
    blob[f"{k}/out"] = out
    blob[f"{k}/vad"] = vad
    blob[f"{k}/taps"] = taps
for seed in (0, 1, 2):
    blob[f"weights{seed}"] = synthetic_weights(seed)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rnnoise_golden.npz"), **blob)
print("wrote", len(blob), "arrays")
