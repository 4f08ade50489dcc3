"""Generates tests/golden/rnnoise_extreme_golden.npz from the C oracle: the weight-extreme cases of
crispy_amd.rnn_weights.extreme_weights (all +127 / -127 / alternating / zero / saturated biases / heavy-tailed /
saturating rows), 12 frames of the seed-100 tone+noise stream and of a loud full-scale stream each.

Like rnnoise_golden.npz these pin the ORACLE (parity unpinned against nnnoiseless itself) and give the
GPU tests fixed expected outputs; the blobs are regenerated from their kind by the tests, not stored.

    python tests/golden/make_rnnoise_extreme_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from crispy_amd import rnn_weights as RW, synth_audio  # noqa: E402
from tests import oracle_lib as O  # noqa: E402

T = 12


def inputs():
    a = (synth_audio.stream_np(100, T, silent=False) * np.float32(32768.0)).reshape(T, 480)
    rng = np.random.default_rng(5)
    t = np.arange(T * 480) / 48000.0
    b = (30000.0 * np.sign(np.sin(2 * np.pi * 211.0 * t)) + 2000.0 * rng.standard_normal(T * 480)).astype(np.float32)
    return {"tone": a, "loud": b.reshape(T, 480)}


if __name__ == "__main__":
    blob = {}
    for name, x in inputs().items():
        blob[f"x/{name}"] = x
        for kind in RW.EXTREME_KINDS:
            out, vad, taps = O.OracleDenoiseState(RW.extreme_weights(kind)).process(x, with_taps=True)
            assert np.isfinite(out).all()
            blob[f"{kind}/{name}/out"] = out
            blob[f"{kind}/{name}/vad"] = vad
            blob[f"{kind}/{name}/gains"] = taps[:, 42:64].copy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rnnoise_extreme_golden.npz"), **blob)
    print("wrote", len(blob), "arrays")
