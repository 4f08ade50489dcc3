"""The pinning kit's harness runs end to end (tools/make_ref_inputs.py -> tests/test_reference_vectors.py): the
"reference" outputs here are STAND-INS written from the oracle itself, so this proves nothing about parity -- only that
the day real vectors arrive, the comparison code reads them, applies the stated tolerances and can fail (a corrupted
vector is caught)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(dirname):
    env = dict(os.environ, CRISPY_REF_VECTORS=str(dirname))
    return subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_reference_vectors.py"), "-q",
                           "-m", "not gpu", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, cwd=ROOT, timeout=600)


def test_kit_harness_end_to_end_with_stand_in_vectors(tmp_path, oracle):
    from crispy_amd.rnn_weights import load_rnnoise_nu_text
    from oracle import resample_oracle as RO
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_ref_inputs.py"), str(tmp_path)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    man = json.load(open(tmp_path / "manifest.json"))
    assert len(man["rnnoise"]) == 9 and man["resampler"]["samples"] == 48000 * 3 - 123
    for c in man["rnnoise"]:
        w = load_rnnoise_nu_text(str(tmp_path / c["model"]))
        x = np.fromfile(tmp_path / c["in"], dtype="<f4").reshape(-1, 480)
        assert x.shape[0] == c["frames"]
        out, vad = oracle.OracleDenoiseState(w).process(x)
        out.astype("<f4").tofile(tmp_path / c["ref_out"])
        vad.astype("<f4").tofile(tmp_path / c["ref_vad"])
    RO.resample_48k_to_16k(np.fromfile(tmp_path / man["resampler"]["in"], dtype="<f4")).astype("<f4").tofile(
        tmp_path / man["resampler"]["ref_out"])
    ok = _run(tmp_path)
    assert ok.returncode == 0 and "10 passed" in ok.stdout, ok.stdout[-1500:]
    # a vector that is off by 2e-4 of the peak must fail the 1e-4 bar
    c = man["rnnoise"][1]
    ref = np.fromfile(tmp_path / c["ref_out"], dtype="<f4")
    ref[1000] += 2e-4 * max(1.0, np.abs(ref).max())
    ref.tofile(tmp_path / c["ref_out"])
    bad = _run(tmp_path)
    assert bad.returncode != 0 and "1 failed" in bad.stdout, bad.stdout[-1500:]
