"""The header itself, not ctypes prototypes: a C99 program compiled against include/crispy_hip.h drives the
single-stream `process_frame` drop-in (one 480-sample frame per call from host slices, audio.rs:260-268) and its
output is compared with the oracle; the same program reports the per-call latency bench.py publishes."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_compiles_against_the_header_as_c99(tmp_path):
    """CPU: the program builds with -std=c99 -pedantic -Werror and, without a device, fails loudly through the ABI's
    own status/message path (no CPU fallback)."""
    from tests import c_dropin
    exe = c_dropin.build(str(tmp_path))
    from crispy_amd import _native as N
    if N.lib().crispy_device_count() > 0:
        pytest.skip("a GPU is present: the device path is covered by the gpu test below")
    from crispy_amd import rnn_weights as RW
    model = tmp_path / "m.txt"
    RW.save_rnnoise_nu_text(str(model), RW.synthetic_weights(0))
    (tmp_path / "in.f32").write_bytes(np.zeros(480, np.float32).tobytes())
    r = subprocess.run([exe, str(model), str(tmp_path / "in.f32"), str(tmp_path / "out.f32"), "1", "0"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_c_dropin_matches_oracle_and_reports_latency(oracle, weights0):
    from crispy_amd import synth_audio as SA
    from tests import c_dropin
    T = 40
    x = (SA.stream_np(11, T, silent=False) * np.float32(32768.0)).reshape(T, 480)
    out, vad, lat = c_dropin.run(weights0, x, timed_calls=300)
    ro, rv = oracle.OracleDenoiseState(weights0).process(x)
    assert np.abs(out - ro).max() <= 1e-4 * np.abs(ro).max() + 1e-3
    assert np.abs(vad - rv).max() < 1e-4
    assert lat["calls"] == 300 and 0 < lat["p50"] <= lat["p99"] <= lat["max"]
    assert lat["p99"] < 10000, f"a process_frame call must fit the 10 ms audio callback budget: {lat}"
