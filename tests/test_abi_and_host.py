"""CPU tests of the boundary: the shared library loads and exports every symbol the header
declares, fails loudly without a GPU, and the host-side mirrors of the reference adapter
(LinearResampler: src-tauri/src/audio.rs:73-134, tests audio.rs:1040-1096) behave like the reference's."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "crispy_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(crispy_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from crispy_amd import _native as N
    L = N.lib()
    syms = header_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(L, s), f"libcrispy_hip.so does not export {s}"
    assert set(N.ALL_SYMBOLS) == set(syms), "crispy_amd._native.ALL_SYMBOLS out of sync with the header"
    assert b"gfx950" in L.crispy_version()


def test_argument_validation_without_touching_a_device():
    from crispy_amd import _native as N
    L = N.lib()
    h = C.c_void_p()
    w = np.zeros(100, np.int8)
    assert L.crispy_rn_create(w.ctypes.data, w.size, 1, 0, C.byref(h)) == -5   # BAD_MODEL
    assert b"87503" in L.crispy_last_error()
    assert L.crispy_rn_create(None, 0, 1, 0, C.byref(h)) == -5
    w = np.zeros(N.RN_WEIGHT_BYTES, np.int8)
    assert L.crispy_rn_create(w.ctypes.data, w.size, 0, 0, C.byref(h)) == -1   # INVALID_ARG
    assert L.crispy_rn_reset(None, 0) == -1
    assert L.crispy_rn_process(None, None, None, None, 1, 0) == -1
    assert L.crispy_rn_n_streams(None) == 0


def test_no_cpu_fallback():
    """On a box without a gfx950 device the product path must fail loudly, never compute on the CPU."""
    import torch
    from crispy_amd import _native as N, synthetic_weights
    from crispy_amd.denoise import DenoiseState
    if N.lib().crispy_device_count() > 0:
        pytest.skip("a GPU is present")
    assert not torch.cuda.is_available()
    with pytest.raises(N.CrispyError) as e:
        DenoiseState(synthetic_weights(0), 4, 0)
    assert e.value.code == -2 and "no CPU path" in str(e.value)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "crispy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "liboracle" not in txt and "rnnoise_oracle" not in txt, f
    import subprocess
    out = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(pkg, "libcrispy_hip.so")],
                         capture_output=True, text=True).stdout
    assert "rno_" not in out


# ---- weights ---------------------------------------------------------------------------------
def test_weight_blob_layout_and_text_round_trip(tmp_path):
    from crispy_amd import rnn_weights as RW
    offs, total = RW.blob_offsets()
    assert total == 87503 == RW.BLOB_BYTES
    assert offs["denoise_gru"]["U"][1] == 96 * 288 and offs["vad_output"]["b"] == (1032 + 3528 + 24, 1)
    w = RW.synthetic_weights(3)
    p = tmp_path / "model.txt"
    RW.save_rnnoise_nu_text(str(p), w)
    assert np.array_equal(RW.load_rnnoise_nu_text(str(p)), w)
    bad = tmp_path / "bad.txt"
    bad.write_text("not a model\n1 2 3\n")
    with pytest.raises(ValueError):
        RW.load_rnnoise_nu_text(str(bad))
    trunc = tmp_path / "trunc.txt"
    trunc.write_text(open(p).read()[:2000])
    with pytest.raises(ValueError):
        RW.load_rnnoise_nu_text(str(trunc))


def test_synthetic_weights_are_seeded_and_int8():
    from crispy_amd import synthetic_weights
    a, b, c = synthetic_weights(0), synthetic_weights(0), synthetic_weights(1)
    assert a.dtype == np.int8 and np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.min() >= -127 and a.max() <= 127


# ---- LinearResampler: the reference's own tests (audio.rs:1040-1096) ---------------------------
def _run(rs, n, val=lambda i: 0.5):
    out = []
    for i in range(n):
        rs.process_sample(val(i), out.append)
    return out


def test_resampler_passthrough_same_rate():
    from crispy_amd.denoise import LinearResampler
    out = _run(LinearResampler(48000.0, 48000.0), 100, lambda i: i / 100)
    assert len(out) == 100 and out[10] == pytest.approx(0.1)


def test_resampler_passthrough_within_1hz():
    from crispy_amd.denoise import LinearResampler
    assert len(_run(LinearResampler(48000.0, 48000.5), 50)) == 50


def test_resampler_downsample_3_to_1():
    from crispy_amd.denoise import LinearResampler
    n = len(_run(LinearResampler(48000.0, 16000.0), 300))
    assert 80 <= n <= 120


def test_resampler_upsample_1_to_3():
    from crispy_amd.denoise import LinearResampler
    n = len(_run(LinearResampler(16000.0, 48000.0), 100))
    assert 250 <= n <= 350


def test_resampler_rates_and_reset():
    from crispy_amd.denoise import LinearResampler
    rs = LinearResampler(44100.0, 48000.0)
    assert rs.rates() == (44100.0, 48000.0)
    _run(rs, 10)
    rs.set_rates(48000.0, 16000.0)
    assert rs.rates() == (48000.0, 16000.0) and not rs.has_last


def test_resampler_interpolates_linearly():
    from crispy_amd.denoise import LinearResampler
    out = _run(LinearResampler(24000.0, 48000.0), 50, lambda i: float(i))
    d = np.diff(out[2:])
    assert np.allclose(d, 0.5, atol=1e-5)


def test_capture_buffers_passthrough_ring_and_level():
    """push_mono_to_buffers without a noise suppressor (audio.rs:682-730): the raw sample goes through the recording
    resampler, the ring holds at most 10 s at 48 kHz, the level meter accumulates mono^2."""
    from crispy_amd.denoise import CaptureBuffers, REC_SAMPLE_RATE
    cb = CaptureBuffers()
    x = (np.arange(96, dtype=np.float32) % 7 - 3) / 8
    for s in x:
        cb.push_mono(s, None, 16000.0)                # a 16 kHz device: three recorded samples per input sample
    assert cb.rec_resampler.rates() == (16000.0, 48000.0)
    assert len(cb.rec_buffer) in (3 * 96, 3 * 96 - 1, 3 * 96 - 2, 3 * 96 - 3)   # linear interpolation lags one sample
    assert abs(cb.rms() - float(np.sqrt(np.mean(x.astype(np.float64) ** 2)))) < 1e-6
    cb.reset_level()
    assert cb.rms() == 0.0
    cb2 = CaptureBuffers()
    cb2.max_len = 100                                   # the reference's cap is SAMPLE_RATE * 10; same eviction rule
    for i in range(250):
        cb2.push_mono(i / 250.0, None, float(REC_SAMPLE_RATE))
    assert len(cb2.rec_buffer) == 100 and abs(cb2.rec_buffer[-1] - 249 / 250.0) < 1e-6 and abs(cb2.rec_buffer[0] - 150 / 250.0) < 1e-6
    assert CaptureBuffers().max_len == 480000
