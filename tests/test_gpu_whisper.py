"""GPU parity tests of the Whisper path (crispy_asr_* behind the C ABI) against the float64 numpy oracle
and the HuggingFace golden vectors (seeded random-init Whisper-tiny: no real weights exist here).

Tolerance: the GPU computes in f32 on the f32-input matrix cores; against the float64 oracle the
encoder output must agree to 1e-4 of its peak (observed ~1e-6), greedy token ids must be identical."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "whisper_tiny_golden.npz")


@pytest.fixture(scope="module")
def tiny():
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    hp = HParams.tiny()
    return hp, synthetic_whisper_weights(hp, 0)


@pytest.fixture(scope="module")
def model(tiny):
    from crispy_amd.asr import WhisperModel
    hp, W = tiny
    return WhisperModel(hp, W)


def test_encoder_matches_hf_golden_and_oracle(tiny, model, oracle):
    from crispy_amd import synth_audio
    from crispy_amd.mel_filters import whisper_mel_filters
    from oracle import whisper_oracle as WO
    hp, W = tiny
    G = np.load(GOLD)
    x = synth_audio.clip16k_np(0, 464000)
    enc = model.encode([x])[0]
    assert enc.shape == (1500, 384) and np.isfinite(enc).all()
    ref_rows = G["enc_rows"]
    assert np.abs(enc[::25] - ref_rows).max() <= 1e-4 * np.abs(ref_rows).max()
    ref = WO.encoder_forward(W, hp, oracle.oracle_logmel(x, whisper_mel_filters(80)))
    err = np.abs(enc - ref).max() / np.abs(ref).max()
    assert err <= 1e-4, err


def test_encoder_batch_ragged_and_independent(model):
    """Clips of different length in one batch; every clip equals its solo run bit for bit."""
    from crispy_amd import synth_audio
    clips = [synth_audio.clip16k_np(50 + i, n) for i, n in enumerate((480000, 160000, 31234, 480000, 8000))]
    enc = model.encode(clips)
    assert enc.shape == (5, 1500, 384) and np.isfinite(enc).all()
    for i in (1, 4):
        solo = model.encode([clips[i]])[0]
        assert np.array_equal(solo, enc[i])


def test_asr_container_errors(tiny):
    import ctypes as C
    from crispy_amd import _native as N
    from crispy_amd.asr import WhisperModel
    from crispy_amd.whisper_weights import HParams
    hp, W = tiny
    bad = dict(W)
    del bad["encoder.ln_post.bias"]
    with pytest.raises(KeyError):
        WhisperModel(hp, bad)
    bad = dict(W)
    bad["encoder.conv1.bias"] = np.zeros(7, np.float32)
    with pytest.raises(ValueError):
        WhisperModel(hp, bad)
    h = C.c_void_p()
    hpa = (C.c_int * 10)(*HParams(n_audio_state=100).as_ints())
    f = np.zeros((80, 201), np.float32)
    assert N.lib().crispy_asr_create(hpa, f.ctypes.data, 0, C.byref(h)) == -5
