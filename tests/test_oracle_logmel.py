"""CPU tests of the log-mel oracle (oracle/logmel_oracle.c, whisper.cpp semantics) against golden
vectors produced by HuggingFace WhisperFeatureExtractor (tests/golden/make_logmel_golden.py) and
against an independent numpy formulation."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "logmel_golden.npz")


def test_mel_filter_bank_matches_hf_and_is_triangular():
    from crispy_amd.mel_filters import whisper_mel_filters
    G = np.load(GOLD)
    F = whisper_mel_filters(80)
    assert F.shape == (80, 201) and F.dtype == np.float32
    assert np.abs(F - G["filters"]).max() < 1e-8
    for m in range(80):
        nz = np.nonzero(F[m])[0]
        assert nz.size >= 1 and nz[-1] - nz[0] + 1 == nz.size and nz.size <= 64     # contiguous support
    assert whisper_mel_filters(128).shape == (128, 201)                                # large-v3 front end


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_matches_hf_golden(oracle, seed):
    from crispy_amd import synth_audio
    G = np.load(GOLD)
    n = int(G[f"clip{seed}/n"])
    x = synth_audio.clip16k_np(seed, n)
    mel = oracle.oracle_logmel(x, G["filters"])
    assert mel.shape == (80, 3000)
    tol = 1e-4 * np.abs(G[f"clip{seed}/mel_every7"]).max()     # north_star: log-mel within 1e-4 relative
    assert np.abs(mel[:, ::7] - G[f"clip{seed}/mel_every7"]).max() < tol
    assert np.abs(mel[:, :40] - G[f"clip{seed}/mel_head"]).max() < tol            # reflect padding at the start
    if n < 480000 - 400:
        assert np.abs(mel[:, -40:] - G[f"clip{seed}/mel_tail"]).max() < tol
    else:
        # full 30 s clip: whisper.cpp zero-pads where OpenAI/HF reflect-pad -> only the last frame differs
        d = np.abs(mel[:, -40:] - G[f"clip{seed}/mel_tail"]).max(axis=0)
        assert d[:-1].max() < tol and d[-1] > 10 * tol


def test_oracle_matches_numpy_stft(oracle):
    """Independent float64 formulation with numpy.fft."""
    from crispy_amd import synth_audio
    from crispy_amd.mel_filters import whisper_mel_filters
    F = whisper_mel_filters(80)
    n = 50000
    x = synth_audio.clip16k_np(5, n)
    xp = np.concatenate([x[1:201][::-1], x, np.zeros(480000 + 200, np.float32)]).astype(np.float64)
    hann = 0.5 * (1 - np.cos(2 * np.pi * np.arange(400) / 400))
    n_len = (xp.size - 400) // 160
    frames = np.lib.stride_tricks.sliding_window_view(xp, 400)[::160][:n_len]
    P = np.abs(np.fft.rfft(frames * hann, axis=1)) ** 2
    logmel = np.log10(np.maximum(P @ F.T.astype(np.float64), 1e-10)).T
    logmel = (np.maximum(logmel, logmel.max() - 8.0) + 4.0) / 4.0
    mel = oracle.oracle_logmel(x, F)
    assert np.abs(mel - logmel[:, :3000]).max() < 2e-5


def test_oracle_silence_and_argument_checks(oracle):
    from crispy_amd.mel_filters import whisper_mel_filters
    F = whisper_mel_filters(80)
    mel = oracle.oracle_logmel(np.zeros(16000, np.float32), F)
    assert np.all(mel == np.float32((-10.0 + 4.0) / 4.0))        # log10(1e-10) everywhere: max - 8 never clamps
    with pytest.raises(ValueError):
        oracle.oracle_logmel(np.zeros(480001, np.float32), F)
    with pytest.raises(ValueError):
        oracle.oracle_logmel(np.zeros(0, np.float32), F)
