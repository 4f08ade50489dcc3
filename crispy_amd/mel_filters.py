"""Slaney-style mel filterbank used by Whisper (80 or 128 mels x 201 FFT bins at 16 kHz).

whisper.cpp reads these filters from the GGML model file; OpenAI ships
them as `mel_filters.npz` (librosa.filters.mel(sr=16000, n_fft=400, n_mels=80)).  No model file is
available here, so this module rebuilds the same matrix from its
definition: mel scale linear below 1 kHz (200/3 Hz per mel) and logarithmic above
(step log(6.4)/27), triangles on the FFT bin frequencies, each scaled by 2 / bandwidth."""
from __future__ import annotations

import numpy as np


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3.0)
    logstep = np.log(6.4) / 27.0
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / logstep, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    logstep = np.log(6.4) / 27.0
    return np.where(m >= 15.0, 1000.0 * np.exp(logstep * (m - 15.0)), m * (200.0 / 3.0))


def whisper_mel_filters(n_mels: int = 80, n_fft: int = 400, sr: int = 16000) -> np.ndarray:
    """[n_mels, n_fft//2+1] float32."""
    n_bins = n_fft // 2 + 1
    fft_freqs = np.linspace(0.0, sr / 2.0, n_bins)
    mel_pts = np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2)
    hz_pts = _mel_to_hz(mel_pts)
    fdiff = np.diff(hz_pts)
    ramps = hz_pts[:, None] - fft_freqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels]))[:, None]
    return w.astype(np.float32)
