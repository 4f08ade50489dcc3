"""Host-side mirror of the reference's ASR surface (transcribe-rs `SpeechModel::transcribe`,
reference call sites src-tauri/src/managers/transcription.rs:183-185), backed by the HIP library.

Round 1 covers the log-mel front end (`LogMel`); the encoder / decoder follow."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .mel_filters import whisper_mel_filters

N_FRAMES = 3000
CHUNK_SAMPLES = 480000  # 30 s at 16 kHz (commands/transcription.rs:175-176)


class LogMel:
    """whisper.cpp `log_mel_spectrogram` on the GPU: f32 PCM (16 kHz, +-1) -> [B, n_mel, 3000]."""

    def __init__(self, n_mel: int = 80, filters: np.ndarray | None = None, device: int = 0):
        f = whisper_mel_filters(n_mel) if filters is None else np.ascontiguousarray(filters, dtype=np.float32)
        if f.shape != (n_mel, 201):
            raise ValueError("filters must be [n_mel, 201]")
        self.n_mel = n_mel
        self._h = C.c_void_p()
        N.check(N.lib().crispy_mel_create(f.ctypes.data, n_mel, device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            N.lib().crispy_mel_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, clips) -> np.ndarray:
        """clips: a list of 1-D float32 arrays (each <= 480000 samples) or a 2-D array."""
        if isinstance(clips, np.ndarray) and clips.ndim == 2:
            lens = np.full(clips.shape[0], clips.shape[1], dtype=np.int32)
            pcm = np.ascontiguousarray(clips, dtype=np.float32)
        else:
            clips = [np.ascontiguousarray(c, dtype=np.float32).ravel() for c in clips]
            if not clips:
                return np.zeros((0, self.n_mel, N_FRAMES), np.float32)
            lens = np.array([c.size for c in clips], dtype=np.int32)
            pcm = np.zeros((len(clips), int(lens.max())), dtype=np.float32)
            for i, c in enumerate(clips):
                pcm[i, :c.size] = c
        out = np.empty((pcm.shape[0], self.n_mel, N_FRAMES), dtype=np.float32)
        N.check(N.lib().crispy_mel_compute(self._h, pcm.ctypes.data, pcm.shape[1], lens.ctypes.data,
                                           pcm.shape[0], out.ctypes.data))
        return out

    def compute_device(self, d_pcm: int, pcm_stride: int, n_samples: np.ndarray, d_out: int = 0, d_out_t: int = 0,
                       stream: int = 0):
        lens = np.ascontiguousarray(n_samples, dtype=np.int32)
        N.check(N.lib().crispy_mel_compute_device(self._h, d_pcm, pcm_stride, lens.ctypes.data, lens.size,
                                                  d_out or None, d_out_t or None, stream or None))

    def synchronize(self):
        N.check(N.lib().crispy_mel_synchronize(self._h))
