"""resample_oracle.py -- TEST INFRASTRUCTURE ONLY (parity oracle).

numpy restatement of the 48 kHz -> 16 kHz resampling step between the two halves of the hot path:
  rubato 0.16.2 `FftFixedIn::<f32>::new(sr_in, 16000, 1024, 1, 1)` as driven by
  src-tauri/src/commands/transcription.rs:198-208 (construction), :314-322 (1024-sample chunks),
  :347-357 (last chunk zero-padded to 1024), plus the s16 WAV hand-off that precedes it in the product
  (recording.rs:101-118 quantise `(x.clamp(-1,1)*32767) as i16`; commands/transcription.rs:306-313 read `/32768`).

[UPSTREAM-RECALL] rubato's source is not vendored (Cargo.lock:4166-4175); restated from the published
algorithm: synchronous FFT resampler, fft_size_in = ceil(1024/3)*3 = 1026, fft_size_out = 342; per block
zero-pad to 2*1026, forward real FFT, multiply by the spectrum of a BlackmanHarris^2-windowed sinc
(cutoff 0.4^(16/1026) * 342/1026, scaled by 1/(2*1026)), keep the first 342 bins, inverse real FFT of
size 684, overlap-add the second half into the next block.  Input left over after the last full block
is never flushed (as upstream).  PARITY UNPINNED (no rubato here, no reference fixture)."""
from __future__ import annotations

import numpy as np

FFT_IN, FFT_OUT, CHUNK = 1026, 342, 1024


def blackman_harris2(n: int) -> np.ndarray:
    x = np.arange(n) / n
    w = 0.35875 - 0.48829 * np.cos(2 * np.pi * x) + 0.14128 * np.cos(4 * np.pi * x) - 0.01168 * np.cos(6 * np.pi * x)
    return w * w


def filter_spectrum() -> np.ndarray:
    cutoff = (0.4 ** (16.0 / FFT_IN)) * FFT_OUT / FFT_IN
    x = np.arange(FFT_IN) - FFT_IN // 2
    y = blackman_harris2(FFT_IN) * np.sinc(x * cutoff)
    y = y / y.sum()
    ft = np.zeros(2 * FFT_IN)
    ft[:FFT_IN] = y / (2 * FFT_IN)
    return np.fft.rfft(ft)


def wav_s16_roundtrip(x: np.ndarray) -> np.ndarray:
    """f32 in +-1 -> s16 as WavWriter stores it (truncation toward zero) -> f32 as run_transcription reads it."""
    q = np.trunc(np.clip(x.astype(np.float32), -1.0, 1.0) * np.float32(32767.0)).astype(np.int16)
    return q.astype(np.float32) / np.float32(32768.0)


def resample_48k_to_16k(x: np.ndarray) -> np.ndarray:
    """One stream: 48 kHz f32 -> 16 kHz f32 exactly as the reference's chunk loop would produce it."""
    x = np.asarray(x, dtype=np.float64)
    n_pad = -(-x.size // CHUNK) * CHUNK          # last chunk zero-padded to 1024
    xp = np.zeros(n_pad)
    xp[:x.size] = x
    n_blk = n_pad // FFT_IN                      # the tail shorter than one FFT block stays in the buffer
    F = filter_spectrum()
    out = np.zeros(n_blk * FFT_OUT)
    overlap = np.zeros(FFT_OUT)
    for b in range(n_blk):
        buf = np.zeros(2 * FFT_IN)
        buf[:FFT_IN] = xp[b * FFT_IN:(b + 1) * FFT_IN]
        spec = np.fft.rfft(buf) * F
        o = np.zeros(FFT_OUT + 1, dtype=complex)
        o[:FFT_OUT] = spec[:FFT_OUT]
        y = np.fft.irfft(o, 2 * FFT_OUT) * (2 * FFT_OUT)      # realfft's inverse is unnormalised
        out[b * FFT_OUT:(b + 1) * FFT_OUT] = y[:FFT_OUT] + overlap
        overlap = y[FFT_OUT:]
    return out.astype(np.float32)
