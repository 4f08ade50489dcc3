"""CPU tests of the resampler oracle (oracle/resample_oracle.py): rubato FftFixedIn-style FFT resampler."""
import numpy as np


def test_unity_gain_delay_and_alias_rejection():
    from oracle import resample_oracle as R
    n = 48000
    t = np.arange(n) / 48000.0
    y = R.resample_48k_to_16k(np.sin(2 * np.pi * 1000 * t).astype(np.float32))
    assert y.size == (-(-n // 1024) * 1024) // 1026 * 342
    tt = np.arange(2000, 6000) / 16000.0
    A = np.stack([np.sin(2 * np.pi * 1000 * tt), np.cos(2 * np.pi * 1000 * tt)], 1)
    c, *_ = np.linalg.lstsq(A, y[2000:6000], rcond=None)
    assert abs(np.hypot(*c) - 1.0) < 1e-4                      # unity pass-band gain
    delay = (-np.arctan2(c[1], c[0]) / (2 * np.pi * 1000) * 16000) % 16
    assert abs(delay - (171 % 16)) < 0.05                       # group delay = 513 input = 171 output samples
    y2 = R.resample_48k_to_16k(np.sin(2 * np.pi * 10000 * t).astype(np.float32))
    assert np.sqrt((y2[2000:6000] ** 2).mean()) < 1e-3           # 10 kHz is above the new Nyquist


def test_fft_block_equals_circulant_operator():
    """The GPU applies each block as a GEMM with g[(3n - j) mod 2052]; same operator as the FFT route."""
    from oracle import resample_oracle as R
    F = R.filter_spectrum()
    k = np.arange(1, 342)
    m = np.arange(2052)
    g = F[0].real + 2 * np.real((F[1:342][None, :] * np.exp(2j * np.pi * np.outer(m, k) / 2052)).sum(1))
    blk = np.random.default_rng(0).standard_normal(1026)
    M = g[(3 * np.arange(684)[:, None] - np.arange(1026)[None, :]) % 2052]
    buf = np.zeros(2052)
    buf[:1026] = blk
    o = np.zeros(343, complex)
    o[:342] = (np.fft.rfft(buf) * F)[:342]
    assert np.abs(M @ blk - np.fft.irfft(o, 684) * 684).max() < 1e-12


def test_wav_s16_roundtrip_matches_reference_writer_tests():
    """recording.rs:483-504: clamp then x32767 truncated toward zero."""
    from oracle import resample_oracle as R
    x = np.array([0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5, 1e-5], np.float32)
    q = (R.wav_s16_roundtrip(x) * 32768).astype(np.int32)
    assert q.tolist() == [0, 32767, -32767, 32767, -32767, 16383, -16383, 0]
