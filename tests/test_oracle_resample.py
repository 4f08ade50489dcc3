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


def test_time_domain_polyphase_derivation_agrees_with_the_fft_route():
    """Independent second derivation of the operator (VERDICT r1 #2e).  The FFT route -- zero-pad a 1026-sample
    block to 2052, multiply by the filter's spectrum, keep 342 bins, inverse FFT of 684, overlap-add -- is a linear
    convolution with the 1026-tap BlackmanHarris^2-windowed sinc followed by an ideal (circular) band limit and
    decimation by 3.  Derived again in the time domain, with no FFT and no blocks: y[n] = sum_j h[3n - j] x[j] over the
    whole zero-padded signal, i.e. a direct-form FIR decimator.  The two differ only by what aliases in the FIR route
    and is cut off in the FFT route.  For input below the transition band (faded tones under 7 kHz) that is only the
    broadband edge transient each 1026-sample block acquires by being cut out of the signal: the FFT route band-limits
    every block on its own before the overlap-add, the FIR route sums the blocks first (where the edges cancel) --
    measured 2-5e-5 of the signal, largest exactly at block boundaries (n = 342 k).  Agreement at 1e-4 therefore pins the
    block alignment, the overlap-add, the gain (1/2052 x 684 x 3 = 1) and the group delay of the restatement.  For
    white noise the difference is the filter's transition band around 8 kHz (cutoff 7.89 kHz, keep-band edge 8 kHz),
    a few 1e-4 of the signal."""
    from oracle import resample_oracle as R
    rng = np.random.default_rng(4)
    n = 48000 + 777                                   # not a multiple of 1024: exercises the zero-padded last chunk
    t = np.arange(n) / 48000.0
    x = (0.4 * np.sin(2 * np.pi * 700.0 * t) + 0.2 * np.sin(2 * np.pi * 3100.0 * t + 1.0)
         + 0.1 * np.sin(2 * np.pi * 6900.0 * t + 2.0))
    ramp = 0.5 - 0.5 * np.cos(np.pi * np.minimum(1.0, np.minimum(np.arange(n), n - 1 - np.arange(n)) / 4800.0))
    x = (x * ramp).astype(np.float32)                # 100 ms fades: an abrupt edge is broadband, not "below 7 kHz"
    y_fft = R.resample_48k_to_16k(x).astype(np.float64)
    # the filter, restated from its definition (not through filter_spectrum())
    cutoff = (0.4 ** (16.0 / 1026)) * 342 / 1026
    k = np.arange(1026) - 513
    arg = np.arange(1026) / 1026.0
    bh = 0.35875 - 0.48829 * np.cos(2 * np.pi * arg) + 0.14128 * np.cos(4 * np.pi * arg) - 0.01168 * np.cos(6 * np.pi * arg)
    h = bh * bh * np.sinc(k * cutoff)
    h /= h.sum()
    n_pad = -(-n // 1024) * 1024
    xp = np.zeros(n_pad)
    xp[:n] = x
    n_out = (n_pad // 1026) * 342
    full = np.convolve(xp[:(n_pad // 1026) * 1026], h)            # only whole 1026-blocks are ever transformed
    y_td = full[::3][:n_out]
    assert y_fft.size == n_out == R.resample_48k_to_16k(x).size
    e = np.abs(y_fft - y_td)
    assert e.max() < 1e-4 * np.abs(y_td).max(), e.max()
    assert int(e.argmax()) % 342 in (0, 1, 341)                   # ... and the residual is the block-edge effect
    # white noise: the transition-band energy the FIR route aliases and the FFT route removes
    w = rng.standard_normal(1026 * 8).astype(np.float32)
    a = R.resample_48k_to_16k(w).astype(np.float64)
    b = np.convolve(w.astype(np.float64), h)[::3][:a.size]
    rel = np.abs(a - b).max() / np.abs(b).max()
    assert 1e-6 < rel < 1e-2, rel


def test_out_len_formula_matches_the_chunk_loop():
    """crispy_resampler_out_len: floor(ceil(n / 1024) * 1024 / 1026) * 342 -- what the reference's loop of 1024-sample
    process() calls (last one zero-padded, commands/transcription.rs:314-357) hands to the chunker when FftFixedIn
    buffers its input in 1026-sample FFT blocks [UPSTREAM-RECALL rubato 0.16.2 synchro.rs: input_buffers + saved_frames]."""
    from crispy_amd import _native as N
    from oracle import resample_oracle as R
    for n in (1, 1023, 1024, 1025, 1026, 2052, 48000, 1_440_000, 1_440_000 + 5):
        want = (-(-n // 1024) * 1024) // 1026 * 342
        assert N.lib().crispy_resampler_out_len(n) == want
        if n <= 48000:
            assert R.resample_48k_to_16k(np.zeros(n, np.float32)).size == want
