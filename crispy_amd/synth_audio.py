"""Seeded synthetic 48 kHz mono audio in the shape of BASELINE cfg 1 / cfg 2.

Streams are a voiced tone (f0 in U[80,400] Hz with decaying harmonics, slow amplitude envelope)
plus white noise at an SNR in U[0,20] dB; every tenth stream is digital silence so that the
`E < 0.04` branch of the frame algorithm is exercised.  Samples are f32 in +-1; multiply by
32768 before `process_frame` as the reference adapter does (audio.rs:264)."""
from __future__ import annotations

import numpy as np

SR = 48000
FRAME = 480


def stream_np(stream_id: int, n_frames: int, silent: bool | None = None) -> np.ndarray:
    """One stream, [n_frames*480] f32 in +-1, seed = stream id."""
    rng = np.random.default_rng(stream_id)
    n = n_frames * FRAME
    if silent is None:
        silent = (stream_id % 10) == 9
    f0 = rng.uniform(80.0, 400.0)
    snr_db = rng.uniform(0.0, 20.0)
    if silent:
        return np.zeros(n, dtype=np.float32)
    t = np.arange(n, dtype=np.float64) / SR
    sig = np.zeros(n, dtype=np.float64)
    for h in range(1, 6):
        sig += np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 2 * np.pi)) / h
    env = 0.5 + 0.5 * np.sin(2 * np.pi * rng.uniform(0.5, 3.0) * t)
    sig *= env
    sig *= 0.2 / max(np.sqrt(np.mean(sig ** 2)), 1e-9)
    noise = rng.standard_normal(n)
    noise *= 0.2 * 10 ** (-snr_db / 20.0)
    return (sig + noise).astype(np.float32)


def cfg1_clip(n_frames: int = 3000) -> np.ndarray:
    """BASELINE cfg 1: 0.3 sin(2 pi 220 t) env(t) + 0.05 N(0,1), default_rng(0)."""
    rng = np.random.default_rng(0)
    n = n_frames * FRAME
    t = np.arange(n, dtype=np.float64) / SR
    env = 0.5 + 0.5 * np.sin(2 * np.pi * 1.0 * t)
    return (0.3 * np.sin(2 * np.pi * 220.0 * t) * env + 0.05 * rng.standard_normal(n)).astype(np.float32)


def batch_np(n_streams: int, n_frames: int, first_stream: int = 0) -> np.ndarray:
    """[n_frames, n_streams, 480] f32 in +-1."""
    x = np.stack([stream_np(first_stream + b, n_frames) for b in range(n_streams)], axis=0)
    return np.ascontiguousarray(x.reshape(n_streams, n_frames, FRAME).transpose(1, 0, 2))


def batch_torch(n_streams: int, n_frames: int, device, first_stream: int = 0, seed: int = 0):
    """Same recipe generated on the device with torch (bench sizes): [n_frames, n_streams, 480],
    already scaled to int16 range.  Per-stream parameters come from a CPU generator seeded with
    `seed`; the noise from a device generator."""
    import torch

    g = torch.Generator(device="cpu").manual_seed(seed + 7919 * first_stream)
    f0 = torch.empty(n_streams).uniform_(80.0, 400.0, generator=g).to(device)
    snr = torch.empty(n_streams).uniform_(0.0, 20.0, generator=g).to(device)
    ph = torch.empty(n_streams, 5).uniform_(0.0, 6.2831853, generator=g).to(device)
    envf = torch.empty(n_streams).uniform_(0.5, 3.0, generator=g).to(device)
    ids = torch.arange(first_stream, first_stream + n_streams, device=device)
    silent = (ids % 10) == 9
    gd = torch.Generator(device=device).manual_seed(seed + 1 + first_stream)
    out = torch.empty(n_frames, n_streams, FRAME, dtype=torch.float32, device=device)
    step = max(1, min(n_frames, (1 << 26) // max(1, n_streams * FRAME)))
    for t0 in range(0, n_frames, step):
        t1 = min(n_frames, t0 + step)
        n = (t1 - t0) * FRAME
        tt = (torch.arange(t0 * FRAME, t0 * FRAME + n, device=device, dtype=torch.float64) / SR)
        sig = torch.zeros(n_streams, n, dtype=torch.float32, device=device)
        for h in range(1, 6):
            arg = (2 * np.pi * h) * f0.double()[:, None] * tt[None, :] + ph[:, h - 1].double()[:, None]
            sig += (torch.sin(arg) / h).float()
        env = (0.5 + 0.5 * torch.sin(2 * np.pi * envf.double()[:, None] * tt[None, :])).float()
        sig = sig * env * (0.2 / 0.56)
        noise = torch.randn(n_streams, n, generator=gd, device=device) * (0.2 * 10 ** (-snr / 20.0))[:, None]
        x = (sig + noise) * 32768.0
        x[silent] = 0.0
        out[t0:t1] = x.view(n_streams, t1 - t0, FRAME).transpose(0, 1)
    return out


def clip16k_np(seed: int, n_samples: int = 480000) -> np.ndarray:
    """Seeded 16 kHz test clip in +-1 for the ASR path (BASELINE cfg 3): sum of chirps with a syllabic
    envelope plus coloured noise.  Deterministic across machines (numpy PCG64)."""
    if n_samples <= 0:
        return np.zeros(0, dtype=np.float32)
    rng = np.random.default_rng(1000 + seed)
    t = np.arange(n_samples, dtype=np.float64) / 16000.0
    x = np.zeros(n_samples, dtype=np.float64)
    for _ in range(4):
        f0, f1 = rng.uniform(100, 1500), rng.uniform(100, 3500)
        ph = 2 * np.pi * (f0 * t + 0.5 * (f1 - f0) * t * t / max(t[-1], 1e-3))
        x += rng.uniform(0.05, 0.25) * np.sin(ph + rng.uniform(0, 6.28))
    env = 0.55 + 0.45 * np.sin(2 * np.pi * rng.uniform(2.0, 5.0) * t + rng.uniform(0, 6.28))
    noise = rng.standard_normal(n_samples)
    noise = np.convolve(noise, np.ones(4) / 4.0, mode="same")
    return (x * env + 0.03 * noise).astype(np.float32)
