"""GPU parity tests of the log-mel front end (crispy_mel_* behind the C ABI) against the CPU oracle
(whisper.cpp semantics) and the HuggingFace golden vectors.
Tolerance (north_star): log-mel within 1e-4 relative -> max |gpu - ref| <= 1e-4 * max |ref|."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "logmel_golden.npz")


def _tol(ref):
    return 1e-4 * np.abs(ref).max()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_logmel_matches_oracle_and_hf_golden(oracle, seed):
    from crispy_amd import synth_audio
    from crispy_amd.asr import LogMel
    G = np.load(GOLD)
    n = int(G[f"clip{seed}/n"])
    x = synth_audio.clip16k_np(seed, n)
    mel = LogMel(80, G["filters"])([x])[0]
    ref = oracle.oracle_logmel(x, G["filters"])
    assert np.abs(mel - ref).max() <= _tol(ref)
    assert np.abs(mel[:, ::7][:, :-1] - G[f"clip{seed}/mel_every7"][:, :-1]).max() <= _tol(ref)


def test_ragged_batch_and_edge_lengths(oracle):
    """Clips of different lengths in one call, incl. 1 sample, < one frame, and exactly 30 s."""
    from crispy_amd import synth_audio
    from crispy_amd.asr import LogMel
    from crispy_amd.mel_filters import whisper_mel_filters
    F = whisper_mel_filters(80)
    lens = [1, 100, 399, 401, 16000, 123457, 480000]
    clips = [synth_audio.clip16k_np(10 + i, n) for i, n in enumerate(lens)]
    clips.append(np.zeros(32000, np.float32))            # digital silence
    mel = LogMel(80, F)(clips)
    for i, c in enumerate(clips):
        ref = oracle.oracle_logmel(c, F)
        assert np.abs(mel[i] - ref).max() <= _tol(ref), f"clip {i} len {c.size}"
    assert np.all(mel[-1] == np.float32(-1.5))


def test_device_entry_point_and_transposed_layout(oracle):
    import torch
    from crispy_amd import synth_audio
    from crispy_amd.asr import LogMel
    from crispy_amd.mel_filters import whisper_mel_filters
    F = whisper_mel_filters(80)
    B = 5
    x = np.stack([synth_audio.clip16k_np(20 + i, 480000) for i in range(B)])
    lm = LogMel(80, F)
    host = lm(x)
    d_pcm = torch.from_numpy(x).cuda()
    d_out = torch.empty(B, 80, 3000, device="cuda")
    d_out_t = torch.full((B, 3002, 80), 7.0, device="cuda")
    d_out_t[:, 0] = 0
    d_out_t[:, -1] = 0
    torch.cuda.synchronize()
    lm.compute_device(d_pcm.data_ptr(), 480000, np.full(B, 480000), d_out.data_ptr(), d_out_t.data_ptr())
    lm.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), host)
    t = d_out_t.cpu().numpy()
    assert np.array_equal(t[:, 1:-1].transpose(0, 2, 1), host)
    assert np.all(t[:, 0] == 0) and np.all(t[:, -1] == 0)


def test_128_mel_front_end(oracle):
    """large-v3 / turbo use 128 mels (model catalog: managers/model.rs:118-137)."""
    from crispy_amd import synth_audio
    from crispy_amd.asr import LogMel
    from crispy_amd.mel_filters import whisper_mel_filters
    F = whisper_mel_filters(128)
    x = synth_audio.clip16k_np(30, 200000)
    mel = LogMel(128, F)([x])[0]
    ref = oracle.oracle_logmel(x, F)
    assert mel.shape == (128, 3000) and np.abs(mel - ref).max() <= _tol(ref)


def test_full_batch_64_clips_properties():
    """BASELINE cfg 3 size (64 x 30 s): batch independence, bounds of the normalised range."""
    import torch
    from crispy_amd import synth_audio
    from crispy_amd.asr import LogMel
    B = 64
    g = torch.Generator(device="cuda").manual_seed(5)
    d_pcm = (torch.randn(B, 480000, generator=g, device="cuda") * 0.1)
    d_pcm[7] = d_pcm[3]
    x3 = synth_audio.clip16k_np(40, 480000)
    d_pcm[11] = torch.from_numpy(x3).cuda()
    d_out = torch.empty(B, 80, 3000, device="cuda")
    lm = LogMel(80)
    torch.cuda.synchronize()
    lm.compute_device(d_pcm.data_ptr(), 480000, np.full(B, 480000), d_out.data_ptr())
    lm.synchronize()
    out = d_out.cpu().numpy()
    assert np.isfinite(out).all()
    assert np.array_equal(out[7], out[3])
    solo = lm([x3])[0]
    assert np.array_equal(solo, out[11])
    for b in range(B):     # after clamping, every clip spans at most 8 decades: (max - min) * 4 <= 8
        assert (out[b].max() - out[b].min()) * 4.0 <= 8.0 + 1e-4


def test_mel_error_paths():
    from crispy_amd import _native as N
    from crispy_amd.asr import LogMel
    lm = LogMel(80)
    with pytest.raises(N.CrispyError):
        lm([np.zeros(480001, np.float32)])
    assert lm([]).shape == (0, 80, 3000)
    with pytest.raises(N.CrispyError):
        LogMel(80, device=99)
