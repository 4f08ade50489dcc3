"""GPU tests of the stage between the two halves of the hot path (48->16 kHz resampler, WAV s16 hand-off:
SURVEY.md 8f ranks 1-2) and of the end-to-end denoise -> ASR pipeline (BASELINE cfg 4) against the oracles."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_resampler_out_len_follows_the_chunk_loop():
    from crispy_amd.pipeline import Resampler48to16
    for n in (0, 1, 1023, 1024, 1025, 1026, 2048, 2052, 48000, 1440000, 1439520):
        n_pad = -(-n // 1024) * 1024
        assert Resampler48to16.out_len(n) == n_pad // 1026 * 342


@pytest.mark.parametrize("handoff", [0, 2])
def test_resampler_matches_oracle(handoff):
    import torch
    from crispy_amd import synth_audio
    from crispy_amd.pipeline import Resampler48to16
    from oracle import resample_oracle as RO
    B, n = 5, 48000 * 2 + 777
    x = np.stack([synth_audio.stream_np(200 + b, (n + 479) // 480, silent=False)[:n] * (1.5 if b == 0 else 1.0)
                  for b in range(B)]).astype(np.float32)      # stream 0 clips at +-1
    rs = Resampler48to16()
    n16 = rs.out_len(n)
    d_in = torch.from_numpy(x * np.float32(32768.0)).cuda()
    d_out = torch.zeros(B, n16, device="cuda")
    torch.cuda.synchronize()
    rs.process_device(d_in.data_ptr(), n, n, B, d_out.data_ptr(), n16, scale=1.0 / 32768.0, handoff=handoff)
    rs.synchronize()
    out = d_out.cpu().numpy()
    for b in range(B):
        src = (x[b] * np.float32(32768.0)) * np.float32(1.0 / 32768.0)
        if handoff == 2:
            src = RO.wav_s16_roundtrip(src)
        ref = RO.resample_48k_to_16k(src)
        assert ref.size == n16
        assert np.abs(out[b] - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), b


def test_resampler_is_a_unity_gain_lowpass():
    """Size-independent property at 30 s: a 1 kHz tone passes with unit gain, a 10 kHz tone (above the 8 kHz
    Nyquist) is rejected."""
    import torch
    from crispy_amd.pipeline import Resampler48to16
    n = 1440000
    t = torch.arange(n, device="cuda", dtype=torch.float64) / 48000.0
    x = torch.stack([torch.sin(2 * np.pi * 1000 * t), torch.sin(2 * np.pi * 10000 * t)]).float().contiguous()
    rs = Resampler48to16()
    n16 = rs.out_len(n)
    y = torch.zeros(2, n16, device="cuda")
    torch.cuda.synchronize()
    rs.process_device(x.data_ptr(), n, n, 2, y.data_ptr(), n16)
    rs.synchronize()
    y = y.cpu().numpy()
    assert abs(np.sqrt((y[0, 1000:-1000] ** 2).mean()) - np.sqrt(0.5)) < 1e-3
    assert np.sqrt((y[1, 1000:-1000] ** 2).mean()) < 1e-3


def test_end_to_end_pipeline_matches_oracle_chain(oracle):
    """cfg 4 in miniature: 3 streams x 3 s: RNNoise -> adapter scaling -> s16 WAV -> resample -> log-mel ->
    Whisper-tiny encoder -> greedy ids, every stage from the oracles."""
    import torch
    from crispy_amd import synth_audio, synthetic_weights
    from crispy_amd.asr import WhisperModel
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.pipeline import DenoiseTranscribePipeline
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    from oracle import resample_oracle as RO, whisper_oracle as WO
    B, T = 3, 300
    w = synthetic_weights(0)
    hp = HParams.tiny()
    W = synthetic_whisper_weights(hp, 0)
    x = np.stack([synth_audio.stream_np(300 + b, T, silent=False) for b in range(B)]).reshape(B, T, 480)
    x = np.ascontiguousarray(x * np.float32(32768.0))
    pipe = DenoiseTranscribePipeline(w, WhisperModel(hp, W), B)
    prompt = [50258, 50259, 50359, 50363]
    toks, pcm16 = pipe.run(torch.from_numpy(x).cuda(), prompt, 4)
    pcm16 = pcm16.cpu().numpy()
    F = whisper_mel_filters(80)
    for b in range(B):
        den, _ = oracle.OracleDenoiseState(w).process(x[b])
        a = np.clip(den[1:].ravel() / np.float32(32768.0), -1, 1)          # first frame dropped, clamp
        ref16 = RO.resample_48k_to_16k(RO.wav_s16_roundtrip(a))
        assert ref16.size == pcm16.shape[1]
        # s16 truncation amplifies 1e-7 differences into one LSB (3e-5) on rare samples
        assert np.abs(pcm16[b] - ref16).max() <= 2e-4
        assert np.mean(np.abs(pcm16[b] - ref16) > 1e-5) < 0.01
        enc = WO.encoder_forward(W, hp, oracle.oracle_logmel(ref16, F))
        rt, rb, rm = WO.greedy_decode(W, hp, enc, prompt, 4)
        from tests.test_gpu_whisper import assert_picks
        assert_picks(toks[b, 0], rt, rm, 1e-2, 3, f"pipeline stream {b}")
