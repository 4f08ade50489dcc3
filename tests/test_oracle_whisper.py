"""CPU tests of the numpy Whisper oracle (oracle/whisper_oracle.py) against golden vectors produced by
HuggingFace transformers with the same seeded weights (tests/golden/make_whisper_golden.py)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "whisper_tiny_golden.npz")


@pytest.fixture(scope="module")
def tiny():
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    hp = HParams.tiny()
    return hp, synthetic_whisper_weights(hp, 0)


@pytest.fixture(scope="module")
def enc_out(tiny, oracle):
    from crispy_amd import synth_audio
    from crispy_amd.mel_filters import whisper_mel_filters
    from oracle import whisper_oracle as WO
    hp, W = tiny
    mel = oracle.oracle_logmel(synth_audio.clip16k_np(0, 464000), whisper_mel_filters(80))
    return WO.encoder_forward(W, hp, mel)


def test_tensor_inventory_and_param_count(tiny):
    from crispy_amd.whisper_weights import HParams, sinusoids, tensor_shapes
    hp, W = tiny
    assert sum(v.size for v in W.values()) == 37760640          # HF random-init tiny, tied output projection
    assert list(W) == list(tensor_shapes(hp))
    assert "encoder.blocks.0.attn.key.bias" not in W            # k has no bias (Appendix B.2)
    assert np.array_equal(W["encoder.positional_embedding"], sinusoids(1500, 384))
    base = tensor_shapes(HParams.base())
    assert base["encoder.blocks.5.mlp.0.weight"] == (2048, 512)


def test_encoder_oracle_matches_hf_golden(enc_out):
    G = np.load(GOLD)
    assert enc_out.shape == (1500, 384)
    ref = G["enc_rows"]
    assert np.abs(enc_out[::25] - ref).max() <= 2e-5 * np.abs(ref).max()
    assert abs(np.abs(enc_out).mean() - float(G["enc_mean_abs"])) < 1e-5


def test_decoder_oracle_matches_hf_golden(tiny, enc_out):
    from oracle import whisper_oracle as WO
    hp, W = tiny
    G = np.load(GOLD)
    prompt = G["prompt"].tolist()
    lg = WO.decoder_logits(W, hp, enc_out, prompt)
    assert np.abs(lg[:, ::997] - G["prompt_logits_sample"]).max() < 5e-5
    assert np.array_equal(lg.argmax(-1), G["prompt_argmax"])
    toks, best, margin = WO.greedy_decode(W, hp, enc_out, prompt, 4)
    assert toks == G["greedy_tokens"][:4].tolist()
    assert np.abs(np.array(best) - G["greedy_logits"][:4]).max() < 5e-5


def test_causal_mask_prefix_invariance(tiny, enc_out):
    """Logits of position i depend only on tokens <= i."""
    from oracle import whisper_oracle as WO
    hp, W = tiny
    a = WO.decoder_logits(W, hp, enc_out, [50258, 50259, 50359, 50363, 11, 22])
    b = WO.decoder_logits(W, hp, enc_out, [50258, 50259, 50359, 50363, 99, 77])
    assert np.abs(a[:4] - b[:4]).max() < 1e-10 and np.abs(a[4] - b[4]).max() > 1e-3
