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


TS_GOLD = os.path.join(os.path.dirname(__file__), "golden", "whisper_tiny_ts_golden.npz")


def test_cached_decoder_equals_full_decoder(tiny, enc_out):
    from oracle import whisper_oracle as WO
    hp, W = tiny
    toks = [50258, 50259, 50359, 100, 200, 300]
    ref = WO.decoder_logits(W, hp, enc_out, toks)
    dc = WO.DecoderCache(W, hp, enc_out)
    for i, t in enumerate(toks):
        assert np.abs(dc.step(t) - ref[i]).max() < 1e-10


def test_timestamp_rules_match_hf_processor_golden(tiny, enc_out):
    """RULES_OPENAI of the oracle reproduces, token for token, greedy decoding through HuggingFace's
    WhisperTimeStampLogitsProcessor (tests/golden/make_whisper_ts_golden.py) -- 40 picks incl. the forced first
    timestamp, pairs, monotonic timestamps and the probability-mass rule."""
    from oracle import whisper_oracle as WO
    hp, W = tiny
    G = np.load(TS_GOLD)
    sp = WO.special_tokens(hp.n_vocab)
    assert (sp["not_"], sp["beg"], sp["transcribe"]) == (50363, 50364, 50359)
    dc = WO.DecoderCache(W, hp, enc_out)                     # enc_out is clip16k_np(0, 464000), golden clip 0
    win = WO.decode_window(dc.step, G["prompt"].tolist(), sp, WO.RULES_OPENAI, 40, 0, 10 ** 9,
                           G["suppress"], G["suppress_first"])
    assert win["tokens"] == G["c0_tokens"].tolist()
    assert np.allclose(win["margins"], G["c0_margins"], atol=2e-3)


def test_timestamp_rule_flavours_and_segments(tiny):
    from oracle import whisper_oracle as WO
    hp, _ = tiny
    sp = WO.special_tokens(hp.n_vocab)
    beg, eot = sp["beg"], sp["eot"]
    rng = np.random.default_rng(3)
    lg = rng.standard_normal(hp.n_vocab)
    # first pick: OPENAI forces a timestamp <= 1.00 s; whisper.cpp only caps timestamps at 1.00 s
    m, _ = WO.timestamp_rules(lg, [], sp, WO.RULES_OPENAI)
    assert np.isneginf(m[:beg]).all() and np.isneginf(m[beg + 51:]).all() and np.isfinite(m[beg:beg + 51]).all()
    lg2 = lg.copy(); lg2[100] = 50.0                       # one dominant text token
    m, _ = WO.timestamp_rules(lg2, [], sp, WO.RULES_WCPP)
    assert int(np.argmax(m)) == 100 and np.isneginf(m[beg + 51:]).all()
    # after a lone timestamp: text only; after text + timestamp: timestamp or EOT only
    m, _ = WO.timestamp_rules(lg, [beg + 10], sp, WO.RULES_WCPP)
    assert np.isneginf(m[beg:]).all()
    lg3 = lg.copy(); lg3[eot] = 50.0                      # EOT dominant, so the mass rule does not remove it
    m, _ = WO.timestamp_rules(lg3, [beg + 10, 7, beg + 20], sp, WO.RULES_WCPP)
    assert np.isneginf(m[:eot]).all() and np.isfinite(m[eot])
    assert np.isneginf(m[beg:beg + 20]).all() and np.isfinite(m[beg + 20])          # may repeat the last one
    m, _ = WO.timestamp_rules(lg, [beg + 10, 7, beg + 20, beg + 20], sp, WO.RULES_OPENAI)
    assert np.isneginf(m[beg:beg + 21]).all()                                         # closed pair: strictly later
    m, _ = WO.timestamp_rules(lg, [beg + 10, 7, beg + 20, beg + 20], sp, WO.RULES_WCPP)
    assert np.isneginf(m[beg:beg + 20]).all()
    # segments: "<|0.20|> a b <|1.00|><|1.00|> c <|2.50|>" then an open tail "d" that result_len drops
    text = lambda t: b" w%d" % t
    win = dict(tokens=[beg + 10, 1, 2, beg + 50, beg + 50, 3, beg + 125, 4, eot], result_len=7, seek_delta=250,
               tids=[beg + 10, beg, beg, beg + 50, beg + 50, beg, beg + 125, beg, beg])
    assert WO.window_segments(win, 1000, sp, text) == [(1020, 1100, b" w1 w2"), (1100, 1250, b" w3")]
    win = dict(tokens=[5, 6, eot], tids=[beg + 3, beg, beg], result_len=3, seek_delta=3000)
    assert WO.window_segments(win, 0, sp, text) == [(6, 3000, b" w5 w6")]


def test_f16_operand_encoder_oracle_is_a_small_perturbation_of_the_exact_one(oracle):
    """encoder_forward_f16 (operands of every matrix product rounded to f16, ggml's mul_mat numerics) against the exact
    graph on two layers of the tiny architecture: different (the rounding is there) and close (it is only rounding)."""
    import dataclasses
    from crispy_amd import synth_audio
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    from oracle import whisper_oracle as WO
    hp = dataclasses.replace(HParams.tiny(), n_audio_layer=2)
    W = synthetic_whisper_weights(hp, 0)
    mel = oracle.oracle_logmel(synth_audio.clip16k_np(3, 100000), whisper_mel_filters(80))
    a = WO.encoder_forward(W, hp, mel)
    b = WO.encoder_forward_f16(W, hp, mel)
    rel = np.abs(a - b).max() / np.abs(a).max()
    assert 1e-4 < rel < 5e-3, rel
    # attn16 (precision mode 2): normalised probabilities rounded instead of the 2^(t - m) mantissas -- another rounding
    # of the same size at another point: different from mode 1's, as far from the exact graph
    c = WO.encoder_forward_f16(W, hp, mel, attn16=True)
    assert 1e-6 < np.abs(c - b).max() / np.abs(a).max() < 5e-3
    assert 1e-4 < np.abs(a - c).max() / np.abs(a).max() < 5e-3
    assert WO._h(1.0 + 2.0 ** -11) == 1.0 and WO._h(1.0 + 3 * 2.0 ** -11) == 1.0 + 2.0 ** -9    # round to nearest even
    assert WO._h(70000.0) == np.inf                                                                # f16 range


def test_final_logits_is_the_decoders_last_block_and_its_f16_form_rounds_both_operands(tiny):
    """oracle final_logits(x) = LN(x) . E^T: (a) on random decoder states it equals the direct formula; (b) the f16 form
    differs from it by the rounding of both operands -- 6e-5 rms of the peak with the seeded Whisper-tiny weights --
    and is what one gets from f16-rounded inputs in exact arithmetic (no other rounding point)."""
    from oracle import whisper_oracle as WO
    hp, W = tiny
    rng = np.random.default_rng(3)
    x = rng.standard_normal((5, hp.n_text_state)) * 2.0 + 0.3
    g, b = W["decoder.ln.weight"].astype(np.float64), W["decoder.ln.bias"].astype(np.float64)
    E = W["decoder.token_embedding.weight"].astype(np.float64)
    mu = x.mean(-1, keepdims=True)
    xn = (x - mu) / np.sqrt(((x - mu) ** 2).mean(-1, keepdims=True) + 1e-5) * g + b
    ref = WO.final_logits(W, x)
    assert np.allclose(ref, xn @ E.T, rtol=0, atol=1e-9 * np.abs(ref).max())
    ref16 = WO.final_logits(W, x, f16=True)
    xh = xn.astype(np.float32).astype(np.float16).astype(np.float64)
    Eh = E.astype(np.float32).astype(np.float16).astype(np.float64)
    assert np.array_equal(ref16, xh @ Eh.T)
    gap = np.sqrt(np.mean((ref16 - ref) ** 2)) / np.abs(ref).max()
    assert 4e-5 < gap < 3e-4, gap


def test_decoder_cache_f16_mode_only_rounds_where_it_says(tiny, enc_out):
    """oracle DecoderCache(f16=True) = the exact cached decoder plus f16 roundings of the K|V caches and of the operands
    of the plain products: (a) with f16=False it reproduces decoder_logits; (b) with f16=True the logits move by
    1e-5 .. 1e-2 of their scale (rounding, not a different model) and the arg-max of a well-separated step is kept."""
    from oracle import whisper_oracle as WO
    hp, W = tiny
    toks = [50258, 50259, 50359, 50363, 1000]
    full = WO.decoder_logits(W, hp, enc_out, toks)
    dc, dh = WO.DecoderCache(W, hp, enc_out), WO.DecoderCache(W, hp, enc_out, f16=True)
    for i, t in enumerate(toks):
        l, lh = dc.step(t), dh.step(t)
        assert np.abs(l - full[i]).max() <= 1e-9 * np.abs(full[i]).max()
        rel = np.abs(lh - l).max() / np.abs(l).max()
        assert 1e-5 < rel < 1e-2, (i, rel)
    # (c) ln16 (on by default with f16: the library's precision modes 1 and 2 since round 5) rounds the LayerNorm outputs and
    # the q | k | v, cross-q, fc1 weights on top of the plain products: a further small move away from the chain without
    # it (ln16=False: the mode 1 of rounds 2 - 4); without f16 the flag does nothing
    d2, d0 = WO.DecoderCache(W, hp, enc_out, f16=True), WO.DecoderCache(W, hp, enc_out, f16=False, ln16=True)
    assert d2.ln16 and not d0.ln16
    dh = WO.DecoderCache(W, hp, enc_out, f16=True, ln16=False)
    dc = WO.DecoderCache(W, hp, enc_out)
    for i, t in enumerate(toks):
        l2, lh, l0, l = d2.step(t), dh.step(t), d0.step(t), dc.step(t)
        assert np.array_equal(l0, l)
        rel = np.abs(l2 - lh).max() / np.abs(lh).max()
        assert 1e-5 < rel < 1e-2, (i, rel)
    # (d) attn16=True (mode 2 as well): q and the normalised probabilities rounded to f16 inside the attentions -- a further
    # small move; without f16 the flag does nothing
    d3, d2 = WO.DecoderCache(W, hp, enc_out, f16=True, ln16=True, attn16=True), WO.DecoderCache(W, hp, enc_out, f16=True, ln16=True)
    d0, dc = WO.DecoderCache(W, hp, enc_out, attn16=True), WO.DecoderCache(W, hp, enc_out)
    for i, t in enumerate(toks):
        l3, l2, l0, l = d3.step(t), d2.step(t), d0.step(t), dc.step(t)
        assert np.array_equal(l0, l)
        rel = np.abs(l3 - l2).max() / np.abs(l2).max()
        assert 1e-6 < rel < 1e-2, (i, rel)
