"""The slow, GPU-independent oracle computations of the full-size GPU tests as plain functions, and the seeded models / inputs
they run on: used by the tests (through tests/oracle_cache.py) and by tests/golden/make_oracle_cache.py, which computes and
commits their results on the CPU."""
import numpy as np

X_TOK, Y_TOK, REP_TOK = 1234, 2345, 777


def wcpp_masks(hp):
    """whisper.cpp's always-suppressed specials; suppress_blank adds " " and EOT at the first position (oracle ids)."""
    from oracle import whisper_oracle as WO
    sp = WO.special_tokens(hp.n_vocab)
    sup = [sp["sot"], sp["nosp"], sp["translate"], sp["transcribe"], sp["prev"], sp["solm"]]
    sup += list(range(sp["lang0"], sp["lang0"] + sp["n_lang"]))
    return sp, sorted(sup), [220, sp["eot"]]


def ladder_model(hp):
    """Bare prompt (generation starts at position 2): "<|0.00|> w1 {X | Y: a one-logit near tie} w3 <|6.00|><|6.00|> EOT"; with the
    text so far in front (position 9): <|0.00|>, one token 40 times, a timestamp pair -- the entropy check fails it."""
    from tests.scripted_model import script_rows, scripted_whisper_weights
    sp, _, _ = wcpp_masks(hp)
    BEG, EOT = sp["beg"], sp["eot"]
    beta = 1.0 - 1.0 * np.sqrt(2.0) / hp.n_text_state            # logit(X) - logit(Y) = 1 at gain 100
    rows = script_rows(2, [BEG, 1001, [(X_TOK, 1.0), (Y_TOK, beta)], 1003, BEG + 300, BEG + 300, EOT])
    rows.update(script_rows(9, [BEG] + [REP_TOK] * 40 + [BEG + 100, BEG + 100, EOT]))
    return scripted_whisper_weights(hp, rows, gain=100.0)


def repeat_model(hp):
    """A model that repeats itself whatever the prompt (window 2: 1 + 43 + 3 tokens of prompt)."""
    from tests.scripted_model import script_rows, scripted_whisper_weights
    sp, _, _ = wcpp_masks(hp)
    BEG, EOT = sp["beg"], sp["eot"]
    rows = script_rows(2, [BEG] + [REP_TOK] * 40 + [BEG + 100, BEG + 100, EOT])
    rows.update(script_rows(46, [BEG] + [REP_TOK] * 40 + [BEG + 100, BEG + 100, EOT]))
    return scripted_whisper_weights(hp, rows, gain=100.0)


def nospeech_model(hp):
    """Sure of <|nospeech|> at the start of every window, unsure of everything it then says (flat logits)."""
    from tests.scripted_model import script_rows, scripted_whisper_weights
    sp, _, _ = wcpp_masks(hp)
    BEG, EOT = sp["beg"], sp["eot"]
    rows = script_rows(2, [[(BEG, 1.0), (sp["nosp"], 1.0)], 1001, BEG + 1400, BEG + 1400, EOT])
    return scripted_whisper_weights(hp, rows, gain=1.0, boost={sp["nosp"]: 6.0})


def scripted_whisper_full(W, hp, n_samples, mode, **kw):
    """The oracle's whisper_full on a scripted model (tests/scripted_model.py: the decoder ignores the audio, so the encoder
    pass is skipped).  Token text = the synthetic vocabulary the tests' model files are written with."""
    from crispy_amd.ggml_io import synthetic_vocab
    from oracle import whisper_oracle as WO
    sp, sup, sup_first = wcpp_masks(hp)
    vocab = synthetic_vocab(hp.n_vocab)
    enc0 = np.zeros((hp.n_audio_ctx, hp.n_audio_state))
    return WO.transcribe_timestamps(W, hp, lambda seek: None, n_samples, [sp["sot"], sp["lang0"], sp["transcribe"]],
                                    WO.RULES_WCPP, lambda t: vocab[t], suppress=sup, suppress_first=sup_first, f16=(mode == 1),
                                    fallback=True, encoder=lambda mel: enc0, **kw)


def scripted_ref(W, hp, n_samples, mode, **kw):
    """scripted_whisper_full through the committed cache."""
    from tests.oracle_cache import cached, fingerprint
    fp = fingerprint("scripted_whisper_full", W["decoder.positional_embedding"], W["decoder.token_embedding.weight"], W["decoder.ln.weight"],
                     n_samples, mode, kw)
    return cached(f"scripted_whisper_full_{fp[:12]}", fp, lambda: scripted_whisper_full(W, hp, n_samples, mode, **kw))


def scripted_cases():
    """(W, hp, n_samples, mode, kw) of every scripted whisper_full run the GPU suite makes (tests/test_gpu_decision.py)."""
    from crispy_amd.whisper_weights import HParams
    hp = HParams.tiny()
    lad, rep, nos = ladder_model(hp), repeat_model(hp), nospeech_model(hp)
    return [(lad, hp, 16000 * 13, 0, {}), (lad, hp, 16000 * 13, 1, {}), (rep, hp, 16000 * 4, 0, {"params": {"best_of": 2}}),
            (nos, hp, 480000, 0, {}), (nos, hp, 480000, 1, {}), (lad, hp, 16000 * 13, 0, {"params": {"beam_size": 3}})]


CATALOG_ROWS = 96      # encoder rows kept per catalog model (every ~15th of 1500): 96 x d floats instead of 1500 x d


def catalog_case(name):
    """A catalog model at full depth (src-tauri/src/managers/model.rs:74-148) with seeded weights: the oracle's encoder output of
    one 10 s clip (CATALOG_ROWS rows of it, spread over the 1500, and its peak) and three greedy picks of the oracle's KV-cached
    decoder with their top-2 margins.  Minutes of numpy for medium / large-v3."""
    from crispy_amd import synth_audio
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    from oracle import whisper_oracle as WO
    from tests import oracle_lib
    hp = getattr(HParams, name)()
    W = synthetic_whisper_weights(hp, 3)
    x = synth_audio.clip16k_np(77, 160000)
    # medium / large-v3: the oracle in single precision (its own rounding is ~1e-6 of the peak against a bar of 1e-4)
    ref = WO.encoder_forward(W, hp, oracle_lib.oracle_logmel(x, whisper_mel_filters(hp.n_mels)),
                             dtype=np.float64 if name == "small" else np.float32).astype(np.float64)
    prompt = WO.default_prompt(hp.n_vocab, no_timestamps=True)
    dc = WO.DecoderCache(W, hp, ref)
    for t in prompt[:-1]:
        dc.step(t)
    tok, picks, margins = prompt[-1], [], []
    for _ in range(3):
        lg = dc.step(tok)
        tok = int(np.argmax(lg))
        top2 = np.partition(lg, -2)[-2:]
        picks.append(tok)
        margins.append(float(top2[1] - top2[0]))
    rows = np.linspace(0, 1499, CATALOG_ROWS).round().astype(np.int64)
    return {"rows": rows, "ref_rows": ref[rows].astype(np.float32), "peak": float(np.abs(ref).max()), "picks": picks, "margins": margins}


def catalog_ref(name):
    from tests.oracle_cache import cached, fingerprint
    fp = fingerprint("catalog_case", name, CATALOG_ROWS, 3, 77, 160000)
    return cached(f"catalog_{name}", fp, lambda: catalog_case(name))
