"""End-to-end parity of PRECISION MODE 1 -- the precision that ships: the Rust binding's default
(bindings/rust/crispy-hip-sys/src/lib.rs, `GpuWhisperEngine::load`), `bench.py --workload cfg5` and every headline ASR
number run in it.  Reference call site: managers/transcription.rs:183-185.

Every test here runs the library entry point a host would call (`crispy_asr_transcribe_tokens`, `crispy_asr_transcribe`,
`crispy_asr_transcribe_batch`, `crispy_asr_detect_language_device`) with `crispy_asr_set_precision(h, 1)` and compares
with the CHAINED f16-operand oracle: oracle log-mel -> `encoder_forward_f16` -> `DecoderCache(f16=True)` (oracle/
whisper_oracle.py: both operands of every plain matrix product rounded to f16, exact accumulation; ggml's mul_mat
arithmetic [UPSTREAM-RECALL]).

Comparison rule (`forced_picks`).  The oracle is teacher-forced with the GPU's own tokens, so EVERY step is compared, not
only the prefix up to the first close call.  Two assertions per step: (1) the GPU's pick is never further than `thr`
below the oracle's best logit; (2) where the oracle's own top-2 margin exceeds `thr` the ids are equal -- and a minimum
number of such resolvable steps is required (a margin-gated loop that compares nothing proves nothing).
`thr` derives from the measured bar of the mode: a picked logit agrees with the f16 oracle to 4e-4 of the logit scale at
the worst value (tests/test_gpu_whisper.py::test_mode_1_decoder_matches_the_f16_arithmetic_oracle), so thr = MODE1_REL x
scale with MODE1_REL = 4 x 4e-4 wherever the oracle decoder runs on the SAME encoder output as the product's.  The full
chain (oracle encoder -> oracle decoder) is compared too, at rel = 0.03: the audio-sensitive weights amplify the encoder's
3e-4 into logit-gap changes of ~1e-2 (see test_mode1_transcribe_tokens_against_the_chained_f16_oracle).

Weights: `synthetic_whisper_weights(..., sensitive=True)` -- sharpened cross-attention, so that the picks depend on the
audio (the tests assert that different clips decode to different ids)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODE1_REL = 4 * 4e-4


def forced_picks(W, hp, enc, prompt, got, need, what, f16=True, suppress=None, rel=MODE1_REL):
    """Teacher-forced comparison of greedy ids `got` (one clip) with the oracle decoder on encoder output `enc`."""
    from oracle import whisper_oracle as WO
    sp = WO.special_tokens(hp.n_vocab)
    dc = WO.DecoderCache(W, hp, enc, f16=f16)
    for t in prompt[:-1]:
        dc.step(t)
    tok, compared, worst = prompt[-1], 0, 0.0
    for i, g in enumerate(got):
        g = int(g)
        lg = dc.step(tok)
        if suppress is not None:
            lg = lg.copy()
            lg[np.asarray(suppress, dtype=np.int64)] = -np.inf
        thr = rel * float(np.abs(lg[np.isfinite(lg)]).max())
        best = int(np.argmax(lg))
        top2 = np.partition(lg, -2)[-2:]
        short = float(lg[best] - lg[g])
        worst = max(worst, short)
        assert short <= thr, (what, i, g, best, short, thr)
        if top2[1] - top2[0] > thr:
            assert g == best, (what, i, g, best, float(top2[1] - top2[0]), thr)
            compared += 1
        tok = g
        if g == sp["eot"]:
            break
    assert compared >= need, (what, f"only {compared} picks had an oracle margin above the bar")
    return compared, worst


MODELS = {  # name: (weights seed, clips as (seed, samples), new tokens, resolvable picks required per clip)
    "tiny": (0, ((300, 464000), (301, 130000), (302, 52000)), 8, 4),
    "base": (1, ((310, 300000),), 6, 4),
    "small": (3, ((320, 160000),), 5, 4),          # one catalog model (managers/model.rs:74-93) at full depth
}


@pytest.mark.parametrize("name", list(MODELS))
def test_mode1_transcribe_tokens_against_the_chained_f16_oracle(oracle, name):
    """`crispy_asr_transcribe_tokens` in mode 1: host PCM -> log-mel -> f16 encoder -> cross K|V -> greedy decode, ids
    against the chained f16 oracle; also closer to that oracle than the exact one would demand, and audio dependent."""
    from crispy_amd import synth_audio
    from crispy_amd.asr import WhisperModel
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    from oracle import whisper_oracle as WO
    wseed, clip_spec, n_new, need = MODELS[name]
    hp = getattr(HParams, name)()
    W = synthetic_whisper_weights(hp, wseed, sensitive=True)
    m = WhisperModel(hp, W)
    m.set_precision(1)
    clips = [synth_audio.clip16k_np(s, n) for s, n in clip_spec]
    prompt = WO.default_prompt(hp.n_vocab, no_timestamps=True)
    toks, n = m.transcribe_tokens(clips, prompt, n_new)
    again, _ = m.transcribe_tokens(clips, prompt, n_new)
    assert np.array_equal(toks, again)                                   # deterministic
    F = whisper_mel_filters(hp.n_mels)
    total = 0
    for b, c in enumerate(clips):
        solo, _ = m.transcribe_tokens([c], prompt, n_new)
        assert np.array_equal(solo[0], toks[b]), (name, b)              # batch == solo in mode 1
        enc16 = WO.encoder_forward_f16(W, hp, oracle.oracle_logmel(c, F))
        # (1) the encoder half of the chain: the product's own encoder output against the f16 oracle, at the encoder's bar
        enc_gpu = m.encode([c])[0]
        e_err = float(np.abs(enc_gpu - enc16).max() / np.abs(enc16).max())
        e_bar = 4e-4 * float(np.sqrt(hp.n_audio_layer / 4.0))        # 4e-4 is the 4-layer figure; roundings add in quadrature
        assert e_err <= e_bar, (name, b, e_err, e_bar)
        # (2) the decoder half at the decoder's bar: teacher-forced oracle decoder ON THE PRODUCT'S encoder output
        k, worst = forced_picks(W, hp, enc_gpu.astype(np.float64), prompt, toks[b], need, f"{name} clip {b} (decoder)")
        # (3) the chain against the oracle's own encoder output.  These weights are audio-SENSITIVE on purpose (the ids must
        # depend on the clip), and they amplify: an encoder difference of 3e-4 of the peak moves the gap between two logits
        # by up to 0.014 here (tools/diag_mode1_long.py, clip 300: the oracle decoder itself picks 44116 on the product's
        # encoder output and 47822 on the oracle's, margins 0.0075 and 0.0066).  So the chain is held to the bar the cfg 4
        # test uses for the same reason (tests/test_gpu_pipeline.py): 0.03 of the logit scale, every step compared.
        # (tiny and base only: at 12 + 12 layers the second teacher-forced oracle pass costs more than the rest of the case)
        kc, worst_c = (forced_picks(W, hp, enc16, prompt, toks[b], 1, f"{name} clip {b} (chain)", rel=0.03)
                       if hp.n_text_layer <= 6 else (0, float("nan")))
        print(f"mode 1 {name} clip {b}: encoder {e_err:.2e} of the peak; decoder {k} of {n_new} picks resolvable, worst shortfall "
              f"{worst:.2e}; chain worst shortfall {worst_c:.2e}; ids {toks[b].tolist()}")
        total += k
    if len(clips) > 1:
        assert len({tuple(t.tolist()) for t in toks}) == len(clips), toks     # the ids depend on the audio
    m.close()


def _engine_file(tmp_path_factory, hp, W, tag):
    from crispy_amd.ggml_io import synthetic_vocab, write_ggml
    from crispy_amd.mel_filters import whisper_mel_filters
    path = tmp_path_factory.mktemp("ggml_m1") / f"ggml-{tag}.bin"
    write_ggml(str(path), hp, W, whisper_mel_filters(hp.n_mels), synthetic_vocab(hp.n_vocab), f16=False)
    return str(path)


def _wcpp_masks(hp):
    from oracle import whisper_oracle as WO
    sp = WO.special_tokens(hp.n_vocab)
    sup = [sp["sot"], sp["nosp"], sp["translate"], sp["transcribe"], sp["prev"], sp["solm"]]
    sup += list(range(sp["lang0"], sp["lang0"] + sp["n_lang"]))
    return sp, sorted(sup), [220, sp["eot"]]


def test_mode1_transcribe_segments_follow_the_f16_seek_loop(oracle, tmp_path_factory):
    """`crispy_asr_transcribe` with whisper.cpp's default options (timestamps on, seek loop) in MODE 1 against
    `transcribe_timestamps(f16=True)`: two windows, kept tokens, segment times and texts; `opts == NULL`; batch == single
    calls in mode 1; a second, long clip on audio-sensitive weights compared up to its first unresolvable pick."""
    import ctypes as C
    from crispy_amd import _native as N, synth_audio
    from crispy_amd.asr import WhisperEngine, transcribe_batch
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    from oracle import whisper_oracle as WO
    hp = HParams.tiny()
    F = whisper_mel_filters(80)
    sp, sup, sup_first = _wcpp_masks(hp)
    prompt = [sp["sot"], sp["lang0"], sp["transcribe"]]
    # (a) plain weights, 3.5 s clip: the oracle's windows all resolve (smallest margin 0.017 against a bar of ~0.007)
    W = synthetic_whisper_weights(hp, 0)
    eng = WhisperEngine(_engine_file(tmp_path_factory, hp, W, "tiny-s0"))
    eng.set_precision(1)
    x = synth_audio.clip16k_np(52, 56000)
    text, segs, toks = eng.transcribe_segments(x, max_new_tokens=10, language_token=sp["lang0"], fallback=False)
    rsegs, rkept, wins = WO.transcribe_timestamps(W, hp, lambda seek: oracle.oracle_logmel(x, F, seek), x.size, prompt,
                                                  WO.RULES_WCPP, eng.token_text, n_max=10, suppress=sup,
                                                  suppress_first=sup_first, max_windows=16, f16=True)
    assert len(wins) >= 2
    bar = MODE1_REL * 4.6                       # logit scale of this model: |logit| <= 4.6
    assert min(min(w["margins"]) for w in wins) > bar, "test clip has a pick the mode cannot resolve; choose another seed"
    assert toks == [t for t in rkept if t != sp["eot"]], (toks, rkept)
    assert [(round(a * 100), round(b * 100), s) for a, b, s in segs] == [(a, b, s.decode()) for a, b, s in rsegs]
    assert text == "".join(s for _, _, s in segs)
    # opts == NULL = TranscribeOptions::default() (managers/transcription.rs:184): language detected, timestamps on
    res = C.c_void_p()
    N.check(N.lib().crispy_asr_transcribe(eng._h, x.ctypes.data, x.size, None, C.byref(res)))
    r = C.cast(res, C.POINTER(N.AsrResult)).contents
    lang_null, n_tok_null = int(r.language_token), int(r.n_tokens)
    N.lib().crispy_asr_free_result(res)
    _, segs_auto, toks_auto = eng.transcribe_segments(x, max_new_tokens=0)
    assert lang_null == eng.last_language_token and n_tok_null == len(toks_auto) and lang_null >= sp["lang0"]
    # batch == single calls, mode 1
    clips = [x, synth_audio.clip16k_np(53, 90000), np.zeros(0, np.float32), synth_audio.clip16k_np(54, 1500)]
    got = transcribe_batch(eng, clips, max_new_tokens=10, language_token=sp["lang0"], timestamps=True, with_segments=True, fallback=False)
    assert got[0][:4] == (text, toks, sp["lang0"], segs)
    t1, s1, k1 = eng.transcribe_segments(clips[1], max_new_tokens=10, language_token=sp["lang0"], fallback=False)
    assert got[1][:4] == (t1, k1, sp["lang0"], s1) and got[2] == ("", [], 0, [], []) and got[3][:2] == ("", [])
    eng.close()
    # (b) audio-sensitive weights, 25 s clip: the kept tokens up to the oracle's first unresolvable pick
    Ws = synthetic_whisper_weights(hp, 0, sensitive=True)
    eng = WhisperEngine(_engine_file(tmp_path_factory, hp, Ws, "tiny-s0-sensitive"))
    eng.set_precision(1)
    y = synth_audio.clip16k_np(53, 400000)
    _, _, ytoks = eng.transcribe_segments(y, max_new_tokens=10, language_token=sp["lang0"], fallback=False)
    _, ykept, ywins = WO.transcribe_timestamps(Ws, hp, lambda seek: oracle.oracle_logmel(y, F, seek), y.size, prompt,
                                               WO.RULES_WCPP, eng.token_text, n_max=10, suppress=sup,
                                               suppress_first=sup_first, max_windows=2, f16=True)
    ok = 0
    for w in ywins:
        kept_m = w["margins"][:w["result_len"]]
        if min(w["margins"]) <= bar:
            ok += int(np.argmax(np.array(kept_m) <= bar)) if (np.array(kept_m) <= bar).any() else len(kept_m)
            break
        ok += w["result_len"]
    ref = [t for t in ykept[:ok]]
    assert ok >= 5, (ok, [w["margins"] for w in ywins])
    assert ytoks[:len([t for t in ref if t != sp["eot"]])] == [t for t in ref if t != sp["eot"]], (ytoks, ykept, ok)
    eng.close()


def test_mode1_language_detection_does_not_depend_on_call_history(oracle):
    """`crispy_asr_detect_language_device` chose the f16 / f32 self K|V form by what the PREVIOUS decode call
    had left in the handle.  A fresh handle, the same handle after a long decode, and the f16 oracle must agree."""
    import torch
    from crispy_amd import synth_audio
    from crispy_amd.asr import WhisperModel
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    from oracle import whisper_oracle as WO
    hp = HParams.tiny()
    W = synthetic_whisper_weights(hp, 0, sensitive=True)
    sp = WO.special_tokens(hp.n_vocab)
    clips = [synth_audio.clip16k_np(330 + i, 100000 + 60000 * i) for i in range(4)]
    m = WhisperModel(hp, W)
    m.set_precision(1)
    d_enc = torch.from_numpy(m.encode(clips)).cuda()
    torch.cuda.synchronize()
    fresh = m.detect_language_device(d_enc.data_ptr(), 4)
    m.decode_greedy_device(d_enc.data_ptr(), 4, [sp["sot"], sp["lang0"], sp["transcribe"], sp["not_"]], 300)   # 304 keys: the 512 class
    after = m.detect_language_device(d_enc.data_ptr(), 4)
    assert np.array_equal(fresh, after), (fresh, after)
    F = whisper_mel_filters(80)
    resolved = 0
    for b, c in enumerate(clips):
        enc16 = WO.encoder_forward_f16(W, hp, oracle.oracle_logmel(c, F))
        lg = WO.DecoderCache(W, hp, enc16, f16=True).step(sp["sot"])
        lang = lg[sp["lang0"]:sp["lang0"] + sp["n_lang"]]
        top2 = np.sort(lang)[-2:]
        if top2[1] - top2[0] > MODE1_REL * np.abs(lg).max():
            assert fresh[b] == sp["lang0"] + int(np.argmax(lang)), (b, fresh[b])
            resolved += 1
    assert resolved >= 2, resolved
    m.close()


def test_mode2_transcribe_tokens_through_the_product_call(oracle):
    """Precision mode 2 (mode 1 + ggml's remaining rounding points: f16 LayerNorm outputs in the decoder, q and the
    normalised probabilities rounded inside every attention; opt-in) through `crispy_asr_transcribe_tokens`:
    deterministic, batch == solo bit for bit, the decoder half against its own oracle (DecoderCache(f16=True, ln16=True,
    attn16=True)) on the product's encoder output, the chain (encoder_forward_f16(attn16=True)) at the amplified bar."""
    from crispy_amd import synth_audio
    from crispy_amd.asr import WhisperModel
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    from oracle import whisper_oracle as WO
    hp = HParams.tiny()
    W = synthetic_whisper_weights(hp, 0, sensitive=True)
    m = WhisperModel(hp, W)
    m.set_precision(2)
    clips = [synth_audio.clip16k_np(s, n) for s, n in ((301, 130000),)]
    prompt = WO.default_prompt(hp.n_vocab, no_timestamps=True)
    toks, _ = m.transcribe_tokens(clips + [synth_audio.clip16k_np(302, 52000)], prompt, 8)      # decoded in a batch of two ...
    again, _ = m.transcribe_tokens(clips + [synth_audio.clip16k_np(302, 52000)], prompt, 8)
    assert np.array_equal(toks, again)
    F = whisper_mel_filters(hp.n_mels)
    sp = WO.special_tokens(hp.n_vocab)
    for b, c in enumerate(clips):
        solo, _ = m.transcribe_tokens([c], prompt, 8)
        assert np.array_equal(solo[0], toks[b])
        enc_gpu = m.encode([c])[0].astype(np.float64)
        enc16 = WO.encoder_forward_f16(W, hp, oracle.oracle_logmel(c, F), attn16=True)
        for enc, rel, need, what in ((enc_gpu, MODE1_REL, 4, "decoder"), (enc16, 0.03, 1, "chain")):
            dc = WO.DecoderCache(W, hp, enc, f16=True, ln16=True, attn16=True)
            for t in prompt[:-1]:
                dc.step(t)
            tok, compared = prompt[-1], 0
            for i, g in enumerate(toks[b]):
                g = int(g)
                lg = dc.step(tok)
                thr = rel * float(np.abs(lg).max())
                best = int(np.argmax(lg))
                top2 = np.partition(lg, -2)[-2:]
                assert float(lg[best] - lg[g]) <= thr, (what, b, i, g, best)
                if top2[1] - top2[0] > thr:
                    assert g == best, (what, b, i)
                    compared += 1
                tok = g
                if g == sp["eot"]:
                    break
            assert compared >= need, (what, b, compared)
    m.close()


def test_mode1_product_call_equals_the_chained_oracle_on_plain_weights_at_the_strict_bar(oracle, tmp_path_factory):
    """The whole product call against the whole oracle chain at the STRICT bar.  The audio-sensitive
    weights of the tests above amplify the encoder's 3e-4 into logit-gap changes of 1e-2, so there the chain is only held
    to 0.03 of the logit scale; plain fan-in-scaled weights do not amplify, and `crispy_asr_transcribe_batch` (mode 1,
    whisper.cpp's default options apart from the temperature ladder: timestamp tokens, seek loop, previous-text
    conditioning, 24 tokens per window) must then return the ids of oracle log-mel -> `encoder_forward_f16` ->
    `DecoderCache(f16=True)` -> `whisper_full` EXACTLY, window for window -- four clips, >= 5 windows, >= 96 tokens,
    every pick's oracle margin above MODE1_REL x the logit scale (asserted: a margin-gated comparison that compares
    nothing proves nothing) -- with the segment times and the per-window statistics."""
    from crispy_amd import synth_audio
    from crispy_amd.asr import WhisperEngine, transcribe_batch
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    from oracle import whisper_oracle as WO
    hp = HParams.tiny()
    W = synthetic_whisper_weights(hp, 1)                           # plain: no sharpened cross-attention
    eng = WhisperEngine(_engine_file(tmp_path_factory, hp, W, "tiny-s1-plain"))
    eng.set_precision(1)
    sp, sup, sup_first = _wcpp_masks(hp)
    F = whisper_mel_filters(hp.n_mels)
    prompt = [sp["sot"], sp["lang0"], sp["transcribe"]]
    clips = [synth_audio.clip16k_np(s, 16000 * 28) for s in (95, 97, 91, 93)]
    got = transcribe_batch(eng, clips, max_new_tokens=24, language_token=sp["lang0"], timestamps=True, with_segments=True,
                           fallback=False)
    n_win = n_tok = 0
    for c, x in enumerate(clips):
        encs = {}                                                   # the oracle's encoder output per window start (window 0 is used twice)

        def mel_window(seek):
            encs["cur"] = seek
            return oracle.oracle_logmel(x, F, seek)

        def encoder(mel):
            if encs["cur"] not in encs:
                encs[encs["cur"]] = WO.encoder_forward_f16(W, hp, mel)
            return encs[encs["cur"]]

        rsegs, rkept, wins = WO.transcribe_timestamps(W, hp, mel_window, x.size, prompt, WO.RULES_WCPP, eng.token_text, n_max=24,
                                                      suppress=sup, suppress_first=sup_first, f16=True, encoder=encoder)
        # the bar: MODE1_REL of the logit scale of this model (the largest |logit| the oracle saw at a pick)
        dc = WO.DecoderCache(W, hp, encs[0], f16=True)
        lg = None
        for t in prompt:
            lg = dc.step(t)
        thr = MODE1_REL * float(np.abs(lg).max())
        worst = min(min(w["margins"]) for w in wins)
        assert worst > thr, (c, worst, thr, "choose another clip: an oracle pick of this one is not resolvable at the strict bar")
        text, toks, lang, segs, gw = got[c]
        assert toks == [t for t in rkept if t != sp["eot"]], (c, toks, rkept)
        assert [(round(a * 100), round(b * 100), s) for a, b, s in segs] == [(a, b, s.decode()) for a, b, s in rsegs], c
        assert len(gw) == len(wins)
        for g, w in zip(gw, wins):
            assert (g["seek"], g["seek_advance"], g["failed"], g["no_speech"]) == (w["seek"], w["seek_advance"], int(w["failed"]), int(w["is_no_speech"]))
            assert abs(g["no_speech_prob"] - w["no_speech_prob"]) <= 2e-2 * w["no_speech_prob"] + 1e-9
            if np.isfinite(w["avg_logprob"]):
                assert abs(g["avg_logprob"] - w["avg_logprob"]) <= 2 * thr, (c, g, w["avg_logprob"])
        n_win += len(wins)
        n_tok += sum(len(w["tokens"]) for w in wins)
        print(f"plain weights clip {c}: {len(wins)} windows, {sum(len(w['tokens']) for w in wins)} picks, smallest oracle margin "
              f"{worst:.4f} against a bar of {thr:.4f}")
    assert n_win >= 5 and n_tok >= 96, (n_win, n_tok)
    eng.close()


@pytest.mark.gpu
def test_mode1_encoder_does_not_depend_on_the_gemm_tile_height():
    """The f16 GEMMs of >= 1024 rows run on 256 x 128 or 192 x 128 tiles, chosen per launch by rounds x height
    (whisper_enc_f16.hip: gemm_hh).  The choice is read once per process, so each height gets a process of its own: the
    encoder output of 8 clips (12 000 rows: every tile form with ragged last row tiles) must be the same bytes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    crcs = {}
    for rows in ("192", "256", ""):
        env = dict(os.environ, B="8", PREC="1", MODEL="tiny")
        env.pop("CRISPY_ASR_TILE_ROWS", None)
        if rows:
            env["CRISPY_ASR_TILE_ROWS"] = rows
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "enc_time.py")], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if " crc " in ln][-1]
        crcs[rows or "auto"] = line.rsplit(" all ", 1)[1].strip()
    assert len(set(crcs.values())) == 1, crcs
