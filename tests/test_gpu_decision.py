"""whisper_full's decision logic on the GPU path against the oracle's restatement of it (oracle/whisper_oracle.py:
whisper_full / decode_temperature; [UPSTREAM-RECALL] whisper.cpp whisper_full_with_state) -- what makes
`crispy_asr_transcribe(opts = NULL)` return what `engine.transcribe(&audio, &TranscribeOptions::default())`
(managers/transcription.rs:183-185) returns on silent or badly decoded 30 s chunks:
  * per window: no_speech_prob, the log-probability of every pick, average log-probability, entropy;
  * the no-speech rule (no text from a window with no_speech_prob > 0.6 and avg_logprob < -1);
  * the temperature ladder with best_of sampling decoders (std::mt19937(j) + std::discrete_distribution, restated);
  * rows with prompts of different lengths decoded in lock step (previous-text conditioning in a batch).
The CPU side of the same logic, branch by branch on a scripted decoder: tests/test_oracle_whisper_full.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _wcpp_masks(hp):
    from tests.oracle_cases import wcpp_masks
    return wcpp_masks(hp)


def _engine_file(tmp_path_factory, hp, W, tag):
    from crispy_amd.ggml_io import synthetic_vocab, write_ggml
    from crispy_amd.mel_filters import whisper_mel_filters
    path = tmp_path_factory.mktemp("ggml_dec") / f"ggml-{tag}.bin"
    write_ggml(str(path), hp, W, whisper_mel_filters(hp.n_mels), synthetic_vocab(hp.n_vocab), f16=False)
    return str(path)


@pytest.fixture(scope="module")
def tiny():
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    hp = HParams.tiny()
    W = synthetic_whisper_weights(hp, 0, sensitive=True)
    W["decoder.ln.weight"] = W["decoder.ln.weight"] * np.float32(4.0)      # sharper distributions: wider sampling intervals
    return hp, W


@pytest.mark.parametrize("mode", [0, 1])
def test_window_pass_statistics_and_rows_with_different_prompts(tiny, oracle, mode):
    """`crispy_asr_decode_window_device`, greedy: (a) three rows whose prompts differ in length (bare; <|startofprev|> + 9
    tokens; + 40 tokens) decoded in ONE batch equal the rows decoded alone bit for bit -- tokens, timestamp ids,
    log-probabilities, no_speech_prob; (b) against the oracle, teacher-forced on the GPU's tokens: every pick within the
    mode's bar of the oracle's best, ids equal where the oracle's margin is resolvable, the log-probability of every pick
    within twice that bar (it is a difference of the pick's logit and a log-sum of all of them), no_speech_prob to 5e-3."""
    import torch
    from crispy_amd import synth_audio
    from crispy_amd.asr import WhisperModel
    from crispy_amd.mel_filters import whisper_mel_filters
    from oracle import whisper_oracle as WO
    hp, W = tiny
    m = WhisperModel(hp, W)
    m.set_precision(mode)
    sp, sup, sup_first = _wcpp_masks(hp)
    init = [sp["sot"], sp["lang0"], sp["transcribe"]]
    rng = np.random.default_rng(5)
    prompts = [init,
               [sp["prev"]] + rng.integers(300, 40000, 9).tolist() + init,
               [sp["prev"]] + rng.integers(300, 40000, 40).tolist() + [sp["sot"], sp["lang0"] + 3, sp["transcribe"]]]
    clips = [synth_audio.clip16k_np(70 + i, n) for i, n in enumerate((200000, 480000, 90000))]
    seek_end = [WO.n_len_org(c.size) for c in clips]
    enc = m.encode(clips)
    d_enc = torch.from_numpy(enc).cuda()
    torch.cuda.synchronize()
    n_new = 12
    toks, tids, plog, nosp, n = m.decode_window_device(d_enc.data_ptr(), prompts, n_new, seek=[0, 0, 0], seek_end=seek_end)
    for b in range(3):
        d_one = d_enc[b:b + 1].contiguous()
        torch.cuda.synchronize()
        t1, i1, p1, s1, n1 = m.decode_window_device(d_one.data_ptr(), [prompts[b]], n_new, seek=[0], seek_end=[seek_end[b]])
        assert np.array_equal(t1[0], toks[b]) and np.array_equal(i1[0], tids[b]) and n1[0] == n[b], (b, t1, toks[b])
        assert np.array_equal(p1[0], plog[b]) and s1[0] == nosp[b], (b, p1[0], plog[b])
    rel = 1e-4 if mode == 0 else 4 * 4e-4
    F = whisper_mel_filters(hp.n_mels)
    compared = 0
    for b in range(3):
        dc = WO.DecoderCache(W, hp, enc[b].astype(np.float64), f16=(mode == 1))
        lg = None
        for t in prompts[b]:
            lg = dc.step(t)
        ref_nosp = float(np.exp(WO._log_softmax(np.asarray(lg, np.float64))[sp["nosp"]]))
        # mode 0: 0.5 %.  Mode 1: a log-probability is a logit minus the row's log-sum-exp, each within the mode's logit bar
        # `rel x scale` of the oracle's (three bars' worth allowed: the probabilities compared here are ~1e-7)
        tol_p = 5e-3 if mode == 0 else 3.0 * rel * float(np.abs(lg).max())
        assert abs(nosp[b] - ref_nosp) <= tol_p * ref_nosp + 1e-9, (b, nosp[b], ref_nosp, tol_p)
        seq = []
        for i in range(int(n[b])):
            g = int(toks[b, i])
            ml, lp, tid = WO.process_logits(lg, seq, sp, WO.RULES_WCPP, sup, sup_first)
            fin = ml[np.isfinite(ml)]
            thr = rel * float(np.abs(lg).max())
            assert np.isfinite(ml[g]) and ml.max() - ml[g] <= thr, (mode, b, i, g, int(np.argmax(ml)), float(ml.max() - ml[g]), thr)
            top2 = np.partition(fin, -2)[-2:] if fin.size > 1 else np.array([-np.inf, fin[0]])
            if top2[1] - top2[0] > thr:
                assert g == int(np.argmax(ml)), (mode, b, i)
                assert abs(plog[b, i] - lp[g]) <= 2 * thr, (mode, b, i, plog[b, i], lp[g], thr)     # two logits' worth of error
                assert int(tids[b, i]) == (g if g >= sp["beg"] else tid), (mode, b, i, tids[b, i], tid)
                compared += 1
            seq.append(g)
            lg = dc.step(g)
    assert compared >= 24, compared
    print(f"mode {mode}: {compared} of {int(n.sum())} picks resolvable; no_speech_prob {nosp.tolist()}")
    m.close()


def test_sampling_pick_is_the_discrete_distribution_over_the_restated_generator(tiny, oracle):
    """The sampling pass of the ladder: five rows over the same clip and prompt (best_of = 5) at temperature 0.4, row j fed
    the variates MT19937(j) yields.  Teacher-forced oracle per row: the GPU's pick must be the index whose interval of
    the cumulative distribution contains u -- with the interval ends computed by the oracle, widened by what f32 logits
    can move them (2e-5) -- and equal to the oracle's own pick wherever u is further than that from an interval end; the
    log-probability of the pick is the log-softmax at that temperature."""
    import torch
    from crispy_amd import synth_audio
    from crispy_amd.asr import WhisperModel
    from oracle import whisper_oracle as WO
    hp, W = tiny
    m = WhisperModel(hp, W)
    sp, sup, sup_first = _wcpp_masks(hp)
    init = [sp["sot"], sp["lang0"], sp["transcribe"]]
    clip = synth_audio.clip16k_np(75, 300000)
    enc = m.encode([clip])
    rows, n_new, T = 5, 10, 0.4
    d_enc = torch.from_numpy(np.repeat(enc, rows, axis=0)).cuda()
    torch.cuda.synchronize()
    gens = [WO.MT19937(j) for j in range(rows)]
    u = np.array([[g.canonical() for g in gens] for _ in range(n_new)])        # [n_new][rows], drawn per row in step order
    gens = [WO.MT19937(j) for j in range(rows)]
    u = np.empty((n_new, rows))
    for j in range(rows):
        for i in range(n_new):
            u[i, j] = gens[j].canonical()
    toks, tids, plog, nosp, n = m.decode_window_device(d_enc.data_ptr(), [init] * rows, n_new, seek=[0] * rows,
                                                       seek_end=[WO.n_len_org(clip.size)] * rows, temperature=T, u=u)
    again = m.decode_window_device(d_enc.data_ptr(), [init] * rows, n_new, seek=[0] * rows,
                                   seek_end=[WO.n_len_org(clip.size)] * rows, temperature=T, u=u)
    assert np.array_equal(again[0], toks) and np.array_equal(again[2], plog)
    assert len({tuple(t.tolist()) for t in toks}) > 1, toks            # the decoders do not all say the same
    tol, exact, total = 2e-5, 0, 0
    for j in range(rows):
        dc = WO.DecoderCache(W, hp, enc[0].astype(np.float64))
        lg = None
        for t in init:
            lg = dc.step(t)
        seq = []
        for i in range(int(n[j])):
            g = int(toks[j, i])
            ml, lp, tid = WO.process_logits(lg, seq, sp, WO.RULES_WCPP, sup, sup_first, temperature=T)
            pr = np.where(np.isfinite(lp), np.exp(lp), 0.0)
            cp = np.cumsum(pr / pr.sum())
            lo = cp[g - 1] if g > 0 else 0.0
            assert pr[g] > 0 and lo - tol <= u[i, j] <= cp[g] + tol, (j, i, g, u[i, j], lo, cp[g])
            own, gap = WO.sample_index(pr, u[i, j])
            if gap > tol:
                assert g == own, (j, i, g, own, gap)
                exact += 1
            assert abs(plog[j, i] - lp[g]) <= 5e-4, (j, i, plog[j, i], lp[g])
            total += 1
            seq.append(g)
            lg = dc.step(g)
    assert exact >= total // 2, (exact, total)
    print(f"sampling: {exact} of {total} picks further than {tol} from an interval end, all inside their interval")
    m.close()


def _scripts(hp):
    from oracle import whisper_oracle as WO
    sp = WO.special_tokens(hp.n_vocab)
    return sp, sp["beg"], sp["eot"]


def _ref(W, hp, n_samples, eng, mode, **kw):
    """The oracle's whisper_full on a scripted model (its decoder ignores the audio: the encoder pass is skipped).  It depends on
    the seeded weights and the options only, and it is 1 300 float64 decoder steps for the ladder model: its result is committed
    (tests/oracle_cases.py, tests/oracle_cache.py, tests/golden/make_oracle_cache.py) and recomputed when the inputs change."""
    from tests.oracle_cases import scripted_ref
    return scripted_ref(W, hp, n_samples, mode, **kw)


def _same_windows(got, ref, tol_lp, what):
    assert len(got) == len(ref), (what, got, [(w["seek"], w["temperature"]) for w in ref])
    for g, w in zip(got, ref):
        assert g["seek"] == w["seek"] and g["seek_advance"] == w["seek_advance"], (what, g, w["seek"], w["seek_advance"])
        assert abs(g["temperature"] - w["temperature"]) < 1e-6 and g["decoder"] == w["decoder"], (what, g, w["temperature"], w["decoder"])
        assert g["failed"] == int(w["failed"]) and g["no_speech"] == int(w["is_no_speech"]), (what, g, w["failed"], w["is_no_speech"])
        assert g["n_tokens"] == (0 if w["is_no_speech"] else len(w["tokens"])), (what, g, w["tokens"])
        assert abs(g["no_speech_prob"] - w["no_speech_prob"]) <= 5e-3 * w["no_speech_prob"] + 1e-7, (what, g, w["no_speech_prob"])
        if np.isfinite(w["avg_logprob"]):
            assert abs(g["avg_logprob"] - w["avg_logprob"]) <= tol_lp, (what, g, w["avg_logprob"])
        else:
            assert g["avg_logprob"] == w["avg_logprob"]
        assert abs(g["entropy"] - w["entropy"]) <= 1e-5, (what, g, w["entropy"])


@pytest.mark.parametrize("mode", [0, 1])
def test_temperature_ladder_on_a_scripted_model(oracle, tmp_path_factory, mode):
    """13 s on a model whose logits follow a script by position (tests/scripted_model.py), through `crispy_asr_transcribe`.
    Bare prompt (generation starts at position 2): "<|0.00|> w1 {X | Y: a one-logit near tie} w3 <|6.00|><|6.00|> EOT".
    With the text so far in front (generation starts at position 9): <|0.00|>, one token 40 times, a timestamp pair -- the
    entropy check fails it.  So: window 1 passes greedily; window 2 fails at temperatures 0, 0.2, 0.4 (conditioned on
    the past), and at 0.6 -- where whisper_full drops the past -- five sampling decoders run the bare-prompt script and
    split between X and Y as their generators say; the best-scoring one is accepted.  The last window starts with under
    5 s left: bare prompt, greedy, closed by its first timestamp, a single-timestamp ending.
    Product == oracle: tokens, segments, and per window temperature, winning decoder, statistics.  Then a model that
    repeats itself on the bare prompt too: every pass fails, the pass at 1.0 is accepted with its failure recorded."""
    from crispy_amd import synth_audio
    from crispy_amd.asr import WhisperEngine, transcribe_batch
    from crispy_amd.whisper_weights import HParams
    from tests.scripted_model import script_rows, scripted_whisper_weights
    hp = HParams.tiny()
    sp, BEG, EOT = _scripts(hp)
    from tests import oracle_cases as OC
    X, Y, REP = OC.X_TOK, OC.Y_TOK, OC.REP_TOK
    W = OC.ladder_model(hp)
    eng = WhisperEngine(_engine_file(tmp_path_factory, hp, W, f"ladder{mode}"))
    eng.set_precision(mode)
    x = synth_audio.clip16k_np(80, 16000 * 13)
    text, segs, toks = eng.transcribe_segments(x, language_token=sp["lang0"])
    wins = eng.last_windows
    rsegs, rkept, rwins = _ref(W, hp, x.size, eng, mode)
    assert [w["temperature"] for w in rwins] == pytest.approx([0.0, 0.6, 0.0]) and [w["seek"] for w in rwins] == [0, 600, 1200]
    # the oracle's sampled picks are resolvable: u is nowhere near an interval end
    assert min(min(d["margins"]) for w in rwins for it in w["iterations"] if it["temperature"] > 0 for d in it["decoders"]) > 1e-4
    assert len({tuple(d["toks"]) for d in rwins[1]["iterations"][-1]["decoders"]}) == 2      # the five decoders split X / Y
    assert toks == [t for t in rkept if t != EOT], (toks, rkept)
    assert [(round(a * 100), round(b * 100), s) for a, b, s in segs] == [(a, b, s.decode()) for a, b, s in rsegs]
    _same_windows(wins, rwins, 1e-4 if mode == 0 else 5e-3, f"ladder mode {mode}")
    # the same clip inside a batch (its neighbours fall back at other moments)
    got = transcribe_batch(eng, [x[:16000 * 7], x, x[:16000 * 3]], language_token=sp["lang0"], timestamps=True, with_segments=True)
    assert got[1][:4] == (text, toks, sp["lang0"], segs) and got[1][4] == wins
    eng.close()
    if mode == 1:
        return        # the rest is decision logic that does not depend on the arithmetic: mode 0 runs it (the oracle's 1 300
                      # float64 decoder steps of it are 25 s of a GPU suite with a time limit)
    # ... and a model that repeats itself whatever the prompt
    W = OC.repeat_model(hp)
    eng = WhisperEngine(_engine_file(tmp_path_factory, hp, W, f"repeat{mode}"))
    eng.set_precision(mode)
    x = synth_audio.clip16k_np(81, 16000 * 4)
    # (best_of = 2: the ladder's shape does not depend on how many decoders fail per pass, the oracle's time does -- 11 passes
    # of 43 float64 decoder steps per window instead of 26)
    text, segs, toks = eng.transcribe_segments(x, language_token=sp["lang0"], best_of=2)
    rsegs, rkept, rwins = _ref(W, hp, x.size, eng, mode, params=dict(best_of=2))
    assert all(w["failed"] and w["temperature"] == pytest.approx(1.0) and len(w["iterations"]) == 6 for w in rwins)
    assert all(len(it["decoders"]) == (2 if it["temperature"] > 0 else 1) for w in rwins for it in w["iterations"])
    assert toks == [t for t in rkept if t != EOT] and len(rwins[0]["tokens"]) == 43
    assert [(round(a * 100), round(b * 100), s) for a, b, s in segs] == [(a, b, s.decode()) for a, b, s in rsegs]
    _same_windows(eng.last_windows, rwins, 1e-4 if mode == 0 else 5e-3, f"repeat mode {mode}")
    # no fallback (temperature_inc < 0): the greedy pass is accepted as it is, failure recorded
    _, _, toks0 = eng.transcribe_segments(x, language_token=sp["lang0"], fallback=False)
    assert toks0 == toks and eng.last_windows[0]["temperature"] == 0.0 and eng.last_windows[0]["failed"] == 1
    # entropy check off: passes at temperature 0
    eng.transcribe_segments(x, language_token=sp["lang0"], entropy_thold=-1.0)
    assert eng.last_windows[0]["failed"] == 0 and eng.last_windows[0]["temperature"] == 0.0
    eng.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_silent_and_noisy_chunks_give_no_text_under_the_no_speech_rule(oracle, tmp_path_factory, mode):
    """A model that is sure of <|nospeech|> at the start of every window and unsure of everything it then says (flat
    logits: log-probability ~ -7 per token) -- what Whisper does on silence and on noise.  `crispy_asr_transcribe` with
    opts = NULL on 30 s of digital silence and on 30 s of noise: no text, no tokens, no segments; every window recorded
    as dropped, with the oracle's no_speech_prob and average log-probability; nothing re-decoded (the fallback only
    takes windows whose no_speech_prob is BELOW the threshold).  With no_speech_thold = 1 the same windows give text."""
    import ctypes as C
    from crispy_amd import _native as N
    from crispy_amd.asr import WhisperEngine, _read_result
    from crispy_amd.whisper_weights import HParams
    from tests.scripted_model import script_rows, scripted_whisper_weights
    hp = HParams.tiny()
    sp, BEG, EOT = _scripts(hp)
    from tests import oracle_cases as OC
    W = OC.nospeech_model(hp)
    eng = WhisperEngine(_engine_file(tmp_path_factory, hp, W, f"nosp{mode}"))
    eng.set_precision(mode)
    rng = np.random.default_rng(9)
    for name, x in (("silence", np.zeros(480000, np.float32)), ("noise", (0.1 * rng.standard_normal(480000)).astype(np.float32))):
        res = C.c_void_p()
        N.check(N.lib().crispy_asr_transcribe(eng._h, x.ctypes.data, x.size, None, C.byref(res)))
        try:
            text, tokens, lang, segs, wins = _read_result(res)
        finally:
            N.lib().crispy_asr_free_result(res)
        assert text == "" and tokens == [] and segs == [], (name, text, tokens)
        rsegs, rkept, rwins = _ref(W, hp, x.size, eng, mode)
        assert rsegs == [] and rkept == [] and len(rwins) == 2 and all(w["is_no_speech"] for w in rwins)
        assert all(len(w["iterations"]) == 1 for w in rwins)
        assert all(w["no_speech"] == 1 and w["no_speech_prob"] > 0.6 and w["avg_logprob"] < -1.0 for w in wins), wins
        _same_windows(wins, rwins, 1e-3 if mode == 0 else 2e-2, f"{name} mode {mode}")
    text, segs, toks = eng.transcribe_segments(x, language_token=sp["lang0"], no_speech_thold=1.0, fallback=False)
    assert toks[:4] == [BEG, 1001, BEG + 1400, BEG + 1400] and len(segs) >= 1 and text != ""
    eng.close()


@pytest.mark.parametrize("best_of,opts", [(5, {}), (1, {"logprob_thold": 1.0}), (3, {"temperature": 0.2})])
def test_fallback_passes_of_a_batch_run_side_by_side_and_equal_the_single_calls(tmp_path_factory, best_of, opts):
    """`crispy_asr_transcribe_batch` decodes the fallback passes of ALL failed clips of a round together -- rows = clips x
    best_of over one cross K|V per clip -- and every clip must still come out as from a call of its own,
    bit for bit: tokens, segments, and per window the temperature it was accepted at, the winning decoder and the
    statistics.  8 clips of the ladder model of the test above, cut to lengths that put their windows in different
    situations (one window; a second window conditioned on the text so far, which the entropy check fails up to 0.4
    and five sampling decoders rescue at 0.6; a second window with under 5 s left, decoded bare; a third, single-timestamp
    window), so each round has clips at temperature 0 next to clips in the ladder.  best_of = 1 with a log-probability
    bar no sequence can pass (the rows of such a pass are different CLIPS, each drawing from its own
    generator 0 -- the variates used to come from the first clip's generators, read past their end) and a call that
    starts above temperature 0 (sampling from the first pass on) are the other two cases."""
    from crispy_amd import synth_audio
    from crispy_amd.asr import WhisperEngine, transcribe_batch
    from crispy_amd.whisper_weights import HParams
    from tests.scripted_model import script_rows, scripted_whisper_weights
    hp = HParams.tiny()
    sp, BEG, EOT = _scripts(hp)
    X, Y, REP = 1234, 2345, 777
    beta = 1.0 - 1.0 * np.sqrt(2.0) / hp.n_text_state
    rows = script_rows(2, [BEG, 1001, [(X, 1.0), (Y, beta)], 1003, BEG + 300, BEG + 300, EOT])
    rows.update(script_rows(9, [BEG] + [REP] * 40 + [BEG + 100, BEG + 100, EOT]))
    W = scripted_whisper_weights(hp, rows, gain=100.0)
    eng = WhisperEngine(_engine_file(tmp_path_factory, hp, W, f"ladder-batch{best_of}"))
    eng.set_precision(1)
    base = synth_audio.clip16k_np(80, 16000 * 13)
    clips = [base[:16000 * n] for n in (13, 7, 12, 3, 13, 10, 5, 12)]
    kw = dict(language_token=sp["lang0"], timestamps=True, with_segments=True, best_of=best_of, **opts)
    got = transcribe_batch(eng, clips, **kw)
    again = transcribe_batch(eng, clips, **kw)
    assert got == again                                            # the generators are re-seeded per call
    temps = set()
    for c, x in enumerate(clips):
        solo = transcribe_batch(eng, [x], **kw)[0]
        assert got[c] == solo, (c, got[c][4], solo[4])
        temps |= {round(w["temperature"], 1) for w in solo[4]}
    print(f"best_of {best_of} {opts}: windows accepted at temperatures {sorted(temps)}")
    assert len(temps) >= (2 if best_of == 5 else 1), temps
    # a subset in another order: a clip's result does not depend on its neighbours or its place
    sub = transcribe_batch(eng, [clips[4], clips[1], clips[7]], **kw)
    assert sub == [got[4], got[1], got[7]]
    eng.close()


@pytest.mark.parametrize("opts", [dict(best_of=5), dict(beam_size=3, best_of=4)], ids=["best_of_5", "beam_3_best_of_4"])
def test_a_clip_of_a_large_batch_of_fallback_rows_decodes_as_alone(tmp_path_factory, opts):
    """A fallback / beam pass of many clips -- 120 clips of the ladder model (fifteen of each of the eight lengths of the test
    above; half of them have a window that walks the ladder together: 300 rows of best-of decoders in one group since the
    groups went from 128 to 512 rows) -- against single calls: a clip's result does not depend on how many rows its pass
    holds.  (Written for a cross block that shared a clip's K | V between its rows, NOTEBOOK 10.9; kept for the group size.)"""
    from crispy_amd import synth_audio
    from crispy_amd.asr import WhisperEngine, transcribe_batch
    from crispy_amd.whisper_weights import HParams
    from tests.scripted_model import script_rows, scripted_whisper_weights
    hp = HParams.tiny()
    sp, BEG, EOT = _scripts(hp)
    X, Y, REP = 1234, 2345, 777
    beta = 1.0 - 1.0 * np.sqrt(2.0) / hp.n_text_state
    rows = script_rows(2, [BEG, 1001, [(X, 1.0), (Y, beta)], 1003, BEG + 300, BEG + 300, EOT])
    rows.update(script_rows(9, [BEG] + [REP] * 40 + [BEG + 100, BEG + 100, EOT]))
    W = scripted_whisper_weights(hp, rows, gain=100.0)
    eng = WhisperEngine(_engine_file(tmp_path_factory, hp, W, "ladder-rows"))
    eng.set_precision(1)
    base = synth_audio.clip16k_np(80, 16000 * 13)
    lengths = (13, 7, 12, 3, 13, 10, 5, 12)
    clips = [base[:16000 * lengths[i % 8]] for i in range(120)]
    kw = dict(language_token=sp["lang0"], timestamps=True, with_segments=True, **opts)
    got = transcribe_batch(eng, clips, **kw)
    laddered = sum(any(w["temperature"] > 0 for w in r[4]) for r in got)
    print(f"{opts}: {laddered} of {len(clips)} clips walked the ladder")
    assert laddered >= 43, laddered                                 # more than 128 rows of best-of decoders in one pass
    for c in range(8):
        solo = transcribe_batch(eng, [clips[c]], **kw)[0]
        assert got[c] == solo, (c, got[c][4], solo[4])
        assert all(got[c + 8 * k] == solo for k in range(1, 15)), c
    eng.close()


def test_fallback_groups_that_grow_between_passes_stay_inside_their_buffers(tmp_path_factory):
    """the gathered encoder outputs of a group of fallback clips were allocated once per call, sized by the
    FIRST group that needed them -- and the group size changes from pass to pass: 128 / beam clips at temperature 0, 128 /
    best_of above.  21 clips of the ladder model, beam 8: clip 0 (7 s) passes its second window, clips 1 .. 20 (13 s) fail
    theirs at temperature 0 in groups of 16 + 4 (not a prefix of the active clips: gathered copies, buffer of 16 clips), then
    decode again at 0.2 .. 0.6 with best_of = 5 as ONE group of 20 clips -- 4 clips past the end of the round-5 buffer (9 MB
    over whatever lay behind it).  Every clip must equal its single call, as in the test above."""
    from crispy_amd import synth_audio
    from crispy_amd.asr import WhisperEngine, transcribe_batch
    from crispy_amd.whisper_weights import HParams
    from tests.scripted_model import script_rows, scripted_whisper_weights
    hp = HParams.tiny()
    sp, BEG, EOT = _scripts(hp)
    X, Y, REP = 1234, 2345, 777
    beta = 1.0 - 1.0 * np.sqrt(2.0) / hp.n_text_state
    rows = script_rows(2, [BEG, 1001, [(X, 1.0), (Y, beta)], 1003, BEG + 300, BEG + 300, EOT])
    rows.update(script_rows(9, [BEG] + [REP] * 40 + [BEG + 100, BEG + 100, EOT]))
    W = scripted_whisper_weights(hp, rows, gain=100.0)
    eng = WhisperEngine(_engine_file(tmp_path_factory, hp, W, "ladder-grow"))
    eng.set_precision(1)
    base = synth_audio.clip16k_np(80, 16000 * 13)
    clips = [base[:16000 * 7]] + [base[:16000 * 13 - 160 * i] for i in range(20)]
    kw = dict(language_token=sp["lang0"], timestamps=True, with_segments=True, beam_size=8)
    got = transcribe_batch(eng, clips, **kw)
    temps = {round(w["temperature"], 1) for r in got for w in r[4]}
    assert 0.0 in temps and max(temps) >= 0.2, temps             # windows accepted at temperature 0 and in the ladder
    for c in (0, 1, 17, 20):
        solo = transcribe_batch(eng, [clips[c]], **kw)[0]
        assert got[c] == solo, (c, got[c][4], solo[4])
    # the same with one sampling decoder and a log-probability bar nothing passes: groups of ALL pending clips
    kw1 = dict(language_token=sp["lang0"], timestamps=True, with_segments=True, best_of=1)
    got1 = transcribe_batch(eng, clips, **kw1)
    for c in (0, 2, 20):
        assert got1[c] == transcribe_batch(eng, [clips[c]], **kw1)[0], c
    eng.close()


def test_non_speech_suppression_initial_prompt_and_carried_context(oracle, tmp_path_factory):
    """The whisper_full parameters a host may set beyond the defaults (crispy_asr_opts, ABI 3), product
    against oracle in precision mode 1:
    (a) suppress_nst on a scripted model whose vocabulary holds "(" and " -": the script's favourites at two positions are
        those tokens -- picked without the option, the runners-up with it; the mask is exactly the oracle's id set;
    (b) an initial prompt and, from the second call on, the context carried over from the call before (no_context = false),
        on plain weights: three consecutive single-chunk calls equal the oracle's whisper_full chain, token for token;
    (c) what is out of range says so: beam_size beyond 8 decoders, beam search without timestamps, carry_context in a batch call."""
    from crispy_amd import _native as N, synth_audio
    from crispy_amd.asr import WhisperEngine, transcribe_batch
    from crispy_amd.ggml_io import synthetic_vocab, write_ggml
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    from oracle import whisper_oracle as WO
    from tests.scripted_model import script_rows, scripted_whisper_weights
    hp = HParams.tiny()
    sp, BEG, EOT = _scripts(hp)
    _, sup, sup_first = _wcpp_masks(hp)
    F = whisper_mel_filters(80)
    # ---- (a)
    PAREN, DASH, A, B = 1001, 1003, 2345, 3456
    beta = 1.0 - 1.0 * np.sqrt(2.0) / hp.n_text_state            # the favourite leads the runner-up by one logit at gain 100
    rows = script_rows(2, [BEG, [(PAREN, 1.0), (A, beta)], [(DASH, 1.0), (B, beta)], BEG + 300, BEG + 300, EOT])
    W = scripted_whisper_weights(hp, rows, gain=100.0)
    vocab = synthetic_vocab(hp.n_vocab)
    vocab[PAREN], vocab[DASH], vocab[4000], vocab[4001] = b"(", b" -", b"-", "♪".encode()
    path = str(tmp_path_factory.mktemp("ggml_nst") / "nst.bin")
    write_ggml(path, hp, W, F, vocab, f16=False)
    eng = WhisperEngine(path)
    eng.set_precision(1)
    x = synth_audio.clip16k_np(80, 16000 * 7)
    nst = WO.non_speech_token_ids(vocab)
    assert nst == [PAREN, DASH, 4001]
    _, _, plain = eng.transcribe_segments(x, language_token=sp["lang0"], fallback=False)
    _, _, masked = eng.transcribe_segments(x, language_token=sp["lang0"], fallback=False, suppress_nst=True)
    assert plain[:5] == [BEG, PAREN, DASH, BEG + 300, BEG + 300] and masked[:5] == [BEG, A, B, BEG + 300, BEG + 300]
    enc0 = np.zeros((hp.n_audio_ctx, hp.n_audio_state))
    for got, extra in ((plain, []), (masked, nst)):
        _, rk, _ = WO.transcribe_timestamps(W, hp, lambda seek: None, x.size, [sp["sot"], sp["lang0"], sp["transcribe"]], WO.RULES_WCPP,
                                            eng.token_text, suppress=sorted(sup + extra), suppress_first=sup_first, f16=True,
                                            encoder=lambda mel: enc0)
        assert got == [t for t in rk if t != EOT], (got, rk)
    # ---- (c)
    for bad, code in ((dict(beam_size=9), -1), (dict(beam_size=-1), -1)):        # at most WHISPER_MAX_DECODERS = 8 (beam search itself: the test below)
        with pytest.raises(N.CrispyError) as e:
            eng.transcribe_segments(x, language_token=sp["lang0"], **bad)
        assert e.value.code == code
    with pytest.raises(N.CrispyError) as e:                                         # beam search lives in whisper_full's window loop
        eng.transcribe(x, language_token=sp["lang0"], timestamps=False, beam_size=2)
    assert e.value.code == -6
    with pytest.raises(N.CrispyError) as e:
        transcribe_batch(eng, [x, x], language_token=sp["lang0"], timestamps=True, carry_context=True)
    assert e.value.code == -1
    eng.transcribe_segments(x, language_token=sp["lang0"], beam_size=1, fallback=False)       # 0 and 1 are the greedy strategy
    eng.close()
    # ---- (b)
    Wp = synthetic_whisper_weights(hp, 1)                          # plain fan-in-scaled weights: the strict bar holds (test_gpu_mode1)
    eng = WhisperEngine(_engine_file(tmp_path_factory, hp, Wp, "tiny-s1-ctx"))
    eng.set_precision(1)
    init = [1000, 2000, 3000, 4000, 5000]
    prompt = [sp["sot"], sp["lang0"], sp["transcribe"]]
    chunks = [synth_audio.clip16k_np(s, 16000 * 9) for s in (95, 97, 91)]
    past, thr_seen = [], []
    memo = {}                                                      # the oracle's encoder output per (chunk, window start)

    def oracle_io(i, c):
        cur = {}

        def mel_window(seek):
            cur["seek"] = seek
            return oracle.oracle_logmel(c, F, seek)

        def encoder(mel):
            key = (i, cur["seek"])
            if key not in memo:
                memo[key] = WO.encoder_forward_f16(Wp, hp, mel)
            return memo[key]
        return mel_window, encoder

    for i, c in enumerate(chunks):
        _, _, toks = eng.transcribe_segments(c, max_new_tokens=12, language_token=sp["lang0"], fallback=False, initial_prompt=init,
                                             carry_context=(i > 0))
        st = {}
        mw, en = oracle_io(i, c)
        _, rk, wins = WO.transcribe_timestamps(Wp, hp, mw, c.size, prompt, WO.RULES_WCPP,
                                               eng.token_text, n_max=12, suppress=sup, suppress_first=sup_first, f16=True,
                                               initial_prompt=init, past0=past, state=st, encoder=en)
        assert wins[0]["prompt"][:1 + len(init)] == [sp["prev"]] + init
        if i > 0:
            assert len(wins[0]["prompt"]) > 1 + len(init) + 3          # the call before left text behind
        thr_seen.append(min(min(w["margins"]) for w in wins))
        assert toks == [t for t in rk if t != EOT], (i, toks, rk, thr_seen)
        past = st["prompt_past"]
    print(f"carried context: smallest oracle margins per call {thr_seen}")
    # without carry_context the second chunk starts clean again: another transcript than with it
    _, _, clean = eng.transcribe_segments(chunks[1], max_new_tokens=12, language_token=sp["lang0"], fallback=False, initial_prompt=init)
    mw, en = oracle_io(1, chunks[1])
    _, rk, wins = WO.transcribe_timestamps(Wp, hp, mw, chunks[1].size, prompt,
                                           WO.RULES_WCPP, eng.token_text, n_max=12, suppress=sup, suppress_first=sup_first, f16=True,
                                           initial_prompt=init, encoder=en)
    assert clean == [t for t in rk if t != EOT] and len(wins[0]["prompt"]) == 1 + len(init) + 3
    eng.close()


def test_beam_search_on_a_scripted_model(oracle, tmp_path_factory):
    """whisper.cpp's BEAM_SEARCH strategy (crispy_asr_opts.beam_size = 3) through `crispy_asr_transcribe`
    on the ladder model: per step every live decoder draws three ids from its distribution (the device pick kernel in its
    candidate form, variates from the decoder's own MT19937), the host sorts the clip's candidates by the sum of all their
    log-probabilities and deals them out without repeating a sequence, and the self K|V rows follow the sequences on the
    device.  Window 1 (bare prompt): the three beams hold the X and the Y reading of the near tie, the X reading wins on its
    score; window 2 (conditioned on the text so far: one token 40 times) fails the entropy check at temperatures 0 .. 0.4
    -- beam passes with best_of = 5 decoders above 0 -- and passes at 0.6 without the past; the last window is short.
    Product == oracle: tokens, segments, per window the temperature, the winning decoder and the statistics -- which also
    pins how many variates every generator has drawn by then.  Then the same clip inside a batch, and mode 1 == itself
    across batch compositions."""
    from crispy_amd import synth_audio
    from crispy_amd.asr import WhisperEngine, transcribe_batch
    from crispy_amd.whisper_weights import HParams
    from tests.scripted_model import script_rows, scripted_whisper_weights
    hp = HParams.tiny()
    sp, BEG, EOT = _scripts(hp)
    from tests import oracle_cases as OC
    X, Y, REP = OC.X_TOK, OC.Y_TOK, OC.REP_TOK
    W = OC.ladder_model(hp)
    eng = WhisperEngine(_engine_file(tmp_path_factory, hp, W, "beam"))
    x = synth_audio.clip16k_np(80, 16000 * 13)
    text, segs, toks = eng.transcribe_segments(x, language_token=sp["lang0"], beam_size=3)
    wins = eng.last_windows
    rsegs, rkept, rwins = _ref(W, hp, x.size, eng, 0, params=dict(beam_size=3))
    first = rwins[0]["iterations"][0]["decoders"]
    assert len(first) == 3 and {tuple(d["toks"][:3]) for d in first} == {(BEG, 1001, X), (BEG, 1001, Y)}      # both readings are held
    assert min(min(d["margins"]) for w in rwins for it in w["iterations"] for d in it["decoders"]) > 1e-4       # the draws are resolvable
    assert any(it["temperature"] > 0 and len(it["decoders"]) == 5 for w in rwins for it in w["iterations"])
    assert toks == [t for t in rkept if t != EOT], (toks, rkept)
    assert [(round(a * 100), round(b * 100), s) for a, b, s in segs] == [(a, b, s.decode()) for a, b, s in rsegs]
    _same_windows(wins, rwins, 1e-4, "beam")
    kw = dict(language_token=sp["lang0"], timestamps=True, with_segments=True, beam_size=3)
    got = transcribe_batch(eng, [x[:16000 * 7], x, x[:16000 * 3]], **kw)
    assert got[1][:4] == (text, toks, sp["lang0"], segs) and got[1][4] == wins
    # greedy on the same clip takes another road through the ladder but ends with the same text (the X reading)
    _, _, toks_g = eng.transcribe_segments(x, language_token=sp["lang0"])
    assert toks_g == toks
    eng.set_precision(1)
    solo = transcribe_batch(eng, [x], **kw)[0]
    both = transcribe_batch(eng, [x[:16000 * 5], x], **kw)
    assert both[1] == solo and any(w["temperature"] > 0 for w in solo[4])
    eng.close()
