"""whisper_full's decision logic in the oracle (oracle/whisper_oracle.py: whisper_full, decode_temperature,
sequence_score, MT19937, sample_index) [UPSTREAM-RECALL: whisper.cpp whisper_full_with_state], driven by a SCRIPTED
decoder in place of the network so that every branch is reached on purpose: the seek loop and previous-text conditioning,
the single-timestamp ending, the no-speech rule, the three ways a decoder fails, the temperature ladder with its
best-of sampling decoders, and the arithmetic of the statistics.  CPU only; the GPU side of the same behaviour is
tests/test_gpu_decision.py."""
from __future__ import annotations

import math

import numpy as np
import pytest

from crispy_amd.whisper_weights import HParams
from oracle import whisper_oracle as WO

HP = HParams.tiny()
SP = WO.special_tokens(HP.n_vocab)
BEG, EOT, V = SP["beg"], SP["eot"], HP.n_vocab
INIT = [SP["sot"], SP["lang0"], SP["transcribe"]]
SUP = [SP["sot"], SP["nosp"], SP["translate"], SP["transcribe"], SP["prev"], SP["solm"]] + list(range(SP["lang0"], SP["lang0"] + 99))
SUP_FIRST = [220, EOT]


class Scripted:
    """Stands in for DecoderCache: the logits of the next position are `script(generated tokens, prompt)`."""

    def __init__(self, script, hist=None):
        self.script, self.hist = script, list(hist or [])

    def step(self, tok):
        self.hist.append(int(tok))
        if SP["transcribe"] not in self.hist:
            return np.zeros(V)                               # inside the prompt: these logits are never looked at
        cut = len(self.hist) - 1 - self.hist[::-1].index(SP["transcribe"])      # last <|transcribe|>: end of the prompt
        return self.script(self.hist[cut + 1:], self.hist[:cut + 1])

    def fork(self):
        return Scripted(self.script, self.hist)


def peaky(tok, height=30.0, nosp=0.0):
    lg = np.zeros(V)
    lg[tok] = height
    lg[SP["nosp"]] = nosp
    return lg


def flat(tok, nosp=0.0):
    """The pick wins by a hair: log-probability ~ -log(n_vocab).  Timestamps pushed down so that their summed mass does
    not force one, unless a timestamp is the pick."""
    lg = np.zeros(V)
    lg[BEG:] = -20.0
    lg[tok] = 0.5
    lg[SP["nosp"]] = nosp
    return lg


def by_list(seq, make=peaky):
    """Script that walks `seq` (then EOT forever)."""
    return lambda gen, prompt: make(seq[len(gen)] if len(gen) < len(seq) else EOT)


@pytest.fixture
def scripted(monkeypatch):
    made = []

    def install(factory):
        """factory(index of the decoder construction, i.e. of the (window, temperature) pass) -> script"""
        def ctor(weights, hp, enc, f16=False, **kw):
            made.append(len(made))
            return Scripted(factory(len(made) - 1))
        monkeypatch.setattr(WO, "DecoderCache", ctor)
        monkeypatch.setattr(WO, "encoder_forward", lambda *a, **k: None)
        return made
    return install


def run(n_samples, **kw):
    kw.setdefault("suppress", SUP)
    kw.setdefault("suppress_first", SUP_FIRST)
    return WO.transcribe_timestamps(None, HP, lambda seek: None, n_samples, INIT, WO.RULES_WCPP, lambda t: b" w%d" % t, **kw)


def test_mt19937_and_the_uniform_variate():
    g = WO.MT19937()
    for _ in range(9999):
        g.next_u32()
    assert g.next_u32() == 4123659995                       # the C++ standard's check value for std::mt19937
    raw = np.random.MT19937()
    raw._legacy_seeding(3)                                  # init_genrand(3), as std::mt19937(3)
    assert [WO.MT19937(3).next_u32() for _ in range(1)] == [int(raw.random_raw(1)[0])]
    a, b = WO.MT19937(7), WO.MT19937(7)
    x0, x1 = b.next_u32(), b.next_u32()
    u = a.canonical()
    assert u == (x0 + x1 * 2.0 ** 32) / 2.0 ** 64 and 0.0 <= u < 1.0
    assert a.next_u32() == b.next_u32()                     # exactly two draws per variate


def test_discrete_distribution_picks_the_first_index_whose_cumulative_share_reaches_u():
    p = [0.0, 0.25, 0.25, 0.0, 0.5]
    pick = lambda u: WO.sample_index(np.array(p) * 3.0, u)[0]      # unnormalised on purpose
    assert [pick(u) for u in (1e-9, 0.1, 0.25, 0.2500001, 0.5, 0.5000001, 0.999999)] == [1, 1, 1, 2, 2, 4, 4]
    i, gap = WO.sample_index(p, 0.3)
    assert i == 2 and abs(gap - 0.05) < 1e-12


def test_sequence_score_arithmetic():
    sc = WO.sequence_score([1, 1, 2, 2], [-1.0, -2.0, -3.0, -4.0], 4)
    assert sc["sum_logprobs"] == -10.0 and sc["avg_logprobs"] == -2.5 and sc["score"] == -2.5
    assert abs(sc["entropy"] - math.log(2.0)) < 1e-12
    sc = WO.sequence_score(list(range(40)), [-0.5] * 40, 40)
    assert abs(sc["entropy"] - math.log(32.0)) < 1e-12                       # the last 32 tokens only
    assert WO.sequence_score([5], [-1.0], 0) is None
    sc = WO.sequence_score([1, 2, 3], [-1.0, -1.0, -1.0], 2, length_penalty=1.0)
    assert abs(sc["score"] - (-2.0 / ((5.0 + 2.0) / 6.0))) < 1e-12


def test_frames_of_a_chunk_and_the_100_ms_floor(scripted):
    assert WO.n_len_org(480000) == 2999 and WO.n_len_org(16000) == 99 and WO.n_len_org(1760) == 10
    made = scripted(lambda i: by_list([BEG, 7, BEG + 45, BEG + 45]))
    assert run(1500) == ([], [], []) and made == []          # 9 frames: "input is too short"
    assert run(1760)[2] == [] and made == []                 # 10 frames: seek + delta_min >= seek_end at once
    segs, kept, wins = run(16000)                            # 1 s: decoded (whisper.cpp < 1.7.6 refused it)
    assert len(wins) == 1 and kept == [BEG, 7, BEG + 45] and wins[0]["seek_advance"] == 99


def test_seek_loop_previous_text_and_the_single_timestamp_ending(scripted):
    """24 s.  Each window says "<|0.00|> a b <|7.00|><|7.00|> c <EOT>": kept up to the closed pair, the window moves on
    by 7 s, the next prompt carries <|startofprev|> + everything kept so far; at 21 s fewer than 5 s are left: the past is
    dropped, the first timestamp within 100 ms of the end closes the window, and a text-then-timestamp ending skips what
    is left of the chunk."""
    seen = []

    def factory(i):
        w = i + 1
        seq = [BEG, 1000 + w, 2000 + w, BEG + 350, BEG + 350, 3000 + w]

        def script(gen, prompt):
            if not gen:
                seen.append(list(prompt))
            return peaky(seq[len(gen)] if len(gen) < len(seq) else EOT)
        return script

    scripted(factory)
    n = 16000 * 24
    segs, kept, wins = run(n, n_max=10)
    assert [w["seek"] for w in wins] == [0, 700, 1400, 2100]
    assert [w["seek_advance"] for w in wins] == [700, 700, 700, 2399 - 2100]
    k = [[BEG, 1000 + w, 2000 + w, BEG + 350, BEG + 350] for w in (1, 2, 3)]
    assert seen[0] == INIT and seen[1] == [SP["prev"]] + k[0] + INIT and seen[2] == [SP["prev"]] + k[0] + k[1] + INIT
    assert seen[3] == INIT                                   # 2100 + 500 >= 2399: the past is dropped
    assert wins[3]["tokens"] == [BEG, 1004, 2004, BEG + 350] and kept == k[0] + k[1] + k[2] + wins[3]["tokens"]
    assert segs[0] == (0, 700, b" w1001 w2001") and segs[-1] == (2100, 2800, b" w1004 w2004")
    assert all(not w["failed"] and not w["is_no_speech"] and w["temperature"] == 0.0 for w in wins)
    assert all(abs(w["avg_logprob"]) < 1e-6 for w in wins)   # peaky picks: probability ~ 1
    seen.clear()
    run(n, n_max=10, prev_text=False)
    assert len(seen) == 4 and all(p == INIT for p in seen)
    # the cut: a past longer than n_text_ctx / 2 keeps its tail, and prompt + n_max never exceeds n_text_ctx
    seen.clear()
    long_seq = [BEG] + list(range(3000, 3150)) + [BEG + 300, BEG + 300]
    scripted(lambda i: (lambda gen, prompt: (seen.append(list(prompt)) if not gen else None,
                                              peaky(long_seq[len(gen)] if len(gen) < len(long_seq) else EOT))[1]))
    run(16000 * 30, params=dict(entropy_thold=-1.0))
    n_max = HP.n_text_ctx // 2 - 4
    assert len(seen[1]) == 1 + 153 + 3
    assert len(seen[2]) == 1 + min(HP.n_text_ctx // 2, HP.n_text_ctx - n_max - 3 - 1) + 3 and len(seen[2]) + n_max <= HP.n_text_ctx
    assert seen[2][-3:] == INIT and seen[2][0] == SP["prev"] and seen[2][-4] == BEG + 300


def test_no_speech_rule_drops_the_window_and_its_text(scripted):
    """no_speech_prob > 0.6 (taken at the last prompt position, before any filtering) and an average log-probability
    below -1: no segment, nothing kept, nothing added to the conditioning text -- and no fallback either (the fallback
    wants no_speech_prob < no_speech_thold).  With the same picks and a quiet <|nospeech|> the window is re-decoded up
    the whole ladder instead."""
    seq = [BEG, 100, 200, BEG + 1400, BEG + 1400]
    scripted(lambda i: (lambda gen, prompt: flat(seq[len(gen)] if len(gen) < len(seq) else EOT, nosp=12.0)))
    segs, kept, wins = run(16000 * 30, fallback=True)
    w = wins[0]
    assert w["is_no_speech"] and not w["failed"] and w["temperature"] == 0.0 and len(w["iterations"]) == 1
    assert w["no_speech_prob"] > 0.6 and -11.0 < w["avg_logprob"] < -5.0
    assert segs == [] and kept == [] and w["tokens"] == seq and w["seek_advance"] == 2800
    assert len(wins) == 2 and wins[1]["prompt"] == INIT      # 28 s: under 5 s left anyway; and nothing was added to the past
    # thresholds are options: a no_speech_thold of 1 keeps the text (and then the low log-probability asks for a fallback)
    scripted(lambda i: (lambda gen, prompt: flat(seq[len(gen)] if len(gen) < len(seq) else EOT, nosp=12.0)))
    segs, kept, wins = run(16000 * 30, fallback=False, params=dict(no_speech_thold=1.0))
    assert not wins[0]["is_no_speech"] and kept[:5] == seq and len(segs) >= 1
    scripted(lambda i: (lambda gen, prompt: flat(seq[len(gen)] if len(gen) < len(seq) else EOT, nosp=0.0)))
    segs, kept, wins = run(16000 * 30, fallback=True, max_windows=1)
    w = wins[0]
    assert [it["temperature"] for it in w["iterations"]] == pytest.approx([0.0, 0.2, 0.4, 0.6, 0.8, 1.0])
    assert [len(it["decoders"]) for it in w["iterations"]] == [1, 5, 5, 5, 5, 5]
    assert w["temperature"] == pytest.approx(1.0) and not w["is_no_speech"]


def test_decoder_failures_walk_the_ladder_and_the_last_temperature_is_accepted(scripted):
    """(a) end of text before any timestamp with more than a window of audio left (only a call over more than 30 s can get
    there: the untouched seek_delta is a whole window -- the reference's 30 s chunks always count as "at the end"):
    failed, at every temperature (the script is deterministic even when sampled), accepted at 1.0 with all its tokens
    and nothing for the conditioning text;
    (b) entropy: 40 repeats of one token between two timestamps; passes with the check off;
    (c) the token limit reached with less than half the window covered."""
    scripted(lambda i: by_list([500]))
    segs, kept, wins = run(16000 * 90, fallback=True, max_windows=1)
    w = wins[0]
    assert w["failed"] and w["temperature"] == pytest.approx(1.0) and len(w["iterations"]) == 6
    assert w["tokens"] == [500, EOT] and w["n_past"] == 0 and w["avg_logprob"] == -np.inf
    assert segs == [(0, 3000, b" w500")] and w["seek_advance"] == 3000
    assert all(d["failed"] for it in w["iterations"] for d in it["decoders"])
    # ... in a 30 s chunk the same picks are a completed window
    scripted(lambda i: by_list([500]))
    w = run(16000 * 30, fallback=True, max_windows=1)[2][0]
    assert not w["failed"] and w["tokens"] == [500, EOT] and w["temperature"] == 0.0 and w["n_past"] == 2
    rep = [BEG] + [7] * 40 + [BEG + 100, BEG + 100]
    scripted(lambda i: by_list(rep))
    w = run(16000 * 30, fallback=True, max_windows=1)[2][0]
    assert w["failed"] and w["temperature"] == pytest.approx(1.0) and w["tokens"] == rep and w["entropy"] < 2.4
    assert w["n_past"] == len(rep) and abs(w["avg_logprob"]) < 1e-6         # scored before it failed: cut to result_len
    scripted(lambda i: by_list(rep))
    w = run(16000 * 30, fallback=True, max_windows=1, params=dict(entropy_thold=-1.0))[2][0]
    assert not w["failed"] and w["temperature"] == 0.0
    scripted(lambda i: by_list([BEG] + list(range(100, 130))))
    w = run(16000 * 30, fallback=False, max_windows=1, n_max=12)[2][0]
    assert w["failed"] and len(w["tokens"]) == 12 and w["n_past"] == 0


def test_a_sampled_pass_rescues_a_window_the_greedy_pass_failed_on(scripted):
    """After <|0.00|> the model slightly prefers a word X (logit 10) that leads into low-probability text to a word Y (9)
    that leads into a confident sentence: the greedy pass takes X, its average log-probability falls below -1 while
    <|nospeech|> is quiet, so the window is decoded again -- five sampling decoders per temperature, and the first pass in
    which one of them takes Y and scores above the threshold is accepted.  The accepted pass is the best-scoring decoder
    that did not fail, decoder j draws from MT19937(j) -- two 32-bit draws per pick, carried from one pass to the next --,
    and the whole thing is deterministic."""
    X, Y = 1234, 2345
    good = [BEG, Y] + list(range(700, 707)) + [BEG + 200, BEG + 200]
    bad = [BEG, X] + list(range(800, 806)) + [BEG + 200, BEG + 200]

    def script(gen, prompt):
        if len(gen) == 0:
            return peaky(BEG)
        if len(gen) == 1:
            lg = np.full(V, -30.0)
            lg[X], lg[Y] = 10.0, 9.0
            return lg
        if gen[1] == Y:
            return peaky(good[len(gen)] if len(gen) < len(good) else EOT)
        return flat(bad[len(gen)]) if len(gen) < len(bad) else peaky(EOT)

    scripted(lambda i: script)
    a = run(16000 * 30, fallback=True, max_windows=1)
    scripted(lambda i: script)
    b = run(16000 * 30, fallback=True, max_windows=1)
    w = a[2][0]
    assert w["tokens"] == b[2][0]["tokens"] and w["temperature"] == b[2][0]["temperature"] and w["decoder"] == b[2][0]["decoder"]
    assert w["temperature"] > 0.0 and not w["failed"] and w["avg_logprob"] > -1.0
    assert w["tokens"] == good and a[1] == good
    its = w["iterations"]
    d0 = its[0]["decoders"][0]
    assert d0["toks"][:2] == [BEG, X] and not d0["failed"] and d0["score"]["avg_logprobs"] < -1.0
    last = its[-1]
    ok = [j for j, d in enumerate(last["decoders"]) if not d["failed"]]
    assert w["decoder"] == max(ok, key=lambda j: (last["decoders"][j]["score"]["score"], -j))
    # the generators: replay them -- every pick of every sampling decoder consumed one variate, in pass order
    rngs = [WO.MT19937(j) for j in range(5)]
    took_y = 0
    for it in its[1:]:
        T = it["temperature"]
        p_x = 1.0 / (1.0 + math.exp(-1.0 / T))
        for j, d in enumerate(it["decoders"]):
            for i, t in enumerate(d["toks"]):
                u = rngs[j].canonical()
                if i == 1 and abs(u - p_x) > 1e-9:           # the one open choice: X vs Y at logits 10 / 9 over T
                    assert t == (X if u < p_x else Y), (T, j, u, p_x)
                    took_y += t == Y
    assert took_y >= 1
    # the plog of the sampled choice is the log-softmax at that temperature
    T = w["temperature"]
    assert abs(w["plogs"][1] - (-math.log(1.0 + math.exp(1.0 / T)))) < 1e-6


def test_past_is_not_used_at_half_temperature_and_above(scripted):
    prompts = []

    def factory(i):
        def script(gen, prompt):
            if not gen:
                prompts.append(list(prompt))
            if len(prompts) == 1:                            # first window: fine
                seq = [BEG, 11, BEG + 300, BEG + 300]
                return peaky(seq[len(gen)] if len(gen) < len(seq) else EOT)
            rep = [BEG] + [7] * 40 + [BEG + 100, BEG + 100]  # second window: fails (entropy) at every temperature
            return peaky(rep[len(gen)] if len(gen) < len(rep) else EOT)
        return script

    scripted(factory)
    segs, kept, wins = run(16000 * 30, fallback=True, max_windows=2)
    assert len(prompts) == 1 + 6
    past = [SP["prev"], BEG, 11, BEG + 300, BEG + 300] + INIT
    assert prompts[1:4] == [past] * 3 and prompts[4:] == [INIT] * 3      # 0, 0.2, 0.4 | 0.6, 0.8, 1.0


def test_non_speech_tokens_are_looked_up_as_they_stand_and_with_a_leading_space():
    """whisper_full_params.suppress_nst [UPSTREAM-RECALL]: the list's strings that the vocabulary holds as ONE token, bare and
    behind a space; " -" and " '" on top ("-" and "'" themselves stay: hyphens and quotes inside words)."""
    vocab = [b" w%d" % i for i in range(40)]
    put = {3: b"(", 4: b" (", 7: b"-", 8: b" -", 9: b"'", 10: b" '", 12: "♪".encode(), 13: " ♪♪".encode(), 15: b"((",
           16: b" hello(", 20: b"\\", 21: b' "', 22: b"(", 30: b"--"}
    for i, t in put.items():
        vocab[i] = t
    ids = WO.non_speech_token_ids(vocab)
    assert ids == [3, 4, 8, 10, 12, 13, 15, 20, 21, 30]        # 22 is a second "(": token_to_id holds one id per string; 7, 9, 16 stay


def test_initial_prompt_and_carried_context_are_where_the_first_window_starts(scripted):
    """prompt_tokens are rotated in FRONT of the state's prompt_past [UPSTREAM-RECALL]; the first window is conditioned on
    both, later windows on what the loop made of them; `state` hands the text on to the next call (no_context = false)."""
    seen = []

    def script(gen, prompt):
        if not gen:
            seen.append(list(prompt))
        seq = [BEG, 700, 701, BEG + 100, BEG + 100, EOT]
        return peaky(seq[len(gen)] if len(gen) < len(seq) else EOT)

    scripted(lambda k: script)
    st = {}
    segs, kept, wins = run(16000 * 20, initial_prompt=[11, 12], past0=[21, 22, 23], state=st, max_windows=2)
    assert seen[0] == [SP["prev"], 11, 12, 21, 22, 23] + INIT
    assert wins[0]["prompt"] == seen[0] and kept[:4] == [BEG, 700, 701, BEG + 100]
    # the second window (2 s in, 18 s left) is conditioned on all of it plus what the first one kept
    first_kept = wins[0]["tokens"][:wins[0]["n_past"]]
    assert wins[1]["prompt"] == [SP["prev"], 11, 12, 21, 22, 23] + first_kept + INIT
    # the text the call ends with: what conditioned its last window + what that window kept
    assert st["prompt_past"] == [11, 12, 21, 22, 23] + first_kept + wins[1]["tokens"][:wins[1]["n_past"]]
    seen.clear()
    run(16000 * 4)
    assert seen[0] == INIT                                    # no_context = true, no prompt: the bare prompt


def test_beam_search_keeps_the_branch_the_greedy_pass_loses(scripted):
    """whisper.cpp's BEAM_SEARCH strategy [UPSTREAM-RECALL] on the X / Y model of the sampled-pass test: X (logit 10) leads
    into near-uniform text, Y (9) into a confident sentence.  Greedy takes X and fails the log-probability bar at
    temperature 0; with beam_size = 3 every live decoder draws three ids per step from its distribution, the candidates are
    sorted by the sum of all their log-probabilities and dealt to the decoders without repeating a sequence -- so as soon
    as one draw of nine is Y a decoder holds [<|0.00|>, Y], and from the next step on that branch sorts first.  The window
    is accepted at temperature 0 with the Y sentence; the decoders that held X-branches run on and are out-scored."""
    X, Y = 1234, 2345
    good = [BEG, Y] + list(range(700, 707)) + [BEG + 200, BEG + 200]
    bad = [BEG, X] + list(range(800, 806)) + [BEG + 200, BEG + 200]

    def script(gen, prompt):
        if len(gen) == 0:
            return peaky(BEG)
        if len(gen) == 1:
            lg = np.full(V, -30.0)
            lg[X], lg[Y] = 10.0, 9.0
            return lg
        if gen[1] == Y:
            return peaky(good[len(gen)] if len(gen) < len(good) else EOT)
        return flat(bad[len(gen)]) if len(gen) < len(bad) else peaky(EOT)

    scripted(lambda i: script)
    g = run(16000 * 30, fallback=False, max_windows=1, n_max=24)
    assert g[2][0]["tokens"][:2] == [BEG, X] and g[2][0]["avg_logprob"] < -1.0            # what greedy does with it
    scripted(lambda i: script)
    a = run(16000 * 30, fallback=True, max_windows=1, n_max=24, params=dict(beam_size=3))
    scripted(lambda i: script)
    b = run(16000 * 30, fallback=True, max_windows=1, n_max=24, params=dict(beam_size=3))
    w = a[2][0]
    assert w["tokens"] == b[2][0]["tokens"] and w["decoder"] == b[2][0]["decoder"]         # deterministic: MT19937(j) per decoder
    assert w["temperature"] == 0.0 and not w["failed"] and w["avg_logprob"] > -1.0
    assert w["tokens"] == good and a[1] == good
    it0 = w["iterations"][0]
    assert len(w["iterations"]) == 1 and len(it0["decoders"]) == 3
    win = it0["decoders"][w["decoder"]]
    assert win["completed"] and win["toks"][:len(good)] == good
    # the sum the candidates are sorted by is the sum of ALL plogs of the sequence
    for d in it0["decoders"]:
        assert abs(d["sum_all"] - sum(float(np.float32(p)) for p in d["plogs"])) < 1e-9
    # the second pick: log-softmax of (10, 9) at Y, no temperature scaling at temperature 0
    assert abs(win["plogs"][1] - (-math.log(1.0 + math.exp(1.0)))) < 1e-6
    # the draws: at step 1 every decoder draws three variates from MT19937(j); X iff u < p(X) -- at least one of the nine is Y
    p_x = 1.0 / (1.0 + math.exp(-1.0))
    ys = 0
    for j in range(3):
        r = WO.MT19937(j)
        [r.canonical() for _ in range(3)]                    # step 0: three draws, all <|0.00|>
        ys += sum(r.canonical() >= p_x for _ in range(3))
    assert ys >= 1


def test_beam_candidates_are_dealt_in_order_of_their_summed_log_probability(scripted):
    """The dealing rule on a model with three comparable continuations: after <|0.00|> the ids A, B, C carry 50 / 30 / 20 %.
    Whatever the nine draws of the step are, the live decoders end up with the DISTINCT drawn sequences in descending
    probability order, decoder 0 first, wrapping around to the best one when fewer distinct sequences were drawn than
    there are decoders (whisper.cpp: `if (cur_c >= beam_candidates.size()) cur_c = 0`)."""
    A, B, C_ = 1111, 2222, 3333

    def script(gen, prompt):
        if len(gen) == 0:
            return peaky(BEG)
        if len(gen) == 1:
            lg = np.full(V, -40.0)
            lg[A], lg[B], lg[C_] = math.log(0.5), math.log(0.3), math.log(0.2)
            return lg
        return peaky([BEG + 100, BEG + 100][len(gen) - 2] if len(gen) < 4 else EOT)

    scripted(lambda i: script)
    dc = WO.DecoderCache(None, HP, None)
    rngs = [WO.MT19937(j) for j in range(3)]
    r = WO.decode_temperature(dc, INIT, SP, WO.RULES_WCPP, 2, 0, 3000, 0.0, 3, rngs, WO.WCPP_PARAMS, SUP, SUP_FIRST, beam_size=3)
    drawn = []
    for j in range(3):
        g = WO.MT19937(j)
        [g.canonical() for _ in range(3)]
        for _ in range(3):
            u = g.canonical()
            drawn.append(A if u <= 0.5 else B if u <= 0.8 else C_)
    order = [t for t in (A, B, C_) if t in drawn]
    want = [order[j % len(order)] if j < len(order) else order[(j - len(order)) % len(order)] for j in range(3)]
    assert [d["toks"] for d in r["decoders"]] == [[BEG, t] for t in want]
    assert all(d["parent"] in (0, 1, 2) for d in r["decoders"])
