"""whisper_oracle.py -- TEST INFRASTRUCTURE ONLY (parity oracle).

float64 numpy restatement of the Whisper encoder / decoder behind
  transcribe_rs::SpeechModel::transcribe (whisper_cpp::WhisperEngine)
  reference call sites: src-tauri/src/managers/transcription.rs:183-185, 213-215
i.e. the whisper.cpp compute graph (whisper-rs-sys 0.15.0, Cargo.lock:6235-6245; source not vendored).
Architecture per SURVEY.md Appendix B.2 [UPSTREAM-RECALL]: conv stem (k3/p1, k3/s2/p1, GELU), fixed
sinusoidal positions, pre-LN blocks (q, v, out with bias; k without), q.k scaled by d_head^-1/2,
ln_post; decoder with learned positions, causal self-attention, cross-attention, tied output embedding.

PARITY UNPINNED against the reference itself (cannot be built here).  Pinned instead against HuggingFace
`WhisperForConditionalGeneration` (transformers) with identical seeded weights: tests/golden/make_whisper_golden.py
and tests/test_oracle_whisper.py.  `encoder_forward` is the exact-arithmetic version of the graph;
`encoder_forward_f16` rounds the operands of every matrix product to f16 as ggml's mul_mat does (what precision
mode 1 of the library implements).  Not restated: ggml's f16 GELU lookup table.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import numpy as np
from scipy.special import erf


def _gelu(x):
    return 0.5 * x * (1.0 + erf(x / x.dtype.type(np.sqrt(2.0))))


def _gelu_ggml(x):
    """GELU as ggml's CPU backend computes it [UPSTREAM-RECALL: ggml_vec_gelu_f32 under GGML_GELU_FP16]: a table indexed
    by the f16 bit pattern of x whose entry is f16(ggml_gelu_f32(x)) = f16(0.5 x (1 + tanhf(sqrt(2 / pi) x (1 + 0.044715
    x^2)))) evaluated in f32; x <= -10 -> 0 and x >= 10 -> x in front of the look-up.  The table memoises a pure function,
    so it is evaluated directly here.  Used by the f16-operand (precision mode 1) chain; the exact oracle keeps erf."""
    x = np.asarray(x)
    xf = x.astype(np.float32)
    xh = xf.astype(np.float16).astype(np.float32)
    one, a, s2pi, half = np.float32(1.0), np.float32(0.044715), np.float32(0.79788456080286535587989211986876), np.float32(0.5)
    with np.errstate(over="ignore"):
        y = (half * xh * (one + np.tanh(s2pi * xh * (one + a * xh * xh)))).astype(np.float16).astype(np.float32)
    y = np.where(xf <= -10.0, np.float32(0.0), np.where(xf >= 10.0, xf, y))
    return y.astype(x.dtype)


def _ln(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def _f64(w, dtype=np.float64):
    return {k: v.astype(dtype) for k, v in w.items()}


def _mha(xq, xkv, W, prefix, n_head, causal=False):
    q = xq @ W[prefix + ".query.weight"].T + W[prefix + ".query.bias"]
    k = xkv @ W[prefix + ".key.weight"].T
    v = xkv @ W[prefix + ".value.weight"].T + W[prefix + ".value.bias"]
    Tq, D = q.shape
    dh = D // n_head
    out = np.empty_like(q)
    for h in range(n_head):
        sl = slice(h * dh, (h + 1) * dh)
        s = (q[:, sl] @ k[:, sl].T) / q.dtype.type(np.sqrt(dh))
        if causal:
            s = s + np.triu(np.full((Tq, k.shape[0]), -np.inf, dtype=q.dtype), k=1 + k.shape[0] - Tq)
        s = s - s.max(-1, keepdims=True)
        p = np.exp(s)
        p /= p.sum(-1, keepdims=True)
        out[:, sl] = p @ v[:, sl]
    return out @ W[prefix + ".out.weight"].T + W[prefix + ".out.bias"]


def encoder_forward(weights, hp, mel, upto_layer=None, dtype=np.float64):
    """mel: [n_mels, 3000] -> [1500, d] (float64; dtype=np.float32 is the single-precision CPU run that bench.py times
    as the ASR cpu_baseline -- BLAS sgemm on the host cores -- never the parity oracle)."""
    W = _f64(weights, dtype)
    x = mel.astype(dtype)
    xp = np.pad(x, ((0, 0), (1, 1)))
    w1 = W["encoder.conv1.weight"]
    h1 = sum(w1[:, :, k] @ xp[:, k:k + 3000] for k in range(3)) + W["encoder.conv1.bias"][:, None]
    h1 = _gelu(h1)
    hp1 = np.pad(h1, ((0, 0), (1, 1)))
    w2 = W["encoder.conv2.weight"]
    h2 = sum(w2[:, :, k] @ hp1[:, k:k + 3000:2][:, :1500] for k in range(3)) + W["encoder.conv2.bias"][:, None]
    x = _gelu(h2).T + W["encoder.positional_embedding"]
    n_layers = hp.n_audio_layer if upto_layer is None else upto_layer
    for i in range(n_layers):
        p = f"encoder.blocks.{i}"
        xn = _ln(x, W[p + ".attn_ln.weight"], W[p + ".attn_ln.bias"])
        x = x + _mha(xn, xn, W, p + ".attn", hp.n_audio_head)
        xn = _ln(x, W[p + ".mlp_ln.weight"], W[p + ".mlp_ln.bias"])
        x = x + _gelu(xn @ W[p + ".mlp.0.weight"].T + W[p + ".mlp.0.bias"]) @ W[p + ".mlp.2.weight"].T + W[p + ".mlp.2.bias"]
    if upto_layer is None:
        x = _ln(x, W["encoder.ln_post.weight"], W["encoder.ln_post.bias"])
    return x


def _h(a):
    """Round to IEEE binary16 (round to nearest even) and come back: the value an f16 operand carries."""
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float64)


def encoder_forward_f16(weights, hp, mel):
    """The encoder with the numerics of whisper.cpp's ggml matrix products [UPSTREAM-RECALL] -- what precision mode 1 of
    the library implements: every matrix product takes BOTH operands rounded to f16 (the 2-D weights, and the
    activation that enters the product) and accumulates exactly (float64 here, f32 on the matrix cores; the difference
    is ~1e-6 of the result); GELU is ggml's (`_gelu_ggml`: the f16-indexed table of the tanh form); everything else --
    biases, LayerNorm statistics, soft-max, the residual stream -- is f32 / exact.  Rounding points, in graph order:
      conv1   the log-mel frames and the conv1 kernel rounded (ggml: im2col in f16 x f16 kernel)
      conv2   GELU(conv1) and the conv2 kernel rounded
      block   LN(x) rounded -> q, k, v products; q, k, v rounded (+bias first); soft-max probabilities rounded as
              2^(t - m) with t = s log2(e) and an INTEGER reference exponent m (the mantissa of 2^t: the rounding does not
              depend on m, which is what lets a tiled kernel reproduce it; the sum uses the unrounded values),
              P.V on rounded operands, the normalised result rounded -> out projection; LN(x) rounded -> fc1;
              GELU(fc1) rounded -> fc2
      ln_post exact."""
    W = _f64(weights)
    x = _h(mel)
    xp = np.pad(x, ((0, 0), (1, 1)))
    w1 = _h(W["encoder.conv1.weight"])
    h1 = sum(w1[:, :, k] @ xp[:, k:k + 3000] for k in range(3)) + W["encoder.conv1.bias"][:, None]
    h1 = _h(_gelu_ggml(h1))
    hp1 = np.pad(h1, ((0, 0), (1, 1)))
    w2 = _h(W["encoder.conv2.weight"])
    h2 = sum(w2[:, :, k] @ hp1[:, k:k + 3000:2][:, :1500] for k in range(3)) + W["encoder.conv2.bias"][:, None]
    x = _gelu_ggml(h2).T + W["encoder.positional_embedding"]
    H = hp.n_audio_head
    for i in range(hp.n_audio_layer):
        p = f"encoder.blocks.{i}"
        xn = _h(_ln(x, W[p + ".attn_ln.weight"], W[p + ".attn_ln.bias"]))
        q = _h(xn @ _h(W[p + ".attn.query.weight"]).T + W[p + ".attn.query.bias"])
        k = _h(xn @ _h(W[p + ".attn.key.weight"]).T)
        v = _h(xn @ _h(W[p + ".attn.value.weight"]).T + W[p + ".attn.value.bias"])
        att = np.empty_like(q)
        dh = q.shape[1] // H
        for hh in range(H):
            sl = slice(hh * dh, (hh + 1) * dh)
            t = (q[:, sl] @ k[:, sl].T) / np.sqrt(dh) * np.log2(np.e)
            pe = np.exp2(t - np.ceil(t.max(-1, keepdims=True)))      # integer reference exponent: the mantissa of 2^t
            att[:, sl] = (_h(pe) @ v[:, sl]) / pe.sum(-1, keepdims=True)
        x = x + _h(att) @ _h(W[p + ".attn.out.weight"]).T + W[p + ".attn.out.bias"]
        xn = _h(_ln(x, W[p + ".mlp_ln.weight"], W[p + ".mlp_ln.bias"]))
        hid = _h(_gelu_ggml(xn @ _h(W[p + ".mlp.0.weight"]).T + W[p + ".mlp.0.bias"]))
        x = x + hid @ _h(W[p + ".mlp.2.weight"]).T + W[p + ".mlp.2.bias"]
    return _ln(x, W["encoder.ln_post.weight"], W["encoder.ln_post.bias"])


def decoder_logits(weights, hp, enc_out, tokens):
    """Full (non-cached) decoder pass: tokens [n] -> logits [n, n_vocab] (float64)."""
    W = _f64(weights)
    tokens = np.asarray(tokens, dtype=np.int64)
    x = W["decoder.token_embedding.weight"][tokens] + W["decoder.positional_embedding"][:len(tokens)]
    enc = enc_out.astype(np.float64)
    for i in range(hp.n_text_layer):
        p = f"decoder.blocks.{i}"
        xn = _ln(x, W[p + ".attn_ln.weight"], W[p + ".attn_ln.bias"])
        x = x + _mha(xn, xn, W, p + ".attn", hp.n_text_head, causal=True)
        xn = _ln(x, W[p + ".cross_attn_ln.weight"], W[p + ".cross_attn_ln.bias"])
        x = x + _mha(xn, enc, W, p + ".cross_attn", hp.n_text_head)
        xn = _ln(x, W[p + ".mlp_ln.weight"], W[p + ".mlp_ln.bias"])
        x = x + _gelu(xn @ W[p + ".mlp.0.weight"].T + W[p + ".mlp.0.bias"]) @ W[p + ".mlp.2.weight"].T + W[p + ".mlp.2.bias"]
    x = _ln(x, W["decoder.ln.weight"], W["decoder.ln.bias"])
    return x @ W["decoder.token_embedding.weight"].T


def final_logits(weights, x, f16=False):
    """The last block of a decoder step on its own: logits = LN(x) . E^T for decoder states x [n, d] (float64).
    f16=True is the arithmetic of whisper.cpp's ggml graph [UPSTREAM-RECALL] for this product -- what precision mode 1 of
    the library implements: the LayerNorm output (src1 of mul_mat) and the token embedding (an f16 tensor in every
    ggml-model.bin) both rounded to f16, exact accumulation (f32 on the matrix cores)."""
    W = _f64(weights)
    xn = _ln(np.asarray(x, dtype=np.float64), W["decoder.ln.weight"], W["decoder.ln.bias"])
    E = W["decoder.token_embedding.weight"]
    return (_h(xn) @ _h(E).T) if f16 else (xn @ E.T)


def greedy_decode(weights, hp, enc_out, prompt, n_new, suppress=None, eot=50257):
    """Greedy continuation of `prompt`: argmax of the last position's logits (ties -> lowest id), suppressed
    ids masked to -inf; stops after n_new tokens or at EOT.  Returns (tokens, logit of each pick, margin to
    the runner-up)."""
    toks = list(prompt)
    picks, best, margin = [], [], []
    for _ in range(n_new):
        lg = decoder_logits(weights, hp, enc_out, toks)[-1]
        if suppress is not None:
            lg = lg.copy()
            lg[np.asarray(suppress, dtype=np.int64)] = -np.inf
        t = int(np.argmax(lg))
        srt = np.partition(lg, -2)[-2:]
        picks.append(t); best.append(float(lg[t])); margin.append(float(srt[1] - srt[0]))
        toks.append(t)
        if t == eot:
            break
    return picks, best, margin


# ---------------------------------------------------------------------------------------------------------
# Timestamp-mode decoding (whisper.cpp `whisper_full` with `no_timestamps = false`, its default) -- the path
# that produces `segments` for managers/transcription.rs:223-233.
#
# Two rule flavours:
#   RULES_OPENAI   openai-whisper `ApplyTimestampRules` as carried by HuggingFace's
#                  `WhisperTimeStampLogitsProcessor`; PINNED by tests/golden/whisper_tiny_golden.npz (ts_*).
#   RULES_WCPP     whisper.cpp `whisper_process_logits` [UPSTREAM-RECALL, source not vendored]: no forced
#                  initial timestamp, timestamps may repeat the last one (`< last` suppressed instead of
#                  `<= last` after a closed pair), monotonicity only once a token > <|0.00|> was sampled.
# Common to both: <|notimestamps|> suppressed, timestamps come in pairs except before EOT, the first
# timestamp is at most `max_initial_ts` (1.0 s = index 50), and a timestamp is forced when the probability
# mass of all timestamps exceeds the most probable text token.
# ---------------------------------------------------------------------------------------------------------
RULES_WCPP = 0
RULES_OPENAI = 1


class DecoderCache:
    """KV-cached incremental decoder (float64); `step(token)` returns the logits of the new position.
    Same arithmetic as `decoder_logits`, restated so that 200-token windows finish in seconds."""

    def __init__(self, weights, hp, enc_out, f16=False, dtype=np.float64, ln16=False):
        """dtype=np.float32: single-precision CPU run for bench.py's cpu_baseline (not the parity oracle).
        f16=True: the decoder arithmetic of the library's precision mode 1, i.e. whisper.cpp's ggml graph
        [UPSTREAM-RECALL] wherever a matrix product has no LayerNorm folded into it on the GPU -- cross K | V from the
        f16-rounded encoder output and f16 weights, stored as f16 (kv_cross); the self-attention K | V cache stored
        as f16 (kv_self); the attention outputs and the GELU'd hidden layer rounded to f16 against f16 weights
        (attn.out, cross_attn.out, mlp.2); the final LayerNorm rounded to f16 against the f16 token embedding.  The
        projections behind a LayerNorm (q | k | v, cross q, mlp.0) stay exact: the library folds the LayerNorm into
        them and keeps them in f32."""
        self.W = _f64(weights, dtype)
        self.hp = hp
        self.f16 = f16
        # ln16 (the library's precision mode 2): the LayerNorm output is rounded to f16 -- and the weight too -- in front of
        # the q | k | v, cross-q and fc1 products, as ggml's mul_mat does with an f32 activation against an f16 weight
        # [UPSTREAM-RECALL]; mode 1 keeps those five products exact (LayerNorm folded into f32 GEMMs)
        self.ln16 = bool(ln16 and f16)
        r = _h if f16 else (lambda a: a)
        self.r = r
        self.rl = _h if self.ln16 else (lambda a: a)
        enc = r(enc_out.astype(dtype))
        self.xk, self.xv, self.k, self.v = [], [], [], []
        for i in range(hp.n_text_layer):
            p = f"decoder.blocks.{i}.cross_attn"
            self.xk.append(r(enc @ r(self.W[p + ".key.weight"]).T))
            self.xv.append(r(enc @ r(self.W[p + ".value.weight"]).T + self.W[p + ".value.bias"]))
            self.k.append(np.zeros((0, hp.n_text_state), dtype=dtype))
            self.v.append(np.zeros((0, hp.n_text_state), dtype=dtype))
        self.pos = 0

    def _att(self, q, k, v):
        H = self.hp.n_text_head
        dh = q.shape[-1] // H
        out = np.empty_like(q)
        for h in range(H):
            sl = slice(h * dh, (h + 1) * dh)
            s = (k[:, sl] @ q[sl]) / q.dtype.type(np.sqrt(dh))
            s = np.exp(s - s.max())
            out[sl] = (s / s.sum()) @ v[:, sl]
        return out

    def step(self, token):
        W, r = self.W, self.r
        x = W["decoder.token_embedding.weight"][int(token)] + W["decoder.positional_embedding"][self.pos]
        for i in range(self.hp.n_text_layer):
            p = f"decoder.blocks.{i}"
            rl = self.rl
            xn = rl(_ln(x, W[p + ".attn_ln.weight"], W[p + ".attn_ln.bias"]))
            q = xn @ rl(W[p + ".attn.query.weight"]).T + W[p + ".attn.query.bias"]
            self.k[i] = np.vstack([self.k[i], r(xn @ rl(W[p + ".attn.key.weight"]).T)])
            self.v[i] = np.vstack([self.v[i], r(xn @ rl(W[p + ".attn.value.weight"]).T + W[p + ".attn.value.bias"])])
            x = x + r(self._att(q, self.k[i], self.v[i])) @ r(W[p + ".attn.out.weight"]).T + W[p + ".attn.out.bias"]
            xn = rl(_ln(x, W[p + ".cross_attn_ln.weight"], W[p + ".cross_attn_ln.bias"]))
            q = xn @ rl(W[p + ".cross_attn.query.weight"]).T + W[p + ".cross_attn.query.bias"]
            x = x + r(self._att(q, self.xk[i], self.xv[i])) @ r(W[p + ".cross_attn.out.weight"]).T + W[p + ".cross_attn.out.bias"]
            xn = rl(_ln(x, W[p + ".mlp_ln.weight"], W[p + ".mlp_ln.bias"]))
            g = (_gelu_ggml if self.f16 else _gelu)(xn @ rl(W[p + ".mlp.0.weight"]).T + W[p + ".mlp.0.bias"])
            x = x + r(g) @ r(W[p + ".mlp.2.weight"]).T + W[p + ".mlp.2.bias"]
        self.pos += 1
        x = _ln(x, W["decoder.ln.weight"], W["decoder.ln.bias"])
        return r(x) @ r(W["decoder.token_embedding.weight"]).T


def special_tokens(n_vocab, eot=None):
    """whisper.cpp vocabulary layout [UPSTREAM-RECALL: `whisper_vocab` defaults + the shift applied when the model is
    loaded].  The defaults are the English-only layout (n_vocab 51864): eot 50256, sot 50257, translate 50357,
    transcribe 50358, solm 50359, prev 50360, nosp 50361, notimestamps 50362, <|0.00|> 50363 -- the 99 language
    slots after sot stay in the .en vocabulary although no prompt names them.  A multilingual vocabulary
    (n_vocab >= 51865) moves eot / sot up by one and everything behind the languages by 1 + (languages - 99)."""
    multilingual = n_vocab >= 51865
    extra = n_vocab - 51865 if multilingual else 0
    if eot is None:
        eot = 50257 if multilingual else 50256
    sot = eot + 1
    t = sot + 100 + extra
    d = dict(sot=sot, lang0=sot + 1, n_lang=99 + extra if multilingual else 0, n_lang_slots=99 + extra,
             translate=t, transcribe=t + 1, solm=t + 2, prev=t + 3, nosp=t + 4, not_=t + 5, beg=t + 6, eot=eot,
             multilingual=multilingual)
    return d


def default_prompt(n_vocab, lang_token=None, translate=False, no_timestamps=False):
    """whisper_full's initial prompt: [sot] for English-only vocabularies, [sot, language, task] for multilingual
    ones, + <|notimestamps|> when timestamps are off."""
    sp = special_tokens(n_vocab)
    p = [sp["sot"]]
    if sp["multilingual"]:
        p += [sp["lang0"] if lang_token is None else lang_token, sp["translate"] if translate else sp["transcribe"]]
    if no_timestamps:
        p.append(sp["not_"])
    return p


def timestamp_rules(lg, seq, sp, rules, suppress=None, suppress_first=None, max_initial_ts=50):
    """Masked copy of the logits `lg` for the next pick, given the tokens `seq` sampled so far in this window.
    Returns (masked logits, index of the most probable timestamp token)."""
    lg = np.array(lg, dtype=np.float64)
    beg, eot = sp["beg"], sp["eot"]
    if suppress is not None and len(suppress):
        lg[np.asarray(suppress, dtype=np.int64)] = -np.inf
    if len(seq) == 0 and suppress_first is not None and len(suppress_first):
        lg[np.asarray(suppress_first, dtype=np.int64)] = -np.inf
    lg[sp["not_"]] = -np.inf
    last_ts = len(seq) >= 1 and seq[-1] >= beg
    pen_ts = len(seq) < 2 or seq[-2] >= beg
    if last_ts:
        if pen_ts:
            lg[beg:] = -np.inf
        else:
            lg[:eot] = -np.inf
    if rules == RULES_OPENAI:
        ts = [t for t in seq if t >= beg]
        if ts:
            last = ts[-1] if (last_ts and not pen_ts) else ts[-1] + 1
            lg[beg:last] = -np.inf
        if len(seq) == 0:
            lg[:beg] = -np.inf
            if max_initial_ts is not None:
                lg[beg + max_initial_ts + 1:] = -np.inf
    else:
        if len(seq) == 0 and max_initial_ts is not None and max_initial_ts > 0:
            lg[beg + max_initial_ts + 1:] = -np.inf
        ts = [t for t in seq if t > beg]          # whisper.cpp: has_ts / seek_delta only move on tokens > <|0.00|>
        if ts:
            lg[beg:ts[-1]] = -np.inf
    tsl = lg[beg:]
    tid = beg + int(np.argmax(tsl)) if np.isfinite(tsl.max()) else beg
    m = tsl.max()
    lse_ts = m + np.log(np.exp(tsl - m).sum()) if np.isfinite(m) else -np.inf
    if lse_ts > lg[:beg].max():
        lg[:beg] = -np.inf
    return lg, tid


def decode_window(step_logits, prompt, sp, rules, n_max, seek, seek_end, suppress=None, suppress_first=None,
                  max_initial_ts=50):
    """One whisper_full window [UPSTREAM-RECALL]: greedy picks under the timestamp rules until EOT, n_max tokens
    or a timestamp within 1 s of the end of the audio.  `step_logits(token)` feeds one token and returns the next
    logits.  Returns dict(tokens, tids, result_len, seek_delta, margins)."""
    lg = None
    for t in prompt:
        lg = step_logits(t)
    beg, eot = sp["beg"], sp["eot"]
    toks, tids, margins = [], [], []
    has_ts, seek_delta, result_len = False, 3000, 0
    for i in range(n_max):
        ml, tid = timestamp_rules(lg, toks, sp, rules, suppress, suppress_first, max_initial_ts)
        t = int(np.argmax(ml))
        top2 = np.partition(ml, -2)[-2:]
        margins.append(float(top2[1] - top2[0]))
        toks.append(t)
        tids.append(tid if t < beg else t)
        if t > beg:
            seek_delta = 2 * (t - beg)
            result_len = i + 1
            has_ts = True
        if t == eot or (has_ts and seek + seek_delta + 100 >= seek_end):
            if t == eot and result_len == 0:
                result_len = i + 1            # no temperature fallback here: keep what was decoded
            break
        lg = step_logits(t)
    else:
        if result_len == 0:
            result_len = len(toks)
    return dict(tokens=toks, tids=tids, result_len=result_len, seek_delta=seek_delta, margins=margins)


def window_segments(win, seek, sp, token_text):
    """Segments of one window as whisper_full builds them [UPSTREAM-RECALL]: text between timestamp tokens,
    t0 / t1 in centiseconds (mel frames) relative to the clip."""
    beg, eot = sp["beg"], sp["eot"]
    toks = win["tokens"][:win["result_len"]]
    tids = win["tids"][:win["result_len"]]
    segs = []
    if not toks:
        return segs
    t0 = seek + 2 * (tids[0] - beg)
    text = b""
    i = 0
    while i < len(toks):
        if toks[i] < eot:
            text += token_text(toks[i])
        if toks[i] > beg:
            t1 = seek + 2 * (tids[i] - beg)
            if text:
                segs.append((t0, t1, text))
            text = b""
            while i < len(toks) and toks[i] > beg:
                i += 1
            i -= 1
            t0 = t1
        i += 1
    if text:
        segs.append((t0, seek + win["seek_delta"], text))
    return segs


def transcribe_timestamps(weights, hp, mel_window, n_samples, prompt, rules, token_text, n_max=None,
                          suppress=None, suppress_first=None, eot=None, max_windows=1501, f16=False, prev_text=True):
    """whisper_full's seek loop over one clip (<= 30 s): `mel_window(seek)` returns the [n_mels, 3000] log-mel
    window starting at mel frame `seek`.  Returns (segments, all kept tokens, windows).
    f16=True chains the f16-operand arithmetic (the library's precision mode 1 = ggml's mul_mat numerics
    [UPSTREAM-RECALL]): `encoder_forward_f16` -> `DecoderCache(f16=True)`.
    prev_text=True is whisper.cpp's `prompt_past` [UPSTREAM-RECALL: whisper_full_with_state]: from the second window on
    the prompt is <|startofprev|> + the last min(n_text_ctx / 2, len(past)) tokens of the text so far + `prompt`; the past
    is dropped when fewer than 5 s of audio are left (`seek > seek_start && seek + 500 >= seek_end`); after a window the
    past becomes the past part of its prompt + its kept tokens (timestamp tokens and all)."""
    sp = special_tokens(hp.n_vocab, eot)
    n_max = hp.n_text_ctx // 2 - 4 if n_max is None else n_max
    seek, seek_end = 0, n_samples // 160
    segs, kept, wins = [], [], []
    if seek_end < 100:                         # whisper.cpp: "input is too short" -> nothing
        return segs, kept, wins
    prompt = list(prompt)
    past = []
    while seek + 100 < seek_end and len(wins) < max_windows:
        if seek > 0 and seek + 500 >= seek_end:
            past = []
        p = list(prompt)
        if prev_text and past:
            n_take = min(hp.n_text_ctx // 2, len(past), hp.n_text_ctx - n_max - len(prompt) - 1)
            if n_take > 0:
                p = [sp["prev"]] + past[len(past) - n_take:] + p
        enc = (encoder_forward_f16 if f16 else encoder_forward)(weights, hp, mel_window(seek))
        dc = DecoderCache(weights, hp, enc, f16=f16)
        win = decode_window(dc.step, p, sp, rules, n_max, seek, seek_end, suppress, suppress_first)
        win["seek"] = seek
        win["prompt"] = p
        wins.append(win)
        segs += window_segments(win, seek, sp, token_text)
        kept += win["tokens"][:win["result_len"]]
        past = (p[1:len(p) - len(prompt)] if p[0] == sp["prev"] else []) + list(win["tokens"][:win["result_len"]])
        seek += win["seek_delta"]
    return segs, kept, wins
