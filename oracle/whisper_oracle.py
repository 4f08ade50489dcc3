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
    as the ASR cpu_baseline -- BLAS sgemm on the host cores -- and the oracle of the two largest catalog models in
    tests/test_gpu_whisper.py::test_catalog_models_at_full_depth, where float64 costs 100 s of the GPU suite's time
    limit and the single-precision rounding is two orders below the bar)."""
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


def encoder_forward_f16(weights, hp, mel, attn16=False):
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
      ln_post exact.
    attn16 (the library's precision mode 2): the soft-max is taken in full, NORMALISED, and the normalised probabilities
    are what is rounded to f16 in front of P.V -- ggml's order (soft_max_ext writes f32 probabilities, mul_mat against
    the f16 V converts them) [UPSTREAM-RECALL] -- instead of mode 1's 2^(t - m) mantissas divided afterwards."""
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
            if attn16:
                pe = np.exp2(t - t.max(-1, keepdims=True))
                att[:, sl] = _h(pe / pe.sum(-1, keepdims=True)) @ v[:, sl]
                continue
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

    def __init__(self, weights, hp, enc_out, f16=False, dtype=np.float64, ln16=None, attn16=False):
        """dtype=np.float32: single-precision CPU run for bench.py's cpu_baseline (not the parity oracle).
        f16=True: the decoder arithmetic of the library's precision mode 1, i.e. whisper.cpp's ggml graph
        [UPSTREAM-RECALL] wherever a matrix product has no LayerNorm folded into it on the GPU -- cross K | V from the
        f16-rounded encoder output and f16 weights, stored as f16 (kv_cross); the self-attention K | V cache stored
        as f16 (kv_self); the attention outputs and the GELU'd hidden layer rounded to f16 against f16 weights
        (attn.out, cross_attn.out, mlp.2); the final LayerNorm rounded to f16 against the f16 token embedding; and
        (ln16, on with f16 unless switched off) the LayerNorm outputs rounded to f16 against f16 weights in front of
        q | k | v, cross q and mlp.0."""
        self.W = _f64(weights, dtype)
        self.hp = hp
        self.f16 = f16
        # ln16: the LayerNorm output is rounded to f16 -- and the weight too -- in front of the q | k | v, cross-q and fc1
        # products, as ggml's mul_mat does with an f32 activation against an f16 weight [UPSTREAM-RECALL].  Part of the
        # library's precision modes 1 and 2 since round 5 (default: follows f16); ln16=False is the arithmetic of rounds
        # 2 - 4, where mode 1 kept those five products exact (LayerNorm folded into f32 GEMMs).
        if ln16 is None:
            ln16 = f16
        self.ln16 = bool(ln16 and f16)
        # attn16 (mode 2 as well): the query is rounded to f16 in front of K.q and the NORMALISED soft-max probabilities in
        # front of P.V -- ggml's mul_mat converts its f32 operand to the f16 of the K / V cache it multiplies
        # [UPSTREAM-RECALL]; mode 1 keeps q and the probabilities in f32 against the f16 caches
        self.attn16 = bool(attn16 and f16)
        r = _h if f16 else (lambda a: a)
        self.r = r
        self.rl = _h if self.ln16 else (lambda a: a)
        enc = r(enc_out.astype(dtype))
        self.xk, self.xv, self.k, self.v = [], [], [], []
        # the weights as the products see them, rounded once (r: the plain products of the f16 chain; rl: the
        # LayerNorm-fed ones of mode 2) -- the same values step() used to round anew at every position
        self.Wr, self.Wl = {}, {}
        for i in range(hp.n_text_layer):
            p = f"decoder.blocks.{i}"
            for k in (".attn.out.weight", ".cross_attn.out.weight", ".mlp.2.weight"):
                self.Wr[p + k] = r(self.W[p + k])
            for k in (".attn.query.weight", ".attn.key.weight", ".attn.value.weight", ".cross_attn.query.weight", ".mlp.0.weight"):
                self.Wl[p + k] = self.rl(self.W[p + k])
        self.Wr["decoder.token_embedding.weight"] = r(self.W["decoder.token_embedding.weight"])
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
            qh = _h(q[sl]).astype(q.dtype) if self.attn16 else q[sl]
            s = (k[:, sl] @ qh) / q.dtype.type(np.sqrt(dh))
            s = np.exp(s - s.max())
            pr = s / s.sum()
            out[sl] = (_h(pr).astype(q.dtype) if self.attn16 else pr) @ v[:, sl]
        return out

    def fork(self):
        """A decoder that continues from this one's state on its own (whisper_kv_cache_seq_cp for a further best-of decoder)."""
        import copy
        o = copy.copy(self)
        o.k, o.v = list(self.k), list(self.v)      # step() replaces the per-layer arrays, it never writes into them
        return o

    def step(self, token):
        W, r, Wr, Wl = self.W, self.r, self.Wr, self.Wl
        x = W["decoder.token_embedding.weight"][int(token)] + W["decoder.positional_embedding"][self.pos]
        for i in range(self.hp.n_text_layer):
            p = f"decoder.blocks.{i}"
            rl = self.rl
            xn = rl(_ln(x, W[p + ".attn_ln.weight"], W[p + ".attn_ln.bias"]))
            q = xn @ Wl[p + ".attn.query.weight"].T + W[p + ".attn.query.bias"]
            self.k[i] = np.vstack([self.k[i], r(xn @ Wl[p + ".attn.key.weight"].T)])
            self.v[i] = np.vstack([self.v[i], r(xn @ Wl[p + ".attn.value.weight"].T + W[p + ".attn.value.bias"])])
            x = x + r(self._att(q, self.k[i], self.v[i])) @ Wr[p + ".attn.out.weight"].T + W[p + ".attn.out.bias"]
            xn = rl(_ln(x, W[p + ".cross_attn_ln.weight"], W[p + ".cross_attn_ln.bias"]))
            q = xn @ Wl[p + ".cross_attn.query.weight"].T + W[p + ".cross_attn.query.bias"]
            x = x + r(self._att(q, self.xk[i], self.xv[i])) @ Wr[p + ".cross_attn.out.weight"].T + W[p + ".cross_attn.out.bias"]
            xn = rl(_ln(x, W[p + ".mlp_ln.weight"], W[p + ".mlp_ln.bias"]))
            g = (_gelu_ggml if self.f16 else _gelu)(xn @ Wl[p + ".mlp.0.weight"].T + W[p + ".mlp.0.bias"])
            x = x + r(g) @ Wr[p + ".mlp.2.weight"].T + W[p + ".mlp.2.bias"]
        self.pos += 1
        x = _ln(x, W["decoder.ln.weight"], W["decoder.ln.bias"])
        return r(x) @ Wr["decoder.token_embedding.weight"].T


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


def n_len_org(n_samples):
    """Mel frames whisper_full may decode from [UPSTREAM-RECALL: log_mel_spectrogram, `mel.n_len_org = 1 + (n_samples +
    stage_2_pad - frame_size) / frame_step` with stage_2_pad = 200, frame 400, step 160; C integer division]: 2999 for a
    full 30 s chunk, not 3000."""
    q = n_samples + 200 - 400
    return 1 + (q // 160 if q >= 0 else -((-q) // 160))


# whisper_full_default_params(WHISPER_SAMPLING_GREEDY) [UPSTREAM-RECALL] -- what TranscribeOptions::default()
# (managers/transcription.rs:184) leaves in force -- plus the two constants of whisper_full_with_state this file needs.
WCPP_PARAMS = dict(temperature=0.0, temperature_inc=0.2, entropy_thold=2.4, logprob_thold=-1.0, no_speech_thold=0.6,
                   best_of=5, length_penalty=-1.0, max_initial_ts=50, n_max_text_ctx=16384,
                   delta_min=10)          # 100 ms: whisper.cpp >= 1.7.6 stops / refuses below it (1 s before)


def _apply_rules(lg, seq, sp, rules, suppress=None, suppress_first=None, max_initial_ts=50, temperature=0.0):
    """whisper_process_logits up to (not including) the probability-mass rule: a float64 copy of `lg`, divided by the
    temperature when it is > 0, with everything that may not be sampled next at -inf."""
    lg = np.array(lg, dtype=np.float64)
    if temperature > 0.0:
        lg = lg / temperature
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
    return lg


def _log_softmax(lg):
    """whisper_compute_logprobs: log-softmax over the entries that are not -inf."""
    m = lg.max()
    if not np.isfinite(m):
        return np.full_like(lg, -np.inf)
    return lg - (m + np.log(np.exp(lg - m).sum()))


def process_logits(lg, seq, sp, rules, suppress=None, suppress_first=None, max_initial_ts=50, temperature=0.0):
    """whisper_process_logits [UPSTREAM-RECALL]: (masked logits, log-probabilities, most probable timestamp token).
    The log-probabilities are normalised over everything allowed BEFORE the probability-mass rule; when that rule
    fires, the text tokens are removed from both arrays afterwards (so the remaining probabilities no longer sum to
    one: whisper_sample_token's discrete_distribution renormalises, a token's `plog` does not)."""
    beg = sp["beg"]
    lg = _apply_rules(lg, seq, sp, rules, suppress, suppress_first, max_initial_ts, temperature)
    lp = _log_softmax(lg)
    tsl = lp[beg:]
    tid = beg + int(np.argmax(tsl)) if np.isfinite(tsl.max()) else beg
    m = tsl.max()
    lse_ts = m + np.log(np.exp(tsl - m).sum()) if np.isfinite(m) else -np.inf
    if lse_ts > lp[:beg].max():
        lg[:beg] = -np.inf
        lp[:beg] = -np.inf
    return lg, lp, tid


def timestamp_rules(lg, seq, sp, rules, suppress=None, suppress_first=None, max_initial_ts=50):
    """Masked copy of the logits `lg` for the next pick, given the tokens `seq` sampled so far in this window.
    Returns (masked logits, index of the most probable timestamp token)."""
    ml, _, tid = process_logits(lg, seq, sp, rules, suppress, suppress_first, max_initial_ts)
    return ml, tid


def decode_window(step_logits, prompt, sp, rules, n_max, seek, seek_end, suppress=None, suppress_first=None,
                  max_initial_ts=50, delta_min=10):
    """One greedy pass over one window, as the library's stage entry point `crispy_asr_decode_timestamps_device` runs it:
    picks under the timestamp rules until EOT, n_max tokens or a timestamp within `delta_min` frames of the end of the
    audio.  `step_logits(token)` feeds one token and returns the next logits.  Returns dict(tokens, tids, plogs,
    result_len, seek_delta, margins).  (whisper_full's own bookkeeping -- failure flags, scores, the temperature ladder
    -- is `decode_temperature` / `whisper_full` below; result_len here is the plain "last timestamp, else everything".)"""
    lg = None
    for t in prompt:
        lg = step_logits(t)
    beg, eot = sp["beg"], sp["eot"]
    toks, tids, margins, plogs = [], [], [], []
    has_ts, seek_delta, result_len = False, 3000, 0
    for i in range(n_max):
        ml, lp, tid = process_logits(lg, toks, sp, rules, suppress, suppress_first, max_initial_ts)
        t = int(np.argmax(ml))
        top2 = np.partition(ml, -2)[-2:]
        margins.append(float(top2[1] - top2[0]))
        toks.append(t)
        plogs.append(float(lp[t]))
        tids.append(tid if t < beg else t)
        if t > beg:
            seek_delta = 2 * (t - beg)
            result_len = i + 1
            has_ts = True
        if t == eot or (has_ts and seek + seek_delta + delta_min >= seek_end):
            if t == eot and result_len == 0:
                result_len = i + 1
            break
        lg = step_logits(t)
    else:
        if result_len == 0:
            result_len = len(toks)
    return dict(tokens=toks, tids=tids, plogs=plogs, result_len=result_len, seek_delta=seek_delta, margins=margins)


# ---------------------------------------------------------------------------------------------------------
# whisper_full's decision logic [UPSTREAM-RECALL: whisper.cpp whisper_full_with_state, whisper_sequence_score,
# whisper_sample_token; source not vendored] -- what decides, on a real recording, whether a window's greedy text is
# kept, re-decoded at a higher temperature, or dropped as silence.
# ---------------------------------------------------------------------------------------------------------
class MT19937:
    """std::mt19937 (32-bit Mersenne twister, `init_genrand` seeding) -- whisper.cpp seeds decoder j with
    std::mt19937(j).  Known answer (the C++ standard's own check): the 10000th output of the default seed 5489 is
    4123659995 (tests/test_oracle_whisper.py)."""

    def __init__(self, seed=5489):
        mt = [0] * 624
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.mt, self.idx = mt, 624

    def _twist(self):
        mt = self.mt
        for i in range(624):
            y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7FFFFFFF)
            mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
        self.idx = 0

    def next_u32(self):
        if self.idx >= 624:
            self._twist()
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def canonical(self):
        """std::generate_canonical<double, 53>(mt19937) as libstdc++ and libc++ both compute it: two draws,
        (x0 + x1 * 2^32) / 2^64 in double arithmetic, a result of 1.0 replaced by the largest double below it."""
        x0, x1 = self.next_u32(), self.next_u32()
        u = (float(x0) + float(x1) * 4294967296.0) / 18446744073709551616.0
        return u if u < 1.0 else float(np.nextafter(1.0, 0.0))


def sample_index(probs, u):
    """std::discrete_distribution<>(probs)(rng) with the uniform variate `u` already drawn [UPSTREAM-RECALL: libstdc++
    -- probabilities divided by their sum, partial sums in double, last one forced to 1.0, std::lower_bound].  Returns
    (index, distance of u to the nearer edge of that index's interval of the cumulative distribution)."""
    p = np.asarray(probs, dtype=np.float64)
    cp = np.cumsum(p / np.cumsum(p)[-1])
    cp[-1] = 1.0
    i = int(np.searchsorted(cp, u, side="left"))
    lo = cp[i - 1] if i > 0 else 0.0
    return i, float(min(u - lo, cp[i] - u))


def sequence_score(toks, plogs, result_len, length_penalty=-1.0):
    """whisper_sequence_score: sum / average of the picks' log-probabilities over the kept tokens, the score the
    best-of decoders are ranked by, and the entropy of the token histogram of the last 32 kept tokens."""
    if result_len == 0:
        return None
    s = float(sum(plogs[:result_len]))
    penalty = float(result_len)
    if length_penalty > 0.0:
        penalty = ((5.0 + penalty) / 6.0) ** length_penalty
    last = toks[max(0, result_len - 32):result_len]
    cnt = {}
    for t in last:
        cnt[t] = cnt.get(t, 0) + 1
    ent = 0.0
    for t in sorted(cnt):                                  # std::map order
        p = cnt[t] / float(len(last))
        ent -= p * np.log(p)
    return dict(sum_logprobs=s, avg_logprobs=s / result_len, score=s / penalty, entropy=float(ent))


def decode_temperature(dc, prompt, sp, rules, n_max, seek, seek_end, t_cur, n_dec, rngs, params, suppress=None,
                       suppress_first=None, no_timestamps=False, beam_size=0):
    """One iteration of whisper_full's temperature ladder over one window: the prompt, then `n_dec` decoders that share
    it (1 at temperature 0, best_of above) sampling in lock step, each with whisper.cpp's completion / failure
    bookkeeping; afterwards the sequences are scored and the best one that did not fail is chosen.
    `dc`: a fresh DecoderCache (fork() gives every further decoder its own copy after the prompt).
    beam_size > 1: the BEAM_SEARCH strategy [UPSTREAM-RECALL: whisper_full_with_state + whisper_sample_token_topk] -- every
    live decoder DRAWS beam_size ids from its distribution (std::discrete_distribution over the probabilities, its own
    generator: in whisper.cpp >= 1.5 the "top-k" are samples, the partial sort's result is unused); the candidates
    (decoder, sequence + id) are sorted by the sum of ALL their log-probabilities (descending; ties: decoder index) and
    dealt to the live decoders in order, a candidate whose token sequence equals the one just dealt being skipped except at
    the first step; the decoder continues from the KV cache of the decoder its candidate came from.
    Returns dict(decoders=[...], best=index or None, no_speech_prob)."""
    beg, eot = sp["beg"], sp["eot"]
    delta_min = params["delta_min"]
    lg0 = None
    for t in prompt:
        lg0 = dc.step(t)
    # before any filtering, at the last prompt position (the only one whose logits the prompt pass produces)
    no_speech_prob = float(np.exp(_log_softmax(np.asarray(lg0, dtype=np.float64))[sp["nosp"]]))
    decs = []
    for j in range(n_dec):
        decs.append(dict(dc=dc if j == 0 else None, toks=[], tids=[], plogs=[], margins=[], has_ts=False,
                         seek_delta=3000, result_len=0, failed=False, completed=False, lg=lg0, score=None, sum_all=0.0))
    for j in range(1, n_dec):
        decs[j]["dc"] = dc.fork()
    for i in range(n_max):
        if beam_size > 1:
            cands = []
            for j, d in enumerate(decs):
                if d["completed"] or d["failed"]:
                    continue
                ml, lp, tid = process_logits(d["lg"], d["toks"], sp, rules, suppress, suppress_first,
                                             params["max_initial_ts"], t_cur)
                pr = np.where(np.isfinite(lp), np.exp(lp), 0.0)
                for _ in range(beam_size):
                    t, gap = sample_index(pr, rngs[j].canonical())
                    # whisper_token_data::plog is a float; sum_logprobs_all a double
                    cands.append(dict(j=j, t=t, plog=float(np.float32(lp[t])), tid=tid if t < beg else t, gap=gap,
                                      sum_all=d["sum_all"] + float(np.float32(lp[t]))))
            cands.sort(key=lambda c: (-c["sum_all"], c["j"]))          # (stable: candidates that tie in both keep their draw order)
            live = [j for j, d in enumerate(decs) if not (d["completed"] or d["failed"])]
            new, cur_c = {}, 0
            for j in live:
                if cur_c >= len(cands):
                    cur_c = 0
                cur = cands[cur_c]
                cur_c += 1
                cur_toks = decs[cur["j"]]["toks"] + [cur["t"]]
                while len(cands) > cur_c and i > 0 and decs[cands[cur_c]["j"]]["toks"] + [cands[cur_c]["t"]] == cur_toks:
                    cur_c += 1
                src = decs[cur["j"]]
                new[j] = dict(dc=src["dc"].fork(), toks=cur_toks, tids=src["tids"] + [cur["tid"]],
                              plogs=src["plogs"] + [cur["plog"]], margins=src["margins"] + [cur["gap"]],
                              has_ts=src["has_ts"], seek_delta=src["seek_delta"], result_len=src["result_len"], failed=False,
                              completed=False, lg=None, score=None, sum_all=cur["sum_all"], parent=cur["j"])
            for j in live:
                decs[j] = new[j]
        for j, d in enumerate(decs if beam_size <= 1 else []):       # GREEDY strategy: one pick per live decoder
            if d["completed"] or d["failed"]:
                continue
            ml, lp, tid = process_logits(d["lg"], d["toks"], sp, rules, suppress, suppress_first,
                                         params["max_initial_ts"], t_cur)
            if t_cur < 1e-6:
                t = int(np.argmax(ml))                        # whisper_sample_token(best): first index of the maximum
                top2 = np.partition(ml, -2)[-2:]
                d["margins"].append(float(top2[1] - top2[0]))
            else:
                pr = np.where(np.isfinite(lp), np.exp(lp), 0.0)
                t, gap = sample_index(pr, rngs[j].canonical())
                d["margins"].append(gap)
            d["toks"].append(t)
            d["plogs"].append(float(lp[t]))
            d["sum_all"] += float(lp[t])
            d["tids"].append(tid if t < beg else t)
        for j, d in enumerate(decs):
            if d["completed"] or d["failed"]:
                continue
            t = d["toks"][-1]
            if t > beg:
                sd_new = 2 * (t - beg)
                if d["has_ts"] and d["seek_delta"] > sd_new and d["result_len"] < i:
                    d["failed"] = True                        # "do not allow to go back in time"
                    continue
                d["seek_delta"], d["result_len"], d["has_ts"] = sd_new, i + 1, True
            if t == eot or (d["has_ts"] and seek + d["seek_delta"] + delta_min >= seek_end):
                if d["result_len"] == 0 and not no_timestamps:
                    if seek + d["seek_delta"] + delta_min >= seek_end:
                        d["result_len"] = i + 1
                    else:
                        d["failed"] = True                    # EOT before any timestamp: nothing to keep
                        continue
                if no_timestamps:
                    d["result_len"], d["seek_delta"] = i + 1, 3000
                d["completed"] = True
                continue
            if i == n_max - 1 and (d["result_len"] == 0 or d["seek_delta"] < 1500):
                d["failed"] = True                            # repetition loop
        if all(d["completed"] or d["failed"] for d in decs):
            break
        for d in decs:
            if not (d["completed"] or d["failed"]):
                d["lg"] = d["dc"].step(d["toks"][-1])
    best, best_score = None, -np.inf
    for j, d in enumerate(decs):
        if d["failed"]:
            continue
        d["kept"] = d["toks"][:d["result_len"]]
        sc = sequence_score(d["toks"], d["plogs"], d["result_len"], params["length_penalty"])
        d["score"] = sc
        if sc is None:
            continue          # whisper_sequence_score returns early: the -inf of the decoder's initialisation stays
        if d["result_len"] > 32 and sc["entropy"] < params["entropy_thold"]:
            d["failed"] = True
            continue
        if best_score < sc["score"]:
            best_score, best = sc["score"], j
    return dict(decoders=decs, best=best, no_speech_prob=no_speech_prob)


# whisper.cpp's `non_speech_tokens` [UPSTREAM-RECALL], suppressed with whisper_full_params.suppress_nst
NON_SPEECH_TOKENS = ['"', "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^", "_", "`", "{", "|", "}", "~",
                     "\u300c", "\u300d", "\u300e", "\u300f", "<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", '("', "((", "))",
                     "(((", ")))", "[[", "]]", "{{", "}}", "\u266a\u266a", "\u266a\u266a\u266a", "\u2669", "\u266a", "\u266b", "\u266c",
                     "\u266d", "\u266e", "\u266f"]


def non_speech_token_ids(vocab):
    """Token ids whisper_process_logits sets to -inf under suppress_nst [UPSTREAM-RECALL]: every string of the list as it
    stands and with a leading space, where the vocabulary (a list of byte strings, index = id) holds it as one token;
    then " -" and " '" (hyphens and quotes are allowed between words, not at the start of one)."""
    first = {}
    for i, t in enumerate(vocab):
        first.setdefault(bytes(t), i)
    out = set()
    for t in NON_SPEECH_TOKENS:
        for cand in (t.encode("utf-8"), (" " + t).encode("utf-8")):
            if cand in first:
                out.add(first[cand])
    for cand in (b" -", b" '"):
        if cand in first:
            out.add(first[cand])
    return sorted(out)


def whisper_full(weights, hp, mel_window, n_samples, prompt, rules, token_text, params=None, n_max=None, suppress=None,
                 suppress_first=None, eot=None, max_windows=1501, f16=False, prev_text=True, decoder_kw=None, encoder=None,
                 initial_prompt=None, past0=None, state=None):
    """whisper_full_with_state over one chunk [UPSTREAM-RECALL], greedy strategy (params["beam_size"] > 1: BEAM_SEARCH, see
    decode_temperature): the seek loop, and per window the
    temperature ladder `temperature, + temperature_inc, .. <= 1.0` -- one greedy decoder at 0, `best_of` sampling
    decoders above; a window's result is accepted unless its best decoder failed (EOT before a timestamp away from the
    end of the audio, repetition to the token limit, entropy of the last 32 tokens below `entropy_thold`) or its average
    log-probability is below `logprob_thold` while `no_speech_prob < no_speech_thold`; at the last temperature it is
    accepted regardless.  A window with `no_speech_prob > no_speech_thold` and an average log-probability below
    `logprob_thold` yields no segment and adds nothing to the conditioning text.  Decoder j draws from MT19937(j),
    seeded anew for every call (whisper.cpp re-seeds decoders >= 1 per call and keeps decoder 0's generator in the state).
    `encoder(mel)` replaces the encoder pass (tests on weight sets whose decoder ignores the audio).
    initial_prompt (whisper_full_params.prompt_tokens): token ids rotated in FRONT of the conditioning text the chunk starts
    with; past0: that text (prompt_past of the state when no_context = false; empty by default: no_context = true).  `state`
    (a dict) receives "prompt_past": the conditioning text the call ends with.
    Returns (segments, kept tokens, windows); a window carries everything the decision used."""
    P = dict(WCPP_PARAMS)
    P.update(params or {})
    sp = special_tokens(hp.n_vocab, eot)
    n_max = hp.n_text_ctx // 2 - 4 if n_max is None else n_max
    delta_min = P["delta_min"]
    seek, seek_end = 0, n_len_org(n_samples)
    segs, kept, wins = [], [], []
    if seek_end < delta_min:                       # "input is too short"
        return segs, kept, wins
    temps = [P["temperature"]]
    if P["temperature_inc"] > 0.0:
        temps = []
        t = np.float32(P["temperature"])
        while t < np.float32(1.0 + 1e-6):
            temps.append(float(t))
            t = np.float32(t + np.float32(P["temperature_inc"]))
    n_best = max(1, int(P["best_of"]))
    beam = int(P.get("beam_size", 0))
    beam = beam if beam > 1 else 0                 # BEAM_SEARCH strategy: beam decoders at temperature 0, best_of above
    rngs = [MT19937(j) for j in range(max(n_best, beam))]
    prompt = list(prompt)
    past = list(initial_prompt or []) + list(past0 or [])
    no_ts = sp["not_"] in prompt
    while len(wins) < max_windows:
        if seek + delta_min >= seek_end:
            break
        if seek > 0 and seek + 500 >= seek_end:
            past = []
        mel = mel_window(seek)
        enc = encoder(mel) if encoder is not None else (encoder_forward_f16 if f16 else encoder_forward)(weights, hp, mel)
        best_id, its, last = 0, [], None
        for it, t_cur in enumerate(temps):
            n_dec = n_best if t_cur > 0.0 else (beam or 1)
            p = list(prompt)
            if prev_text and past and t_cur < 0.5 and P["n_max_text_ctx"] > 0:
                n_take = min(P["n_max_text_ctx"], hp.n_text_ctx // 2, len(past), hp.n_text_ctx - n_max - len(prompt) - 1)
                if n_take > 0:
                    p = [sp["prev"]] + past[len(past) - n_take:] + p
            dc = DecoderCache(weights, hp, enc, f16=f16, **(decoder_kw or {}))
            r = decode_temperature(dc, p, sp, rules, n_max, seek, seek_end, t_cur, n_dec, rngs, P, suppress,
                                   suppress_first, no_ts, beam_size=beam)
            r["temperature"], r["prompt"] = t_cur, p
            its.append(r)
            if r["best"] is not None:
                best_id = r["best"]                          # (best_decoder_id survives an iteration in which every decoder failed)
            last = r
            success = True
            if it != len(temps) - 1:
                d = r["decoders"][best_id] if best_id < len(r["decoders"]) else r["decoders"][0]
                avg = d["score"]["avg_logprobs"] if (d["score"] and not d["failed"]) else -np.inf
                if d["failed"] or (avg < P["logprob_thold"] and r["no_speech_prob"] < P["no_speech_thold"]):
                    success = False
            if success:
                break
        if best_id >= len(last["decoders"]):
            best_id = 0
        d = last["decoders"][best_id]
        toks_cur = d["kept"] if "kept" in d else list(d["toks"])        # a failed decoder's tokens are not cut
        tids_cur = d["tids"][:len(toks_cur)]
        sc = d["score"] if "kept" in d else None
        avg = sc["avg_logprobs"] if sc else -np.inf
        is_no_speech = last["no_speech_prob"] > P["no_speech_thold"] and avg < P["logprob_thold"]
        p = last["prompt"]
        past = list(p[1:len(p) - len(prompt)]) if p[0] == sp["prev"] else []
        if not is_no_speech:
            past += list(d["toks"][:d["result_len"]])
        seek_delta = d["seek_delta"]
        win = dict(seek=seek, prompt=p, tokens=list(toks_cur), tids=list(tids_cur), result_len=len(toks_cur),
                   n_past=d["result_len"],                    # what of it conditions the next window (0 for an early failure)
                   plogs=list(d["plogs"][:len(toks_cur)]), margins=list(d["margins"][:len(toks_cur)]),
                   seek_delta=seek_delta, no_speech_prob=last["no_speech_prob"], avg_logprob=avg,
                   entropy=sc["entropy"] if sc else 0.0, temperature=last["temperature"], decoder=best_id,
                   failed=bool(d["failed"]), is_no_speech=bool(is_no_speech), iterations=its)
        if toks_cur and not is_no_speech:
            segs += window_segments(win, seek, sp, token_text)
            kept += list(toks_cur)
        if len(toks_cur) > 1 and toks_cur[-2] < sp["beg"] and toks_cur[-1] > sp["beg"]:
            seek_delta = min(seek_end - seek, 3000)          # single timestamp ending: nothing after it in this chunk
        win["seek_advance"] = seek_delta
        wins.append(win)
        seek += seek_delta
    if state is not None:
        state["prompt_past"] = list(past)
    return segs, kept, wins


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
                          suppress=None, suppress_first=None, eot=None, max_windows=1501, f16=False, prev_text=True,
                          fallback=False, params=None, decoder_kw=None, encoder=None, **full_kw):
    """`whisper_full` with (fallback=True) or without (temperature_inc = 0: one greedy pass per window, accepted as it
    is) the temperature ladder; everything else -- no-speech rule, failure flags, single-timestamp ending -- applies in
    both.  Returns (segments, all kept tokens, windows)."""
    P = dict(params or {})
    if not fallback:
        P.setdefault("temperature_inc", 0.0)
    return whisper_full(weights, hp, mel_window, n_samples, prompt, rules, token_text, P, n_max, suppress, suppress_first,
                        eot, max_windows, f16, prev_text, decoder_kw, encoder, **full_kw)
