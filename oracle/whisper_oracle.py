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
and tests/test_oracle_whisper.py.  Known whisper.cpp deviations that are NOT restated: ggml's f16 GELU
lookup table and f16 x f16 matmul operands (this oracle is the exact-arithmetic version of the graph).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import numpy as np
from scipy.special import erf


def _gelu(x):
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def _ln(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def _f64(w):
    return {k: v.astype(np.float64) for k, v in w.items()}


def _mha(xq, xkv, W, prefix, n_head, causal=False):
    q = xq @ W[prefix + ".query.weight"].T + W[prefix + ".query.bias"]
    k = xkv @ W[prefix + ".key.weight"].T
    v = xkv @ W[prefix + ".value.weight"].T + W[prefix + ".value.bias"]
    Tq, D = q.shape
    dh = D // n_head
    out = np.empty_like(q)
    for h in range(n_head):
        sl = slice(h * dh, (h + 1) * dh)
        s = (q[:, sl] @ k[:, sl].T) / np.sqrt(dh)
        if causal:
            s = s + np.triu(np.full((Tq, k.shape[0]), -np.inf), k=1 + k.shape[0] - Tq)
        s = s - s.max(-1, keepdims=True)
        p = np.exp(s)
        p /= p.sum(-1, keepdims=True)
        out[:, sl] = p @ v[:, sl]
    return out @ W[prefix + ".out.weight"].T + W[prefix + ".out.bias"]


def encoder_forward(weights, hp, mel, upto_layer=None):
    """mel: [n_mels, 3000] -> [1500, d] (float64)."""
    W = _f64(weights)
    x = mel.astype(np.float64)
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
