"""A Whisper weight set whose decoder says what a script tells it to -- test infrastructure for whisper_full's decision
logic (tests/test_gpu_decision.py).

Every residual branch of the decoder is closed (attention / cross-attention / MLP output projections zero), so the
state that reaches the final LayerNorm is token embedding + positional embedding.  The token embedding is 0.01 R with R
seeded Gaussian, the positional embedding of position p is a combination of rows of R: LN(x) then points along those rows
and logits = g x^ . (0.01 R)^T peak at the scripted tokens -- the logits of position p depend on p alone (plus a 1 %
trace of the token fed in).  g = 100 ("peaky": the scripted token ~ 384 against a crowd whose maximum is ~ 85, log-
probability ~ 0), g = 1 ("flat": 3.84 against ~ 0.85, log-probability ~ -7).  Rows of R that scripts name are made
zero-mean, orthogonal and of norm sqrt(d), so that a two-token row `[(X, 1), (Y, b)]` has the logit gap its weights say.

It is a legitimate model file: the library, the float64 oracle and the f16-operand oracle all run it as they run any
other, the encoder included; only the outcome is predictable."""
from __future__ import annotations

import numpy as np


def scripted_whisper_weights(hp, rows, gain=100.0, seed=0, boost=None, default_token=None):
    """rows: {position: [(token, weight), ...]} -- what the logits of that position point at.  boost: {token: factor} on
    that token's embedding row (a louder <|nospeech|>).  Unscripted positions point at `default_token` (EOT)."""
    from crispy_amd.whisper_weights import synthetic_whisper_weights
    from oracle import whisper_oracle as WO
    sp = WO.special_tokens(hp.n_vocab)
    default_token = sp["eot"] if default_token is None else default_token
    W = synthetic_whisper_weights(hp, seed)
    d, V = hp.n_text_state, hp.n_vocab
    rng = np.random.default_rng([seed, 4242])
    R = rng.standard_normal((V, d))
    named = sorted({t for r in rows.values() for t, _ in r} | {default_token} | set((boost or {}).keys()))
    assert len(named) < d - 1
    # zero-mean orthogonal rows of norm sqrt(d): QR of [1, random columns], the all-ones direction dropped
    Q, _ = np.linalg.qr(np.concatenate([np.ones((d, 1)), rng.standard_normal((d, len(named)))], axis=1))
    for k, t in enumerate(named):
        R[t] = Q[:, k + 1] * np.sqrt(d)
    E = 0.01 * R
    for t, f in (boost or {}).items():
        E[t] *= f
    pos = np.tile(R[default_token], (hp.n_text_ctx, 1))
    for p, r in rows.items():
        pos[p] = sum(w * R[t] for t, w in r)
    W["decoder.token_embedding.weight"] = E.astype(np.float32)
    W["decoder.positional_embedding"] = pos.astype(np.float32)
    W["decoder.ln.weight"] = np.full(d, gain, np.float32)
    W["decoder.ln.bias"] = np.zeros(d, np.float32)
    for i in range(hp.n_text_layer):
        for blk in ("attn.out", "cross_attn.out", "mlp.2"):
            for part in ("weight", "bias"):
                k = f"decoder.blocks.{i}.{blk}.{part}"
                W[k] = np.zeros_like(W[k])
    return W


def script_rows(start, tokens):
    """Rows for a run of picks: the logits of position start + i point at tokens[i] (a token, or a list of (token, weight))."""
    out = {}
    for i, t in enumerate(tokens):
        out[start + i] = t if isinstance(t, list) else [(t, 1.0)]
    return out
