"""Whisper model description: hyper-parameters, tensor inventory (whisper.cpp / OpenAI names, PyTorch
layouts) and a seeded synthetic weight generator.

No Whisper weights exist in this environment; tests and the benchmark use
random-init models of the real architecture.  The same named tensors are what
`crispy_asr_set_tensor` expects and what a GGML model file carries (Appendix B.5)."""
from __future__ import annotations

import zlib
from collections import OrderedDict
from dataclasses import asdict, dataclass

import numpy as np


@dataclass(frozen=True)
class HParams:
    n_vocab: int = 51865
    n_audio_ctx: int = 1500
    n_audio_state: int = 384
    n_audio_head: int = 6
    n_audio_layer: int = 4
    n_text_ctx: int = 448
    n_text_state: int = 384
    n_text_head: int = 6
    n_text_layer: int = 4
    n_mels: int = 80

    @staticmethod
    def tiny():
        return HParams()

    @staticmethod
    def base():
        return HParams(n_audio_state=512, n_audio_head=8, n_audio_layer=6, n_text_state=512, n_text_head=8,
                       n_text_layer=6)

    # the models of the reference's catalog (src-tauri/src/managers/model.rs:74-148) at full depth
    @staticmethod
    def small():
        return HParams(n_audio_state=768, n_audio_head=12, n_audio_layer=12, n_text_state=768, n_text_head=12,
                       n_text_layer=12)

    @staticmethod
    def medium():
        return HParams(n_audio_state=1024, n_audio_head=16, n_audio_layer=24, n_text_state=1024, n_text_head=16,
                       n_text_layer=24)

    @staticmethod
    def large_v3():
        return HParams(n_vocab=51866, n_audio_state=1280, n_audio_head=20, n_audio_layer=32, n_text_state=1280,
                       n_text_head=20, n_text_layer=32, n_mels=128)

    @staticmethod
    def large_v3_turbo():
        return HParams(n_vocab=51866, n_audio_state=1280, n_audio_head=20, n_audio_layer=32, n_text_state=1280,
                       n_text_head=20, n_text_layer=4, n_mels=128)

    def as_ints(self):
        return [int(v) for v in asdict(self).values()]


# special tokens of the multilingual vocabulary
TOK_EOT, TOK_SOT, TOK_EN, TOK_TRANSCRIBE, TOK_NOTIMESTAMPS = 50257, 50258, 50259, 50359, 50363


def tensor_shapes(hp: HParams) -> "OrderedDict[str, tuple]":
    d, dt = hp.n_audio_state, hp.n_text_state
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["encoder.conv1.weight"] = (d, hp.n_mels, 3)
    s["encoder.conv1.bias"] = (d,)
    s["encoder.conv2.weight"] = (d, d, 3)
    s["encoder.conv2.bias"] = (d,)
    s["encoder.positional_embedding"] = (hp.n_audio_ctx, d)

    def attn(prefix, dim):
        s[prefix + ".query.weight"] = (dim, dim)
        s[prefix + ".query.bias"] = (dim,)
        s[prefix + ".key.weight"] = (dim, dim)
        s[prefix + ".value.weight"] = (dim, dim)
        s[prefix + ".value.bias"] = (dim,)
        s[prefix + ".out.weight"] = (dim, dim)
        s[prefix + ".out.bias"] = (dim,)

    def ln(prefix, dim):
        s[prefix + ".weight"] = (dim,)
        s[prefix + ".bias"] = (dim,)

    def mlp(prefix, dim):
        s[prefix + ".mlp.0.weight"] = (4 * dim, dim)
        s[prefix + ".mlp.0.bias"] = (4 * dim,)
        s[prefix + ".mlp.2.weight"] = (dim, 4 * dim)
        s[prefix + ".mlp.2.bias"] = (dim,)

    for i in range(hp.n_audio_layer):
        p = f"encoder.blocks.{i}"
        ln(p + ".attn_ln", d)
        attn(p + ".attn", d)
        ln(p + ".mlp_ln", d)
        mlp(p, d)
    ln("encoder.ln_post", d)
    s["decoder.token_embedding.weight"] = (hp.n_vocab, dt)
    s["decoder.positional_embedding"] = (hp.n_text_ctx, dt)
    for i in range(hp.n_text_layer):
        p = f"decoder.blocks.{i}"
        ln(p + ".attn_ln", dt)
        attn(p + ".attn", dt)
        ln(p + ".cross_attn_ln", dt)
        attn(p + ".cross_attn", dt)
        ln(p + ".mlp_ln", dt)
        mlp(p, dt)
    ln("decoder.ln", dt)
    return s


def sinusoids(length: int, channels: int) -> np.ndarray:
    """Whisper's fixed encoder position table: concat(sin, cos) of geometric timescales."""
    inc = np.log(10000.0) / (channels // 2 - 1)
    inv = np.exp(-inc * np.arange(channels // 2))
    t = np.arange(length)[:, None] * inv[None, :]
    return np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32)


def synthetic_tensor(name: str, shape, seed: int = 0, sensitive: bool = False) -> np.ndarray:
    """One tensor of the seeded random-init model (its own stream: seed, crc32(name)) -- see synthetic_whisper_weights."""
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    if name == "encoder.positional_embedding":
        w = sinusoids(shape[0], shape[1])
    elif name.endswith("_ln.weight") or name.endswith("ln_post.weight") or name == "decoder.ln.weight":
        w = 1.0 + 0.1 * rng.standard_normal(shape)
    elif name.endswith(".bias"):
        w = 0.02 * rng.standard_normal(shape)
    elif name == "decoder.token_embedding.weight":
        w = 0.05 * rng.standard_normal(shape)
    elif name == "decoder.positional_embedding":
        w = 0.02 * rng.standard_normal(shape)
    else:  # linear / conv: fan-in scaling keeps activations O(1) through the stack
        fan_in = int(np.prod(shape[1:]))
        w = rng.standard_normal(shape) / np.sqrt(fan_in)
        if sensitive and ".cross_attn." in name:
            w = w * (8.0 if (".query." in name or ".key." in name) else 4.0)
    return np.ascontiguousarray(w, dtype=np.float32)


def synthetic_whisper_weights(hp: HParams, seed: int = 0, sensitive: bool = False) -> "OrderedDict[str, np.ndarray]":
    """Seeded random-init model; every tensor has its own stream (seed, crc32(name)), so the result
    does not depend on generation order and is identical on every machine (numpy PCG64).

    sensitive=True sharpens and amplifies the decoder's cross-attention (query / key x 8, value / out x 4).  With plain
    fan-in scaling the soft-max over the 1500 encoder rows is nearly uniform, the cross-attention output is the same
    average for every clip and the greedy picks barely depend on the audio ("distinct first tokens: 3" over 1024
    streams) -- token equality then checks the decoder, not the chain.  With peaky attention the
    picks differ from clip to clip and from step to step, so an error anywhere between the PCM and the logits moves
    them."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in tensor_shapes(hp).items():
        out[name] = synthetic_tensor(name, shape, seed, sensitive)
    return out


class LazyWeights:
    """`items()` of a seeded model, one tensor alive at a time: catalog-size files (medium 0.77 G, large-v3 1.55 G
    parameters) are written without holding 3 - 6 GB of f32 on the host (tools/bench_resident.py)."""

    def __init__(self, hp: HParams, seed: int = 0, sensitive: bool = False):
        self.hp, self.seed, self.sensitive = hp, seed, sensitive

    def items(self):
        for name, shape in tensor_shapes(self.hp).items():
            yield name, synthetic_tensor(name, shape, self.seed, self.sensitive)
