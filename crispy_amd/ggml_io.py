"""Writer for whisper.cpp GGML model files -- the container format
`WhisperEngine::load` (managers/transcription.rs:138-141) and `crispy_asr_load` read.

Follows the upstream converter's conventions [UPSTREAM-RECALL]: tensors keep their PyTorch names and
C-order data, dimensions are written innermost first, 2-D+ weights are stored as f16 when ftype = 1
except the positional embeddings, conv biases are stored as [n, 1] f32."""
from __future__ import annotations

import struct

import numpy as np

GGML_MAGIC = 0x67676D6C


def write_ggml(path: str, hp, weights: dict, filters: np.ndarray, vocab: list, f16: bool = True) -> None:
    f32_always = {"encoder.conv1.bias", "encoder.conv2.bias", "encoder.positional_embedding",
                  "decoder.positional_embedding"}
    with open(path, "wb") as f:
        f.write(struct.pack("<I", GGML_MAGIC))
        f.write(struct.pack("<11i", *hp.as_ints(), 1 if f16 else 0))
        filt = np.ascontiguousarray(filters, dtype=np.float32)
        f.write(struct.pack("<2i", filt.shape[0], filt.shape[1]))
        f.write(filt.tobytes())
        f.write(struct.pack("<i", len(vocab)))
        for tok in vocab:
            b = tok if isinstance(tok, bytes) else tok.encode("utf-8")
            f.write(struct.pack("<I", len(b)))
            f.write(b)
        for name, w in weights.items():
            data = np.ascontiguousarray(w, dtype=np.float32)
            if name in ("encoder.conv1.bias", "encoder.conv2.bias"):
                data = data.reshape(-1, 1)
            as_f16 = f16 and data.ndim >= 2 and name not in f32_always
            nb = name.encode("utf-8")
            f.write(struct.pack("<3i", data.ndim, len(nb), 1 if as_f16 else 0))
            for i in range(data.ndim):
                f.write(struct.pack("<i", data.shape[data.ndim - 1 - i]))
            f.write(nb)
            f.write((data.astype(np.float16) if as_f16 else data).tobytes())


# ---- ggml block quantisation (blocks of 32 along the innermost dimension) [UPSTREAM-RECALL ggml-quants] ----
GGML_TYPES = {"f32": 0, "f16": 1, "q4_0": 2, "q4_1": 3, "q5_0": 6, "q5_1": 7, "q8_0": 8}


def quantize_blocks(x: np.ndarray, kind: str):
    """x: float32 [..., 32*k] -> (raw block bytes, de-quantised float32 array of x's shape)."""
    xb = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 32)
    nb = xb.shape[0]
    if kind == "q8_0":
        d = (np.abs(xb).max(1) / 127.0).astype(np.float16)
        dd = d.astype(np.float32)
        q = np.where(dd[:, None] > 0, np.rint(xb / np.where(dd[:, None] > 0, dd[:, None], 1)), 0).clip(-127, 127).astype(np.int8)
        raw = np.concatenate([d.view(np.uint8).reshape(nb, 2), q.view(np.uint8)], axis=1)
        deq = q.astype(np.float32) * dd[:, None]
    elif kind in ("q4_0", "q5_0"):
        levels = 8 if kind == "q4_0" else 16
        idx = np.abs(xb).argmax(1)
        mx = xb[np.arange(nb), idx]
        d = (mx / -levels).astype(np.float16)
        dd = d.astype(np.float32)
        inv = np.where(dd != 0, 1.0 / np.where(dd != 0, dd, 1), 0).astype(np.float32)
        q = np.minimum(2 * levels - 1, (xb * inv[:, None] + (levels + 0.5)).astype(np.int32)).clip(0, 2 * levels - 1).astype(np.uint8)
        deq = (q.astype(np.float32) - levels) * dd[:, None]
        raw = _pack_nibbles(d, None, q, five=(kind == "q5_0"))
    elif kind in ("q4_1", "q5_1"):
        nlev = 15 if kind == "q4_1" else 31
        mn, mx = xb.min(1), xb.max(1)
        d = ((mx - mn) / nlev).astype(np.float16)
        m = mn.astype(np.float16)
        dd, mm = d.astype(np.float32), m.astype(np.float32)
        inv = np.where(dd != 0, 1.0 / np.where(dd != 0, dd, 1), 0).astype(np.float32)
        q = np.minimum(nlev, ((xb - mn[:, None]) * inv[:, None] + 0.5).astype(np.int32)).clip(0, nlev).astype(np.uint8)
        deq = q.astype(np.float32) * dd[:, None] + mm[:, None]
        raw = _pack_nibbles(d, m, q, five=(kind == "q5_1"))
    else:
        raise ValueError(kind)
    return raw.tobytes(), deq.reshape(x.shape).astype(np.float32)


def _pack_nibbles(d, m, q, five: bool) -> np.ndarray:
    nb = q.shape[0]
    parts = [d.view(np.uint8).reshape(nb, 2)]
    if m is not None:
        parts.append(m.view(np.uint8).reshape(nb, 2))
    if five:
        bits = (q >> 4).astype(np.uint32)                      # fifth bit of elements 0..31
        qh = (bits << np.arange(32, dtype=np.uint32)[None, :]).sum(1).astype(np.uint32)
        parts.append(qh.view(np.uint8).reshape(nb, 4))
    lo, hi = q[:, :16] & 0x0F, q[:, 16:] & 0x0F
    parts.append((lo | (hi << 4)).astype(np.uint8))
    return np.concatenate(parts, axis=1)


def write_ggml_quantized(path: str, hp, weights: dict, filters: np.ndarray, vocab: list, kind: str = "q5_0",
                         keep: bool = True, also_positional: bool = False) -> dict:
    """Like the upstream `quantize` tool: 2-D `.weight` matrices whose rows are multiples of 32 become `kind`
    blocks, the rest stays f16/f32.  Returns the weights as the loader will see them (de-quantised; keep=False returns
    an empty dict: catalog-size files).  `weights` needs `.items()` only (crispy_amd.whisper_weights.LazyWeights).
    also_positional: quantise the two positional embeddings too -- the upstream tool never does, the file format allows
    it (the loaders must cope: tests/test_gpu_resident.py)."""
    import io
    seen = {}
    f32_always = {"encoder.conv1.bias", "encoder.conv2.bias", "encoder.positional_embedding",
                  "decoder.positional_embedding"}
    if also_positional:
        f32_always -= {"encoder.positional_embedding", "decoder.positional_embedding"}
    with open(path, "wb") as f:
        f.write(struct.pack("<I", GGML_MAGIC))
        f.write(struct.pack("<11i", *hp.as_ints(), GGML_TYPES[kind]))
        filt = np.ascontiguousarray(filters, dtype=np.float32)
        f.write(struct.pack("<2i", filt.shape[0], filt.shape[1]))
        f.write(filt.tobytes())
        f.write(struct.pack("<i", len(vocab)))
        for tok in vocab:
            b = tok if isinstance(tok, bytes) else tok.encode("utf-8")
            f.write(struct.pack("<I", len(b)))
            f.write(b)
        for name, w in weights.items():
            data = np.ascontiguousarray(w, dtype=np.float32)
            if name in ("encoder.conv1.bias", "encoder.conv2.bias"):
                data = data.reshape(-1, 1)
            quant = (data.ndim == 2 and (name.endswith(".weight") or (also_positional and name.endswith("positional_embedding")))
                     and data.shape[1] % 32 == 0 and name not in f32_always)
            as_f16 = (not quant) and data.ndim >= 2 and name not in f32_always
            if quant:
                payload, deq = quantize_blocks(data, kind)
                ttype = GGML_TYPES[kind]
            elif as_f16:
                payload, deq, ttype = data.astype(np.float16).tobytes(), data.astype(np.float16).astype(np.float32), 1
            else:
                payload, deq, ttype = data.tobytes(), data, 0
            if keep:
                seen[name] = deq.reshape(np.asarray(w).shape)
            nb = name.encode("utf-8")
            f.write(struct.pack("<3i", data.ndim, len(nb), ttype))
            for i in range(data.ndim):
                f.write(struct.pack("<i", data.shape[data.ndim - 1 - i]))
            f.write(nb)
            f.write(payload)
    return seen


def synthetic_vocab(n_vocab: int) -> list:
    """Stand-in vocabulary for random-init models: ' w<i>' for text tokens, '[_TOK_i]' for specials."""
    eot = 50257 if n_vocab >= 51865 else 50256
    return [f" w{i}".encode() if i < eot else f"[_TOK_{i}]".encode() for i in range(n_vocab)]
