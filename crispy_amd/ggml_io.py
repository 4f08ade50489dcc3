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


def synthetic_vocab(n_vocab: int) -> list:
    """Stand-in vocabulary for random-init models: ' w<i>' for text tokens, '[_TOK_i]' for specials."""
    eot = 50257 if n_vocab >= 51865 else 50256
    return [f" w{i}".encode() if i < eot else f"[_TOK_{i}]".encode() for i in range(n_vocab)]
