"""RNNoise gain-network weights: blob layout, loaders and a seeded synthetic generator.

The reference uses the weights built into nnnoiseless 0.5.2 (`DenoiseState::new()`,
reference call site src-tauri/src/audio.rs:229).  That blob is not present in this
environment, so the library takes the weights as an explicit
flat int8 blob in the order of SURVEY.md Appendix A.5 and this module provides

* `LAYERS` / `BLOB_BYTES`  -- the layout (87 503 bytes),
* `synthetic_weights(seed)` -- seeded, mildly contractive int8 weights for tests/bench,
* `load_rnnoise_nu_text(path)` -- parser for the "rnnoise-nu model file version 1" text
  format that `RnnModel::from_read` accepts upstream [UPSTREAM-RECALL, Appendix A.7].
"""
from __future__ import annotations

import numpy as np

# (name, kind, n_in, n_out)   kind: "dense" -> W[in][out], b[out]
#                                   "gru"   -> W[in][3N], U[N][3N], b[3N]  (gate order z, r, h)
LAYERS = (
    ("input_dense", "dense", 42, 24),
    ("vad_gru", "gru", 24, 24),
    ("vad_output", "dense", 24, 1),
    ("noise_gru", "gru", 90, 48),
    ("denoise_gru", "gru", 114, 96),
    ("denoise_output", "dense", 96, 22),
)


def _layer_sizes(kind: str, n_in: int, n_out: int):
    if kind == "dense":
        return (("W", n_in * n_out), ("b", n_out))
    return (("W", n_in * 3 * n_out), ("U", n_out * 3 * n_out), ("b", 3 * n_out))


def blob_offsets():
    """{layer: {part: (offset, count)}} for the flat blob."""
    off = 0
    out = {}
    for name, kind, n_in, n_out in LAYERS:
        parts = {}
        for part, cnt in _layer_sizes(kind, n_in, n_out):
            parts[part] = (off, cnt)
            off += cnt
        out[name] = parts
    return out, off


_OFFSETS, BLOB_BYTES = blob_offsets()
assert BLOB_BYTES == 87503


def synthetic_weights(seed: int = 0, gain: float = 0.7) -> np.ndarray:
    """Seeded int8 weights whose GRUs are mildly contractive (per-layer std = 256*gain/sqrt(K)),
    so the recurrent state stays bounded on any input, like trained RNNoise weights do."""
    rng = np.random.default_rng(seed)
    blob = np.zeros(BLOB_BYTES, dtype=np.int8)
    for name, kind, n_in, n_out in LAYERS:
        k_total = n_in + (n_out if kind == "gru" else 0)
        std = 256.0 * gain / np.sqrt(k_total)
        for part, (off, cnt) in _OFFSETS[name].items():
            if part == "b":
                vals = rng.normal(0.0, 20.0, size=cnt)
            else:
                vals = rng.normal(0.0, std, size=cnt)
            blob[off:off + cnt] = np.clip(np.rint(vals), -127, 127).astype(np.int8)
    return blob


def load_rnnoise_nu_text(path: str) -> np.ndarray:
    """Parse the rnnoise-nu text model format into the flat blob.

    File order is input_dense, vad_gru, noise_gru, denoise_gru, denoise_output, vad_output;
    each layer is `n_in n_out activation` followed by its integer arrays.  The activations
    are fixed by the built-in topology (tanh / ReLU GRUs / sigmoid outputs), a file that
    declares anything else is rejected."""
    with open(path, "r", encoding="ascii") as f:
        header = f.readline().strip()
        if header != "rnnoise-nu model file version 1":
            raise ValueError(f"not an rnnoise-nu model file: {header!r}")
        toks = f.read().split()
    pos = 0

    def take(n):
        nonlocal pos
        if pos + n > len(toks):
            raise ValueError("truncated model file")
        vals = np.array(toks[pos:pos + n], dtype=np.int64)
        pos += n
        return vals

    file_order = ("input_dense", "vad_gru", "noise_gru", "denoise_gru", "denoise_output", "vad_output")
    expect_act = {"input_dense": 0, "vad_gru": 2, "noise_gru": 2, "denoise_gru": 2,
                  "denoise_output": 1, "vad_output": 1}  # 0 tanh, 1 sigmoid, 2 relu
    spec = {name: (kind, n_in, n_out) for name, kind, n_in, n_out in LAYERS}
    blob = np.zeros(BLOB_BYTES, dtype=np.int8)
    for name in file_order:
        kind, n_in, n_out = spec[name]
        hdr = take(3)
        if (int(hdr[0]), int(hdr[1])) != (n_in, n_out):
            raise ValueError(f"{name}: expected {n_in}x{n_out}, file has {hdr[0]}x{hdr[1]}")
        if int(hdr[2]) != expect_act[name]:
            raise ValueError(f"{name}: unsupported activation id {hdr[2]}")
        for part, (off, cnt) in _OFFSETS[name].items():
            vals = take(cnt)
            if vals.min() < -128 or vals.max() > 127:
                raise ValueError(f"{name}.{part}: weight out of int8 range")
            blob[off:off + cnt] = vals.astype(np.int8)
    return blob


def save_rnnoise_nu_text(path: str, blob: np.ndarray) -> None:
    """Inverse of `load_rnnoise_nu_text` (used by the loader round-trip test)."""
    blob = np.asarray(blob, dtype=np.int8)
    assert blob.size == BLOB_BYTES
    file_order = ("input_dense", "vad_gru", "noise_gru", "denoise_gru", "denoise_output", "vad_output")
    act = {"input_dense": 0, "vad_gru": 2, "noise_gru": 2, "denoise_gru": 2,
           "denoise_output": 1, "vad_output": 1}
    spec = {name: (kind, n_in, n_out) for name, kind, n_in, n_out in LAYERS}
    with open(path, "w", encoding="ascii") as f:
        f.write("rnnoise-nu model file version 1\n")
        for name in file_order:
            _, n_in, n_out = spec[name]
            f.write(f"{n_in} {n_out} {act[name]}\n")
            for part, (off, cnt) in _OFFSETS[name].items():
                f.write(" ".join(str(int(v)) for v in blob[off:off + cnt]) + "\n")


EXTREME_KINDS = ("pos127", "neg127", "alt127", "zero", "bias_pos127", "bias_neg127", "heavy_tail", "row_saturating")


def extreme_weights(kind: str, seed: int = 0) -> np.ndarray:
    """Weight blobs at the corners of int8 (parity hardening: `synthetic_weights` keeps the GRUs in the linear part
    of tansig_approx by construction, these do not).

    pos127 / neg127     every weight and bias +127 / -127
    alt127              +-127 alternating along the blob (sign = parity of the flat index)
    zero                all zeros: gates 0.5, candidate 0, gains 0.5
    bias_pos127 / bias_neg127   the seeded synthetic matrices with every bias +127 / -127 (gates pinned near 0.62 / 0.38
                        before the inputs move them)
    heavy_tail          "trained-like": Student-t (3 degrees of freedom) draws scaled so that ~2 % of the weights
                        clip at +-127, biases N(0, 40)
    row_saturating      synthetic matrices, but every 5th output row of every matrix is all +127 and every 7th all
                        -127 (+-8 clamp of tansig_approx and table index 200 on those rows, ordinary rows beside them)
    """
    rng = np.random.default_rng(1000 + seed)
    if kind == "pos127":
        return np.full(BLOB_BYTES, 127, np.int8)
    if kind == "neg127":
        return np.full(BLOB_BYTES, -127, np.int8)
    if kind == "alt127":
        return np.where(np.arange(BLOB_BYTES) % 2 == 0, 127, -127).astype(np.int8)
    if kind == "zero":
        return np.zeros(BLOB_BYTES, np.int8)
    blob = synthetic_weights(seed).copy()
    if kind in ("bias_pos127", "bias_neg127"):
        for name, _, _, _ in LAYERS:
            off, cnt = _OFFSETS[name]["b"]
            blob[off:off + cnt] = 127 if kind == "bias_pos127" else -127
        return blob
    if kind == "heavy_tail":
        for name, k, n_in, n_out in LAYERS:
            for part, (off, cnt) in _OFFSETS[name].items():
                if part == "b":
                    vals = rng.normal(0.0, 40.0, size=cnt)
                else:
                    vals = rng.standard_t(3, size=cnt) * 22.0
                blob[off:off + cnt] = np.clip(np.rint(vals), -127, 127).astype(np.int8)
        return blob
    if kind == "row_saturating":
        for name, k, n_in, n_out in LAYERS:
            rows = n_out * (3 if k == "gru" else 1)
            for part, (off, cnt) in _OFFSETS[name].items():
                if part == "b":
                    continue
                m = blob[off:off + cnt].reshape(-1, rows)        # [K][rows]
                m[:, 4::5] = 127
                m[:, 6::7] = -127
        return blob
    raise ValueError(f"unknown kind {kind!r}")
