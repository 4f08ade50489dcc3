"""Committed results of the slow float64 oracle runs of the full-size GPU tests (tests/golden/oracle_cache/*.json).

The GPU suite has a time limit, and a good part of it was numpy: whisper_full's decision logic over 1 300 float64 decoder
steps of a scripted model, three catalog encoders at full depth.  Those oracle results depend on seeded weights and seeded
audio only -- not on anything the GPU computes -- so they are computed once, here on the CPU, by
`python tests/golden/make_oracle_cache.py` (which runs the same functions with CRISPY_ORACLE_CACHE=write), and committed.

A cache entry carries a fingerprint of its inputs; a test whose inputs no longer match it recomputes the oracle (slow, and says
so) instead of trusting stale data.  CRISPY_ORACLE_CACHE=off recomputes everything; =write recomputes and stores."""
import base64
import hashlib
import json
import os

import numpy as np

MAX_ARRAY = 131072      # elements; larger arrays are dropped from a cached structure (catalog encoder rows: 96 x 1280 are kept)
CACHE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_cache")


def fingerprint(*parts) -> str:
    h = hashlib.sha1()
    for p in parts:
        if isinstance(p, np.ndarray):
            a = np.ascontiguousarray(p)
            h.update(str(a.shape).encode() + str(a.dtype).encode())
            flat = a.reshape(-1)
            step = max(1, flat.size // 4096)            # a strided sample: enough to tell two seeded tensors apart
            h.update(np.ascontiguousarray(flat[::step]).tobytes())
        elif isinstance(p, dict):
            for k in sorted(p):
                h.update(str(k).encode())
                h.update(fingerprint(p[k]).encode())
        else:
            h.update(repr(p).encode())
    return h.hexdigest()


def _enc(o):
    if isinstance(o, dict):
        return {str(k): _enc(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_enc(v) for v in o]
    if isinstance(o, np.ndarray):
        if o.size > MAX_ARRAY or (o.ndim >= 1 and o.shape[-1] >= 40000):      # a decoder's last logits row (n_vocab wide) and the like: working state, nothing a test compares
            return {"__dropped__": f"ndarray{tuple(o.shape)}"}
        if o.size > 4096:       # bulk rows (an encoder's output): the raw little-endian bytes, a third of the size of decimal text
            return {"__nd__": o.dtype.str, "shape": list(o.shape), "b64": base64.b64encode(np.ascontiguousarray(o).tobytes()).decode()}
        return {"__nd__": o.dtype.str, "shape": list(o.shape), "data": o.reshape(-1).tolist()}
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.bool_,)):
        return bool(o)
    if isinstance(o, bytes):
        return {"__bytes__": o.hex()}
    if o is None or isinstance(o, (bool, int, float, str)):
        return o
    return {"__dropped__": type(o).__name__}      # live objects an oracle hands back beside its results (a decoder's cache): not data


def _dec(o):
    if isinstance(o, dict):
        if "__bytes__" in o:
            return bytes.fromhex(o["__bytes__"])
        if "__nd__" in o and "b64" in o:
            return np.frombuffer(base64.b64decode(o["b64"]), dtype=np.dtype(o["__nd__"])).reshape(o["shape"]).copy()
        if "__nd__" in o:
            return np.array(o["data"], dtype=np.dtype(o["__nd__"])).reshape(o["shape"])
        return {k: _dec(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_dec(v) for v in o]
    return o


def cached(name: str, fp: str, compute):
    """compute() -> a structure of dicts / lists / tuples / numpy values / bytes (tuples come back as lists)."""
    mode = os.environ.get("CRISPY_ORACLE_CACHE", "")
    path = os.path.join(CACHE_DIR, name + ".json")
    if mode not in ("off", "write") and os.path.exists(path):
        with open(path) as f:
            blob = json.load(f)
        if blob.get("fingerprint") == fp:
            return _dec(blob["value"])
        print(f"[oracle cache] {name}: inputs changed (fingerprint mismatch) -- recomputing the oracle; regenerate with "
              "tests/golden/make_oracle_cache.py")
    value = compute()
    if mode == "write":
        os.makedirs(CACHE_DIR, exist_ok=True)
        text = json.dumps({"fingerprint": fp, "value": _enc(value)}, separators=(",", ":"))
        with open(path + ".tmp", "w") as f:
            f.write(text)
        os.replace(path + ".tmp", path)
        with open(path) as f:                      # what a later run will see: the round trip through JSON
            return _dec(json.load(f)["value"])
    return _dec(json.loads(json.dumps(_enc(value))))      # the same shapes (lists, not tuples) whether cached or not
