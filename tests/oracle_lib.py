"""ctypes binding of oracle/liboracle.so -- the CPU parity oracle (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_LIB = None

FRAME = 480
TAPS = 72
DBG_FLOATS = 4304
WEIGHT_BYTES = 87503


def build():
    subprocess.run(["make", "-s", "-C", _ORACLE_DIR], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        f32p = C.POINTER(C.c_float)
        L.rno_create.restype = C.c_void_p
        L.rno_create.argtypes = [C.c_void_p, C.c_size_t]
        L.rno_destroy.argtypes = [C.c_void_p]
        L.rno_reset.argtypes = [C.c_void_p]
        L.rno_process_frame.restype = C.c_float
        L.rno_process_frame.argtypes = [C.c_void_p, f32p, f32p]
        L.rno_process_frames.argtypes = [C.c_void_p, f32p, f32p, C.c_int, f32p]
        L.rno_last_taps.argtypes = [C.c_void_p, f32p]
        L.rno_last_pitch_margin.argtypes = [C.c_void_p]
        L.rno_last_pitch_margin.restype = C.c_float
        L.rno_last_debug.argtypes = [C.c_void_p, f32p]
        L.rno_forward_transform.argtypes = [f32p, f32p, f32p]
        L.rno_inverse_transform.argtypes = [f32p, f32p, f32p]
        L.rno_biquad.argtypes = [f32p, f32p, f32p, C.c_int]
        L.rno_band_energy.argtypes = [f32p, f32p, f32p]
        L.rno_interp_band_gain.argtypes = [f32p, f32p]
        L.rno_dct.argtypes = [f32p, f32p]
        L.rno_half_window.argtypes = [f32p]
        L.rno_tansig_approx.restype = C.c_float
        L.rno_tansig_approx.argtypes = [C.c_float]
        L.rno_sigmoid_approx.restype = C.c_float
        L.rno_sigmoid_approx.argtypes = [C.c_float]
        L.rno_pitch_downsample.argtypes = [f32p, f32p]
        L.rno_pitch_search.restype = C.c_int
        L.rno_pitch_search.argtypes = [f32p, f32p, C.c_int, C.c_int]
        L.rno_remove_doubling.restype = C.c_float
        L.rno_remove_doubling.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                          C.c_int, C.c_float]
        L.rno_compute_rnn.argtypes = [C.c_void_p, f32p, f32p, f32p, f32p]
        L.wlo_logmel.restype = C.c_int
        L.wlo_logmel.argtypes = [f32p, C.c_int, f32p, C.c_int, f32p]
        L.wlo_logmel_window.restype = C.c_int
        L.wlo_logmel_window.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_int, f32p]
        _LIB = L
    return _LIB


def fp(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleDenoiseState:
    """`DenoiseState` of the oracle: one stream, `process_frame(out, in) -> vad`."""

    def __init__(self, weights: np.ndarray):
        w = np.ascontiguousarray(weights, dtype=np.int8)
        assert w.size == WEIGHT_BYTES
        self._w = w
        self._h = lib().rno_create(w.ctypes.data_as(C.c_void_p), w.size)
        if not self._h:
            raise RuntimeError("rno_create failed")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().rno_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def reset(self):
        lib().rno_reset(self._h)

    def process_frame(self, frame: np.ndarray):
        x = np.ascontiguousarray(frame, dtype=np.float32)
        assert x.size == FRAME
        out = np.empty(FRAME, dtype=np.float32)
        vad = lib().rno_process_frame(self._h, fp(out), fp(x))
        return out, float(vad)

    def taps(self):
        t = np.empty(TAPS, dtype=np.float32)
        lib().rno_last_taps(self._h, fp(t))
        return t

    def debug(self):
        d = np.empty(DBG_FLOATS, dtype=np.float32)
        lib().rno_last_debug(self._h, fp(d))
        return d

    def pitch_margin(self) -> float:
        """Smallest gap at a comparison that decided the last frame's pitch index (oracle/rnnoise_oracle.c: margin_note)."""
        return float(lib().rno_last_pitch_margin(self._h))

    def process(self, x: np.ndarray, with_taps: bool = False, with_margin: bool = False):
        """x: [n_frames, 480] -> out [n_frames, 480], vad [n_frames] (, taps [n_frames, 72]) (, pitch margins [n_frames])."""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, FRAME)
        n = x.shape[0]
        out = np.empty_like(x)
        vad = np.empty(n, dtype=np.float32)
        if not with_taps:
            lib().rno_process_frames(self._h, fp(out), fp(x), n, fp(vad))
            return out, vad
        taps = np.empty((n, TAPS), dtype=np.float32)
        margin = np.empty(n, dtype=np.float32)
        for t in range(n):
            out[t], vad[t] = self.process_frame(x[t])
            taps[t] = self.taps()
            margin[t] = self.pitch_margin()
        return (out, vad, taps, margin) if with_margin else (out, vad, taps)


def oracle_logmel(x: np.ndarray, filters: np.ndarray, seek: int = 0) -> np.ndarray:
    """oracle/logmel_oracle.c: one clip (<= 480000 samples) -> [n_mel, 3000], frames [seek, seek + 3000)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    f = np.ascontiguousarray(filters, dtype=np.float32)
    out = np.empty((f.shape[0], 3000), dtype=np.float32)
    rc = lib().wlo_logmel_window(fp(x), x.size, fp(f), f.shape[0], int(seek), fp(out))
    if rc != 0:
        raise ValueError("wlo_logmel_window rejected its arguments")
    return out
