"""Builds and runs tests/c/process_frame_dropin.c: the single-stream process_frame drop-in from plain C against
include/crispy_hip.h.  Shared by tests/test_gpu_c_dropin.py and bench.py's latency_us leg (test / bench infrastructure)."""
from __future__ import annotations

import json
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "process_frame_dropin.c")


def build(out_dir: str) -> str:
    """gcc -std=c99 -pedantic against the header; links the in-tree libcrispy_hip.so (rpath)."""
    exe = os.path.join(out_dir, "process_frame_dropin")
    lib_dir = os.path.join(ROOT, "crispy_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O2", "-D_POSIX_C_SOURCE=199309L",
                    "-I", os.path.join(ROOT, "include"), SRC, "-o", exe, "-L", lib_dir, "-lcrispy_hip",
                    f"-Wl,-rpath,{lib_dir}"], check=True, capture_output=True, text=True)
    return exe


def run(weights: np.ndarray, frames: np.ndarray, timed_calls: int = 0, timeout: float = 300.0):
    """frames [T, 480] f32 (int16 range) -> (out [T, 480], vad [T], latency dict or None)."""
    from crispy_amd import rnn_weights as RW
    frames = np.ascontiguousarray(frames, dtype=np.float32).reshape(-1, 480)
    T = frames.shape[0]
    with tempfile.TemporaryDirectory() as td:
        exe = build(td)
        model = os.path.join(td, "model.txt")
        RW.save_rnnoise_nu_text(model, weights)
        fin, fout = os.path.join(td, "in.f32"), os.path.join(td, "out.f32")
        frames.tofile(fin)
        r = subprocess.run([exe, model, fin, fout, str(T), str(int(timed_calls))], capture_output=True, text=True,
                           timeout=timeout)
        if r.returncode != 0:
            raise RuntimeError(f"process_frame_dropin exited with {r.returncode}: {r.stderr.strip()}")
        raw = np.fromfile(fout, dtype=np.float32)
    assert raw.size == T * 481
    lat = json.loads(r.stdout.strip().splitlines()[-1]) if timed_calls > 0 else None
    return raw[:T * 480].reshape(T, 480), raw[T * 480:], lat
