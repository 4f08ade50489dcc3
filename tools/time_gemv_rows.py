"""Developer tool (GPU box): a generated token's step of a catalog width at 1 ... 128 rows, matrix-vector kernels (all rows
since round 6) against the skinny kernels (developer build, CRISPY_ASR_GEMV=0).  SPEC=medium:q4_1|small:dense FLAVOUR=resident."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights, LazyWeights
from crispy_amd.asr import WhisperModel, WhisperEngine
from tests.native_variant import library_variant

name, kind = os.environ.get("SPEC", "small:dense").split(":")
hp = getattr(HParams, name)()
prompt = [50258, 50259, 50359, 50363]


def make():
    if kind == "dense":
        m = WhisperModel(hp, synthetic_whisper_weights(hp, 3))
        m.set_precision(1)
        return m
    from crispy_amd.ggml_io import synthetic_vocab, write_ggml_quantized
    from crispy_amd.mel_filters import whisper_mel_filters
    path = os.path.join(tempfile.gettempdir(), f"rows-{name}-{kind}.bin")
    if not os.path.exists(path):
        write_ggml_quantized(path, hp, LazyWeights(hp, 3), whisper_mel_filters(hp.n_mels), synthetic_vocab(hp.n_vocab), kind)
    return WhisperEngine(path, resident=True)


def run(m, label):
    for B in (1, 4, 8, 16, 32, 64, 128):
        enc = torch.randn(B, 1500, hp.n_audio_state, device="cuda") * 0.8
        torch.cuda.synchronize()
        ts = {}
        for n in (9, 33):
            m.decode_greedy_device(enc.data_ptr(), B, prompt, n)
            t0 = time.perf_counter()
            for _ in range(3):
                m.decode_greedy_device(enc.data_ptr(), B, prompt, n)
            ts[n] = (time.perf_counter() - t0) / 3
        print(f"{label:7s} {B:4d} rows: {(ts[33] - ts[9]) / 24 * 1e3:7.3f} ms per position", flush=True)
        del enc


m = make()
run(m, "gemv")
m.close()
with library_variant("dev", {"CRISPY_ASR_GEMV": "0"}):
    m = make()
    run(m, "skinny")
    m.close()
