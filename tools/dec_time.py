"""Decode-step timing: python tools/dec_time.py  (env: MODEL=tiny|base B=64 NEW=32 PREC=0|1|2).
Prints ms per decode call, ms per position and a checksum of the tokens (A/B of decode-step changes: must not move)."""
import sys, os, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
from crispy_amd.asr import WhisperModel
hp = getattr(HParams, os.environ.get("MODEL", "tiny"))()
m = WhisperModel(hp, synthetic_whisper_weights(hp, 0))
m.set_precision(int(os.environ.get("PREC", 0)))
B = int(os.environ.get("B", 64)); NEW = int(os.environ.get("NEW", 32))
g = torch.Generator(device="cpu").manual_seed(1)
enc = torch.randn(B, 1500, hp.n_audio_state, generator=g).to("cuda")
torch.cuda.synchronize()
prompt = [50258, 50259, 50359, 50363]
toks = m.decode_greedy_device(enc.data_ptr(), B, prompt, NEW)
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    toks = m.decode_greedy_device(enc.data_ptr(), B, prompt, NEW)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
t = float(np.median(ts))
print(f"MODEL={hp.n_audio_state} B={B} NEW={NEW} PREC={os.environ.get('PREC', 0)}"
      f" decode {t:.2f} ms  {t / (NEW + len(prompt)):.3f} ms/position  crc {zlib.crc32(np.ascontiguousarray(toks[0]).tobytes()):08x}")
