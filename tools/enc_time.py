"""Encoder timing: MODEL=tiny|base B=64 PREC=1 -> ms per encoder pass (median of 5), a checksum of the first two clips' output (comparable across batch sizes) and one of all of it."""
import sys, os, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
from crispy_amd.asr import WhisperModel
hp = getattr(HParams, os.environ.get("MODEL", "tiny"))()
m = WhisperModel(hp, synthetic_whisper_weights(hp, 0))
m.set_precision(int(os.environ.get("PREC", 1)))
B = int(os.environ.get("B", 64))
g = torch.Generator(device="cpu").manual_seed(2)
melt = (torch.randn(B, 3002, hp.n_mels, generator=g) * 0.3).to("cuda")
enc = torch.empty(B, 1500, hp.n_audio_state, device="cuda")
torch.cuda.synchronize()
ts = []
for _ in range(6):
    t0 = time.perf_counter(); m.encode_device(melt.data_ptr(), B, enc.data_ptr()); m.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"MODEL={hp.n_audio_state} B={B} PREC={os.environ.get('PREC', 1)} encoder {np.median(ts[1:]):.3f} ms  crc {zlib.crc32(enc[:2].cpu().numpy().tobytes()):08x}  all {zlib.crc32(enc.cpu().numpy().tobytes()):08x}")
