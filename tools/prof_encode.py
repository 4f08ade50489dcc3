"""One encoder pass under rocprofv3: MODEL=tiny|base|... B=64 PREC=0|1 (two passes: the second is the one to read)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
from crispy_amd.asr import WhisperModel
hp = getattr(HParams, os.environ.get("MODEL", "tiny"))()
m = WhisperModel(hp, synthetic_whisper_weights(hp, 0))
m.set_precision(int(os.environ.get("PREC", 1)))
B = int(os.environ.get("B", 64))
melt = torch.randn(B, 3002, hp.n_mels, device="cuda") * 0.3
enc = torch.empty(B, 1500, hp.n_audio_state, device="cuda")
torch.cuda.synchronize()
for _ in range(2):
    m.encode_device(melt.data_ptr(), B, enc.data_ptr()); m.synchronize()
