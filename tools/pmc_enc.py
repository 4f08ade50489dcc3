"""Developer tool: one mode-1 encoder pass (64 x 30 s clips, Whisper-tiny) for rocprofv3 --pmc runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
from crispy_amd.asr import WhisperModel
B = int(os.environ.get("B", 64))
hp = HParams.tiny()
m = WhisperModel(hp, synthetic_whisper_weights(hp, 0))
m.set_precision(int(os.environ.get("PREC", 1)))
dev = torch.device("cuda:0")
melt = torch.randn(B, 3002, hp.n_mels, device=dev) * 0.3
enc = torch.empty(B, 1500, hp.n_audio_state, device=dev)
torch.cuda.synchronize()
for _ in range(2):
    m.encode_device(melt.data_ptr(), B, enc.data_ptr()); m.synchronize()
