"""One decode call under rocprofv3: PREC=0|1 B=64 MODEL=tiny|base|... MODE=greedy|ts  (ts = the timestamp-rule path transcribe() uses)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
from crispy_amd.asr import WhisperModel
hp = getattr(HParams, os.environ.get("MODEL", "tiny"))(); m = WhisperModel(hp, synthetic_whisper_weights(hp, 0))
m.set_precision(int(os.environ.get("PREC", 0)))
B = int(os.environ.get("B", 64))
enc = torch.randn(B, 1500, hp.n_audio_state, device="cuda")
torch.cuda.synchronize()
if os.environ.get("MODE", "greedy") == "ts":
    m.decode_timestamps_device(enc.data_ptr(), B, [50258, 50259, 50359], 33)
else:
    m.decode_greedy_device(enc.data_ptr(), B, [50258, 50259, 50359, 50363], 33)
