"""Developer tool: timing of the ASR stages at BASELINE cfg 3 size (64 x 30 s clips, Whisper-tiny)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
from crispy_amd.asr import WhisperModel, LogMel
B = int(os.environ.get("B", 64)); NEW = int(os.environ.get("NEW", 32))
hp = HParams.tiny() if os.environ.get("MODEL", "tiny") == "tiny" else HParams.base()
W = synthetic_whisper_weights(hp, 0)
m = WhisperModel(hp, W)
m.set_precision(int(os.environ.get("PREC", 0)))   # 1: f16 operands for the encoder GEMMs
lm = LogMel(hp.n_mels)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
pcm = torch.randn(B, 480000, generator=g, device=dev) * 0.1
melt = torch.zeros(B, 3002, hp.n_mels, device=dev)
enc = torch.empty(B, 1500, hp.n_audio_state, device=dev)
lens = np.full(B, 480000)
torch.cuda.synchronize()
def run_mel():
    lm.compute_device(pcm.data_ptr(), 480000, lens, 0, melt.data_ptr()); lm.synchronize()
def run_enc():
    m.encode_device(melt.data_ptr(), B, enc.data_ptr()); m.synchronize()
for f, name, flops in ((run_mel, "log-mel", None), (run_enc, "encoder", None)):
    f()
    t0 = time.perf_counter()
    for _ in range(3): f()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name:8s} B={B}: {dt*1e3:8.2f} ms  -> {B*30/dt:10.0f} x real time")
    if name == "encoder":
        gf = 36.9e9 if hp.n_audio_state == 384 else 87.4e9
        print(f"         {B*gf/dt/1e12:.1f} TFLOP/s f32 (peak 157)")
    if name == "log-mel":
        print(f"         {B*2.88e6/dt/1e9:.1f} GB/s algorithmic (2.88 MB per clip)")
prompt = [50258, 50259, 50359, 50363]
m.decode_greedy_device(enc.data_ptr(), B, prompt, 2)
t0 = time.perf_counter()
toks, n, lg = m.decode_greedy_device(enc.data_ptr(), B, prompt, NEW)
dt = time.perf_counter() - t0
print(f"decode   B={B}: {dt*1e3:8.2f} ms for {NEW} tokens ({dt/NEW*1e3:.2f} ms/step) -> {B*30/dt:10.0f} x real time")
