"""One decode call of a catalog-size model under rocprofv3: SPEC=medium:q4_1|large_v3:q5_0|small:f16 FLAVOUR=resident|inflated B=1 NEW=17
(a seeded model file of that shape is written to /tmp first; resident = crispy_asr_load_resident, inflated = crispy_asr_load + mode 1).
Prints ms per generated token of an untraced second call (wall clock)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crispy_amd import _native as N
from crispy_amd.asr import WhisperEngine
from crispy_amd.ggml_io import synthetic_vocab, write_ggml, write_ggml_quantized
from crispy_amd.mel_filters import whisper_mel_filters
from crispy_amd.whisper_weights import HParams, LazyWeights
name, kind = os.environ.get("SPEC", "medium:q4_1").split(":")
hp = getattr(HParams, name)()
path = os.path.join(tempfile.mkdtemp(prefix="crispy_cat_", dir="/tmp"), f"{name}-{kind}.bin")
W = LazyWeights(hp, 0, sensitive=True)
if kind in ("f16", "f32"):
    write_ggml(path, hp, W, whisper_mel_filters(hp.n_mels), synthetic_vocab(hp.n_vocab), f16=(kind == "f16"))
else:
    write_ggml_quantized(path, hp, W, whisper_mel_filters(hp.n_mels), synthetic_vocab(hp.n_vocab), kind, keep=False)
flavour = os.environ.get("FLAVOUR", "resident")
eng = WhisperEngine(path, resident=(flavour == "resident"))
if flavour != "resident":
    eng.set_precision(1)
os.remove(path)
B, NEW = int(os.environ.get("B", 1)), int(os.environ.get("NEW", 17))
enc = torch.randn(B, 1500, hp.n_audio_state, device="cuda") * 0.5
torch.cuda.synchronize()
sp = N.vocab_specials(hp.n_vocab)
prompt = [sp.sot, sp.sot + 1, sp.transcribe, sp.notimestamps]
eng.decode_greedy_device(enc.data_ptr(), B, prompt, NEW)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.decode_greedy_device(enc.data_ptr(), B, prompt, NEW)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t1 = min(ts)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.decode_greedy_device(enc.data_ptr(), B, prompt, 2 * NEW)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"SPEC={name}:{kind} {flavour} B={B}: {1e3 * (min(ts) - t1) / NEW:.3f} ms per generated token (difference of a {2 * NEW}- and a {NEW}-token call)")
