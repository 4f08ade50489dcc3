"""BASELINE cfg 4: end-to-end denoise -> s16 WAV hand-off -> 48->16 kHz -> log-mel -> Whisper-tiny greedy decode,
B streams x 30 s of 48 kHz audio resident in HBM, one MI355X.  Prints per-stage and total real-time factors."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crispy_amd import synthetic_weights, synth_audio
from crispy_amd.asr import WhisperModel
from crispy_amd.pipeline import DenoiseTranscribePipeline
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights

B = int(os.environ.get("B", 1024)); T = int(os.environ.get("T", 3001)); NEW = int(os.environ.get("NEW", 32))
dev = torch.device("cuda:0")
hp = HParams.tiny()
wm = WhisperModel(hp, synthetic_whisper_weights(hp, 0))
wm.set_precision(int(os.environ.get("PREC", 0)))     # 1: f16-operand encoder / cross K|V
pipe = DenoiseTranscribePipeline(synthetic_weights(0), wm, B)
x = synth_audio.batch_torch(B, T, dev).transpose(0, 1).contiguous()     # [B, T, 480] (BTF)
torch.cuda.synchronize()
prompt = [50258, 50259, 50359, 50363]
for rep in range(2):
    pipe.ds.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks, pcm16 = pipe.run(x, prompt, NEW)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
audio_s = B * (T - 1) * 480 / 48000.0
print(f"cfg4 pipeline B={B} streams x {(T-1)/100:.0f} s: {dt*1e3:.1f} ms total -> {audio_s/dt:,.0f} x real time; "
      f"tokens {toks.shape}, 16 kHz samples per stream {pcm16.shape[1]}")
print("stages (ms): " + ", ".join(f"{k} {v*1e3:.1f}" for k, v in pipe.timings.items()))
print("finite:", bool(torch.isfinite(pcm16).all()), " distinct first tokens:", len(np.unique(toks[:, 0, 0])))
