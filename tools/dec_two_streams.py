"""Developer experiment: one decode call over B clips against the same clips split over N handles (own stream, own captured
graphs) driven from N host threads.  python tools/dec_two_streams.py   (env: B=64 NEW=32 PREC=1 MODEL=tiny)"""
import sys, os, time, threading, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
from crispy_amd.asr import WhisperModel
hp = getattr(HParams, os.environ.get("MODEL", "tiny"))()
W = synthetic_whisper_weights(hp, 0)
B = int(os.environ.get("B", 64)); NEW = int(os.environ.get("NEW", 32)); PREC = int(os.environ.get("PREC", 1))
g = torch.Generator(device="cpu").manual_seed(1)
enc = torch.randn(B, 1500, hp.n_audio_state, generator=g).to("cuda")
prompt = [50258, 50259, 50359, 50363]
esz = enc[0].numel() * 4
for n in (1, 2, 4):
    ms = [WhisperModel(hp, W) for _ in range(n)]
    for m in ms: m.set_precision(PREC)
    nb = B // n
    out = [None] * n
    def work(k):
        out[k] = ms[k].decode_greedy_device(enc.data_ptr() + k * nb * esz, nb, prompt, NEW)[0]
    def run():
        th = [threading.Thread(target=work, args=(k,)) for k in range(n)]
        for t in th: t.start()
        for t in th: t.join()
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    toks = np.concatenate([np.asarray(o) for o in out])
    print(f"{n} handle(s) x {nb} clips: {np.median(ts):.2f} ms per {B}-clip decode, crc {zlib.crc32(np.ascontiguousarray(toks).tobytes()):08x}", flush=True)
    for m in ms: m.close()
