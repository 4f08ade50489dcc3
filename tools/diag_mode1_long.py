"""Diagnostic: precision mode 1, long greedy decodes -- the logit of every pick against the teacher-forced f16 oracle,
step by step (where does the GPU leave the oracle?).  python tools/diag_mode1_long.py [n_new] [sensitive 0/1] [precision]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crispy_amd import synth_audio
from crispy_amd.asr import WhisperModel
from crispy_amd.mel_filters import whisper_mel_filters
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
from oracle import whisper_oracle as WO
from tests import oracle_lib as O

n_new = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sens = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 1
hp = HParams.tiny()
W = synthetic_whisper_weights(hp, 0, sensitive=sens)
m = WhisperModel(hp, W)
m.set_precision(prec)
x = synth_audio.clip16k_np(int(os.environ.get("CLIP_SEED", 7)), int(os.environ.get("CLIP_N", 300000)))
prompt = [50258, 50259, 50359, 50363]
mel = O.oracle_logmel(x, whisper_mel_filters(80))
enc_ref = (WO.encoder_forward_f16 if prec else WO.encoder_forward)(W, hp, mel)
enc = m.encode([x])
print("encoder max err / peak", np.abs(enc[0] - enc_ref).max() / np.abs(enc_ref).max())
for feed in ("gpu_enc", "oracle_enc"):
    e = enc[0] if feed == "gpu_enc" else enc_ref.astype(np.float32)
    d = torch.from_numpy(np.ascontiguousarray(e[None])).cuda()
    torch.cuda.synchronize()
    toks, n, lg = m.decode_greedy_device(d.data_ptr(), 1, prompt, n_new)
    dc = WO.DecoderCache(W, hp, e.astype(np.float64), f16=bool(prec))
    for t in prompt[:-1]:
        dc.step(t)
    tok = prompt[-1]
    print(feed)
    for i in range(n_new):
        l = dc.step(tok)
        g = int(toks[0, i])
        best = int(np.argmax(l))
        top2 = np.partition(l, -2)[-2:]
        print(f"  step {i:3d} pos {i + 4:3d} gpu {g:6d} oracle {best:6d} margin {top2[1] - top2[0]:.4f} "
              f"gpu_logit {lg[0, i]:.5f} oracle_logit_of_gpu_tok {l[g]:.5f} err {lg[0, i] - l[g]:+.5f}")
        tok = g
