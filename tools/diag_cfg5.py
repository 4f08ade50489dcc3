"""Diagnosis of the intermittent cfg5 sub-batch mismatch: batch-of-256 encoder output vs solo runs, with every
intermediate compared twice.  Run after other GPU work in the same process (allocator state matters): WARM=1 runs a
few unrelated allocations first."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crispy_amd.asr import LogMel, WhisperModel
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
dev = torch.device("cuda:0")
if os.environ.get("WARM"):
    junk = [torch.full((n,), float("nan"), device=dev) for n in (1 << 28, 1 << 27, 1 << 26, 3 << 24)]
    torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
hp = HParams.base()
m = WhisperModel(hp, synthetic_whisper_weights(hp, 0))
lm = LogMel(hp.n_mels)
B = 256
for rep in range(int(os.environ.get("REPS", 3))):
    g = torch.Generator(device=dev).manual_seed(1000 + 3072 + rep)
    pcm = torch.randn(B, 480000, generator=g, device=dev) * 0.1
    melt = torch.zeros(B, 3002, hp.n_mels, device=dev)
    enc = torch.empty(B, 1500, hp.n_audio_state, device=dev)
    enc2 = torch.empty_like(enc)
    torch.cuda.synchronize()
    lm.compute_device(pcm.data_ptr(), 480000, np.full(B, 480000), 0, melt.data_ptr()); lm.synchronize()
    melt2 = torch.zeros_like(melt); torch.cuda.synchronize()
    lm.compute_device(pcm.data_ptr(), 480000, np.full(B, 480000), 0, melt2.data_ptr()); lm.synchronize()
    print(rep, "mel batch twice equal:", bool(torch.equal(melt, melt2)))
    m.encode_device(melt.data_ptr(), B, enc.data_ptr()); m.synchronize()
    m.encode_device(melt.data_ptr(), B, enc2.data_ptr()); m.synchronize()
    same = torch.equal(enc, enc2)
    print(rep, "enc batch twice equal:", bool(same))
    if not same:
        d = (enc - enc2).abs().amax(dim=(1, 2)).cpu().numpy()
        print("   clips differing:", np.flatnonzero(d > 0)[:20], d.max())
    for b in (0, 100, 255):
        x = pcm[b].cpu().numpy()
        e1 = m.encode([x])[0]; e1b = m.encode([x])[0]
        m1 = torch.zeros(1, 3002, hp.n_mels, device=dev); torch.cuda.synchronize()
        lm.compute_device(pcm[b:b + 1].contiguous().data_ptr(), 480000, np.full(1, 480000), 0, m1.data_ptr()); lm.synchronize()
        eb = enc[b].cpu().numpy()
        print(rep, b, "solo twice equal:", np.array_equal(e1, e1b), " mel solo == batch:", bool(torch.equal(m1[0], melt[b])),
              " enc solo == batch:", np.array_equal(e1, eb))
        if not np.array_equal(e1, eb):
            dd = np.abs(e1 - eb)
            rows = np.flatnonzero(dd.max(1) > 0)
            print("    rows differing:", rows.size, rows[:8], rows[-8:], "max", dd.max(), "cols", np.flatnonzero(dd.max(0) > 0).size)
