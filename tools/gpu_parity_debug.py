"""Developer tool (run on the GPU box): stage-by-stage comparison of the HIP RNNoise path with
the oracle on a handful of streams.  Not part of the test suite."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import oracle_lib as O
from crispy_amd import synthetic_weights
from crispy_amd.denoise import DenoiseState
from crispy_amd import synth_audio as SA

B = int(os.environ.get("B", 12)); T = int(os.environ.get("T", 60))
w = synthetic_weights(0)
x = SA.batch_np(B, T) * np.float32(32768.0)      # [T,B,480]
x[:, 1] = SA.cfg1_clip(T).reshape(T, 480) * 32768.0
ref_out = np.empty_like(x); ref_taps = np.empty((T, B, 72), np.float32); ref_dbg = np.empty((B, 4304), np.float32)
for b in range(B):
    st = O.OracleDenoiseState(w)
    o, v, tp = st.process(x[:, b], with_taps=True)
    ref_out[:, b] = o; ref_taps[:, b] = tp
    ref_dbg[b] = st.debug()

dev = torch.device("cuda:0")
ds = DenoiseState(w, B, 0)
ds.debug_capture(True)
d_in = torch.from_numpy(x).to(dev); d_out = torch.empty_like(d_in)
d_taps = torch.zeros(T, B, 72, device=dev); d_vad = torch.zeros(T, B, device=dev)
torch.cuda.synchronize()
ds.process_device(d_in.data_ptr(), d_out.data_ptr(), T, d_vad.data_ptr(), d_taps.data_ptr())
ds.synchronize()
out = d_out.cpu().numpy(); taps = d_taps.cpu().numpy()

def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
secs = {"X": (0, 962), "Ex": (962, 984), "lp": (984, 1848), "pitch_pre": (1848, 1849), "P": (1856, 2818),
        "Ep": (2818, 2840), "Exp": (2840, 2862), "XOUT": (2862, 3824), "HP": (3824, 4304)}
print("== last-frame stage errors (max abs err / max abs ref) per stream ==")
for b in range(B):
    d = ds.debug_read(b)
    line = [f"b{b}"]
    for k, (s, e) in secs.items():
        if k == "pitch_pre":
            line.append(f"{k}={int(d[s])}/{int(ref_dbg[b, s])}")
        else:
            line.append(f"{k}={rel(d[s:e], ref_dbg[b, s:e]):.1e}")
    print(" ".join(line))
print("== per-stream over all frames ==")
for b in range(B):
    pm = (taps[:, b, 64] == ref_taps[:, b, 64]).mean()
    sil = ref_taps[:, b, 67].mean()
    fe = rel(taps[:, b, :42], ref_taps[:, b, :42]); ge = np.abs(taps[:, b, 42:64] - ref_taps[:, b, 42:64]).max()
    pg = np.abs(taps[:, b, 65] - ref_taps[:, b, 65]).max(); ve = np.abs(taps[:, b, 66] - ref_taps[:, b, 66]).max()
    oe = rel(out[:, b], ref_out[:, b])
    print(f"b{b}: pitch_match={pm:.3f} silence={sil:.2f} feat_rel={fe:.1e} gain_abs={ge:.1e} pgain_abs={pg:.1e} vad_abs={ve:.1e} out_rel={oe:.2e} out_peak={np.abs(ref_out[:, b]).max():.1f}")
fr = np.abs(out - ref_out).reshape(T, B, -1).max(-1) / (np.abs(ref_out).reshape(T, B, -1).max(-1) + 1e-3)
print("worst frames (t,b,rel):", [(int(i // B), int(i % B), float(fr.flat[i])) for i in np.argsort(fr.ravel())[-5:]])
bad = np.argwhere(taps[:, :, 64] != ref_taps[:, :, 64])
print("pitch mismatches (t,b,gpu,ref):", [(int(t), int(b), int(taps[t, b, 64]), int(ref_taps[t, b, 64])) for t, b in bad[:20]])
# second call continues the state (chunk boundary / history roll)
T2 = 7
x2 = SA.batch_np(B, T + T2)[T:] * np.float32(32768.0)
