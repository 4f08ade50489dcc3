"""Developer tool: per-stage cycle shares of rn_frame_kernel from the in-kernel stamps of the
diagnostic build (make -C crispy_amd/csrc prof).  Never quote this build's run time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crispy_amd import _native as N
N.LIB_PATH = os.environ.get("PROF_LIB") or os.path.join(os.path.dirname(N.LIB_PATH), "libcrispy_hip_prof.so")
from crispy_amd import synthetic_weights, synth_audio
from crispy_amd.denoise import DenoiseState
B = int(os.environ.get("B", 1024)); T = int(os.environ.get("T", 50))
names = ["0 downsample+lpc+fir", "1 pack+coarse xcorr", "2 Syy prefix+top2", "3 fine search", "4 remove_doubling",
         "5 X window+fft+post", "6 band Ex", "7 P window+fft+post", "8 band Ep/Exp+park P", "9 features",
         "10 dense+vad gru", "11 noise gru", "12 denoise gru+out", "13 pitch filter+gains", "14 taps+inverse fft", "15 OLA+store"]
ds = DenoiseState(synthetic_weights(0), B, 0)
ds.debug_capture(True)
x = synth_audio.batch_torch(B, T, torch.device("cuda:0")); y = torch.empty_like(x)
torch.cuda.synchronize()
ds.process_device(x.data_ptr(), y.data_ptr(), T); ds.synchronize()
acc = np.zeros(24)
n = 0
for b in range(0, B, max(1, B // 64)):
    if b % 10 == 9: continue
    d = ds.debug_read(b)[3824:3848]; acc += d; n += 1
acc /= n * T
tot = acc.sum()
print(f"B={B} T={T}: {tot:.0f} cycles per frame per wave (s_memtime ticks)")
for k, nm in enumerate(names):
    print(f"  {nm:28s} {acc[k]:9.0f}  {100*acc[k]/tot:5.1f}%")
