"""Developer tool: same-box A/B timing of builds of the library (alternating runs, medians).
    python tools/ab_time.py base=crispy_amd/libcrispy_hip.so v1=crispy_amd/csrc/build/variants/lib_v1.so ...
Each build runs the BASELINE cfg 2 step (4096 streams x 100 frames, device-resident) REPS times, interleaved."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crispy_amd import _native as N, synthetic_weights, synth_audio
from crispy_amd.denoise import DenoiseState
B, T, REPS = int(os.environ.get("B", 4096)), int(os.environ.get("T", 100)), int(os.environ.get("REPS", 7))
w = synthetic_weights(0)
x = synth_audio.batch_torch(B, T, torch.device("cuda:0")); y = torch.empty_like(x)
torch.cuda.synchronize()
builds = []
for spec in sys.argv[1:]:
    name, path = spec.split("=", 1)
    L = N.load_library(os.path.abspath(path))
    ds = DenoiseState(w, B, 0, lib=L)
    for _ in range(2):
        ds.process_device(x.data_ptr(), y.data_ptr(), T); ds.synchronize()
    ds.set_timing(True)
    builds.append((name, ds, [], []))
for r in range(REPS):
    for name, ds, fk, tot in builds:
        ds.process_device(x.data_ptr(), y.data_ptr(), T); ds.synchronize()
        a, b = ds.last_kernel_ms(); fk.append(a); tot.append(b)
ref = None
for name, ds, fk, tot in builds:
    m, mt = statistics.median(fk), statistics.median(tot)
    ref = ref or mt
    print(f"{name:12s} frame-kernel sum {m:7.3f} ms  step {mt:7.3f} ms (min {min(tot):.3f} max {max(tot):.3f})  {100*(mt/ref-1):+5.1f}%  -> {B*T/(mt*1e-3)/100/1e3:6.1f} k streams")
