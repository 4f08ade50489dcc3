"""RNNoise throughput against the stream count, one wave per stream vs the three-wave stage pipeline (CRISPY_RN_WAVES).
python tools/rn_small_batch.py [frames]   -> one line per (streams, waves): ms per call, M stream-frames / s"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from crispy_amd import synthetic_weights, synth_audio
from crispy_amd.denoise import DenoiseState

T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
w = synthetic_weights(0)
dev = torch.device("cuda", 0)
for B in [int(v) for v in os.environ.get("STREAMS", "256,1024,1280").split(",")]:
    x = synth_audio.batch_torch(B, T, dev, first_stream=0, seed=0)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    outs = {}
    for waves in (1, 3):
        os.environ["CRISPY_RN_WAVES"] = str(waves)
        ds = DenoiseState(w, B, 0)
        ds.process_device(x.data_ptr(), y.data_ptr(), T)
        ds.synchronize()
        ds.reset()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            ds.process_device(x.data_ptr(), y.data_ptr(), T)
            ds.synchronize()
            ts.append(time.perf_counter() - t0)
        outs[waves] = y.clone()
        dt = min(ts)
        ds.set_timing(True)
        ds.process_device(x.data_ptr(), y.data_ptr(), T)
        ds.synchronize()
        k_ms, tot_ms = ds.last_kernel_ms()
        ds.set_timing(False)
        print(f"streams {B:5d} waves {waves}: {dt * 1e3:8.2f} ms per {T} frames = {B * T / dt / 1e6:7.2f} M stream-frames/s; "
              f"frame kernels alone {k_ms:7.2f} ms = {k_ms * 1e3 / T:6.1f} us per frame", flush=True)
        ds.close()
    # both forms run the same arithmetic per stage: the LAST call's outputs (same state history) must agree closely
    d = (outs[1] - outs[3]).abs().max().item() / outs[1].abs().max().item()
    print(f"streams {B:5d}: max |waves 1 - waves 3| / peak = {d:.2e}", flush=True)
