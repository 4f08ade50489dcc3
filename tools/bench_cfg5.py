"""BASELINE configs[4]: Whisper-base full transcribe, 8192 streams sharded across 8 GPUs = 1024 x 30 s clips per
GPU (static shard by stream id, no data-path collective: crispy_amd/sharding.py).  Run as one process per GPU:

    python tools/bench_cfg5.py                                  # one GPU = one shard (1/8 of the job)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_cfg5.py

Per shard: PCM resident in HBM -> log-mel -> encoder -> greedy decode of NEW tokens (random-init weights never
emit EOT, so the decode length is fixed), in sub-batches of SUB clips.  Prints one JSON line on rank 0."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from crispy_amd.asr import LogMel, WhisperModel
from crispy_amd.sharding import shard_range
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = int(os.environ.get("LOCAL_RANK", 0))
TOTAL = int(os.environ.get("STREAMS", 1024 * world)); SUB = int(os.environ.get("SUB", 256)); NEW = int(os.environ.get("NEW", 32))
hp = HParams.base() if os.environ.get("MODEL", "base") == "base" else HParams.tiny()
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
lo, hi = shard_range(TOTAL, rank, world)
mine = hi - lo
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
model = WhisperModel(hp, synthetic_whisper_weights(hp, 0), device=local)
lm = LogMel(hp.n_mels, device=local)
g = torch.Generator(device=dev).manual_seed(1000 + rank)
pcm = torch.randn(SUB, 480000, generator=g, device=dev) * 0.1      # every sub-batch reuses one resident buffer
melt = torch.zeros(SUB, 3002, hp.n_mels, device=dev)
enc = torch.empty(SUB, 1500, hp.n_audio_state, device=dev)
lens = np.full(SUB, 480000)
prompt = [50258, 50259, 50359, 50363]


def sub_batch(nb):
    lm.compute_device(pcm.data_ptr(), 480000, lens[:nb], 0, melt.data_ptr(), stream=0)
    lm.synchronize()
    model.encode_device(melt.data_ptr(), nb, enc.data_ptr())
    model.synchronize()
    model.decode_greedy_device(enc.data_ptr(), nb, prompt, NEW)


sub_batch(min(SUB, mine))       # warm-up (allocations, graph capture)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
t0 = time.perf_counter()
done = 0
while done < mine:
    nb = min(SUB, mine - done)
    sub_batch(nb)
    done += nb
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
dt = time.perf_counter() - t0
if world > 1:
    t = torch.tensor([dt], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
if rank == 0:
    print(json.dumps({"workload": f"Whisper-{'base' if hp.n_audio_state == 512 else 'tiny'} full transcribe, {TOTAL} streams x 30 s "
                                  f"over {world} GPU(s), {NEW} greedy tokens per clip, sub-batches of {SUB}",
                      "seconds": dt, "rtfx_whole_job": TOTAL * 30.0 / dt, "clips_per_s": TOTAL / dt, "n_gpus": world}))
if world > 1:
    dist.destroy_process_group()
