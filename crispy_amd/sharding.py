"""Stream sharding across the GPUs of one node.

Streams are independent units: rank r owns the contiguous block [lo, hi) of stream ids and its own
state tensors; there is NO data-path collective.  torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests) is used only for the barrier around the timed region,
the max-over-ranks time and the sum of per-rank frame counters."""
from __future__ import annotations

from typing import Tuple


def shard_range(n_streams: int, rank: int, world: int) -> Tuple[int, int]:
    """Static block partition by stream id; the first n_streams % world ranks get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    q, r = divmod(n_streams, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def reduce_job_stats(elapsed_s: float, frames_done: int, device=None):
    """(max elapsed over ranks, total frames over ranks).  One tiny all-reduce each; latency-bound."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_s), int(frames_done)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    n = torch.tensor([frames_done], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())


def gather_token_ids(ids):
    """The one data-product exchange of the job: every rank's fixed-width greedy token ids
    `[clips, new_tokens]` int32, concatenated in rank order on every rank -> `[world * clips, new_tokens]`.  One
    `all_gather_into_tensor` (RCCL on the GPU box, gloo in the CPU tests); without a process group the tensor itself."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return ids
    world = dist.get_world_size()
    out = torch.empty((world * ids.shape[0],) + tuple(ids.shape[1:]), dtype=ids.dtype, device=ids.device)
    dist.all_gather_into_tensor(out, ids.contiguous())
    return out
