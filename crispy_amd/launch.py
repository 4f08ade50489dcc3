"""One process per GPU, started by a parent that has made no GPU call.

`bench.py --gpus N` without a launcher's environment (no WORLD_SIZE) calls `spawn_ranks`: N fresh child
processes of the same script, each with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, exactly what
`python -m torch.distributed.run --nproc-per-node N` would hand them.  No `os.exec*` anywhere: the parent stays a
plain supervisor, relays rank 0's stdout (so that the JSON line is the last thing on the parent's stdout), sends the
other ranks' stdout to stderr and exits non-zero as soon as any rank fails (the rest of the job is then stopped by
pid, never by pattern).  Nothing here imports torch or touches HIP."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    """Environment of one rank: one node, local rank == rank, rendezvous on 127.0.0.1 (the container's hostname may
    not resolve)."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    # the host driver of this pool only supports dmabuf IPC; RCCL across processes needs it
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _pump(src, dst, keep: Optional[List[str]] = None, prefix: str = "") -> None:
    for line in iter(src.readline, ""):
        if keep is not None:
            keep.append(line)
        dst.write(prefix + line)
        dst.flush()
    src.close()


def spawn_ranks(script: str, argv: Sequence[str], world: int, timeout_s: Optional[float] = None,
                extra_env: Optional[Dict[str, str]] = None) -> int:
    """Run `python script argv...` as `world` ranks; returns the job's exit code (0 only if every rank returned 0).

    Rank 0's stdout is relayed line by line to this process's stdout; if its last non-empty line is not the last
    thing relayed (it always is, ranks > 0 write to stderr) nothing is re-ordered -- the relay is verbatim."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs: List[subprocess.Popen] = []
    pumps: List[threading.Thread] = []
    rank0_lines: List[str] = []
    base = dict(os.environ)
    if extra_env:
        base.update(extra_env)
    for r in range(world):
        p = subprocess.Popen([sys.executable, script, *argv], env=rank_env(r, world, port, base),
                             stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1)
        procs.append(p)
        t = threading.Thread(target=_pump, daemon=True,
                             args=(p.stdout, sys.stdout if r == 0 else sys.stderr, rank0_lines if r == 0 else None,
                                   "" if r == 0 else f"[rank {r}] "))
        t.start()
        pumps.append(t)
    deadline = None if timeout_s is None else time.monotonic() + timeout_s
    rc = 0
    alive = set(range(world))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench launcher: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for o in sorted(alive):
                    procs[o].terminate()          # by pid: only the children started above
        if deadline is not None and time.monotonic() > deadline and alive:
            rc = rc or 124
            print("bench launcher: timeout; stopping all ranks", file=sys.stderr)
            for o in sorted(alive):
                procs[o].terminate()
            deadline = time.monotonic() + 15
            timeout_s = None
            for o in sorted(alive):
                try:
                    procs[o].wait(15)
                except subprocess.TimeoutExpired:
                    procs[o].kill()
            break
        time.sleep(0.05)
    for p in procs:
        try:
            p.wait(15)
        except subprocess.TimeoutExpired:
            p.kill()
    for t in pumps:
        t.join(5)
    return rc
