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


# ------------------------------------------------------------------------------------------------------------------
# CPU / NUMA placement of a rank (the 6.5x target at 8 GPUs is about host input staging, not xGMI).
# Everything below reads sysfs only: no HIP call, no torch import -- it runs BEFORE the rank touches its GPU.
# ------------------------------------------------------------------------------------------------------------------
def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-")
            cpus.extend(range(int(lo), int(hi) + 1))
        else:
            cpus.append(int(part))
    return cpus


def gpu_pci_addresses(kfd_root: str = "/sys/class/kfd/kfd/topology/nodes") -> List[str]:
    """PCI addresses (dddd:bb:dd.f) of the GPU nodes KFD lists, in KFD order = HIP device order when no
    *_VISIBLE_DEVICES variable re-orders them.  CPU nodes (simd_count 0) are skipped."""
    out: List[str] = []
    try:
        nodes = sorted((int(n) for n in os.listdir(kfd_root) if n.isdigit()))
    except OSError:
        return out
    for n in nodes:
        props: Dict[str, int] = {}
        try:
            with open(os.path.join(kfd_root, str(n), "properties")) as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    if v.strip().lstrip("-").isdigit():
                        props[k] = int(v)
        except OSError:
            continue
        if props.get("simd_count", 0) <= 0:
            continue
        loc, dom = props.get("location_id", 0), props.get("domain", 0)
        out.append(f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}")
    return out


def bind_rank_to_gpu_numa(local_rank: int, world_local: int = 1, pci_root: str = "/sys/bus/pci/devices",
                          kfd_root: str = "/sys/class/kfd/kfd/topology/nodes") -> Optional[Dict[str, object]]:
    """Restrict this process to the CPUs local to GPU `local_rank` (its PCI device's `local_cpulist`), and within them
    to this rank's share when several ranks sit on one NUMA node -- so that a rank's host buffers are first-touched
    on, and its copy threads run on, the socket its GPU hangs off.  Best effort: returns what it did, or None when the
    topology cannot be read or the binding is refused (containers often pin the CPU set already); never raises.
    Skipped when CRISPY_NO_NUMA_BIND=1 or when *_VISIBLE_DEVICES re-maps the devices (the KFD order is then not the HIP
    order)."""
    try:
        if os.environ.get("CRISPY_NO_NUMA_BIND") == "1":
            return None
        if any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
            return None
        gpus = gpu_pci_addresses(kfd_root)
        if not (0 <= local_rank < len(gpus)):
            return None
        dev = os.path.join(pci_root, gpus[local_rank])
        with open(os.path.join(dev, "local_cpulist")) as f:
            local = _parse_cpulist(f.read())
        node = -1
        try:
            with open(os.path.join(dev, "numa_node")) as f:
                node = int(f.read().strip())
        except (OSError, ValueError):
            pass
        allowed = sorted(set(local) & set(os.sched_getaffinity(0)))
        if not allowed:
            return None
        # ranks whose GPUs share this CPU set split it evenly, in rank order
        peers = []
        for r in range(min(world_local, len(gpus))):
            try:
                with open(os.path.join(pci_root, gpus[r], "local_cpulist")) as f:
                    if _parse_cpulist(f.read()) == local:
                        peers.append(r)
            except OSError:
                pass
        if local_rank in peers and len(peers) > 1 and len(allowed) >= len(peers):
            share = len(allowed) // len(peers)
            k = peers.index(local_rank)
            allowed = allowed[k * share:(k + 1) * share]
        os.sched_setaffinity(0, allowed)
        return {"gpu_pci": gpus[local_rank], "numa_node": node, "cpus": len(allowed), "first_cpu": allowed[0]}
    except Exception:
        return None
