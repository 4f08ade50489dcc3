"""world_size-2 CPU test (gloo) of the multi-GPU plumbing: block partition of streams and the
job-level reductions bench.py uses.  The data path has no collective."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from crispy_amd.sharding import reduce_job_stats, shard_range


def test_shard_range_partitions_exactly():
    for n in (1, 7, 8, 4096, 8193):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            for a, b in zip(parts, parts[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(10, rank, world)
    # each rank "processes" its own streams: per-stream result depends on the stream id only
    mine = np.arange(lo, hi) * 2.0
    dist.barrier()
    elapsed, frames = reduce_job_stats(0.5 + rank, (hi - lo) * 100)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine.tolist())
    # the job's data product: fixed-width token ids of every rank, one all_gather_into_tensor, rank order
    from crispy_amd.sharding import gather_token_ids
    ids = torch.arange(3 * 4, dtype=torch.int32).reshape(3, 4) + 1000 * rank
    table = gather_token_ids(ids)
    q.put((rank, elapsed, frames, gathered, table.tolist()))
    dist.destroy_process_group()


def test_two_rank_gloo_job_stats_and_assembly():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = np.concatenate([np.arange(12).reshape(3, 4) + 1000 * r for r in range(2)]).tolist()
    for rank, elapsed, frames, gathered, table in res:
        assert elapsed == 1.5            # max over ranks
        assert frames == 1000            # all 10 streams x 100 frames
        assert sum(gathered, []) == [2.0 * i for i in range(10)]   # shards tile the stream ids in order
        assert table == want             # [world * clips, new_tokens], rank order, on every rank


def test_gather_token_ids_without_a_process_group_is_the_identity():
    from crispy_amd.sharding import gather_token_ids
    ids = torch.arange(6, dtype=torch.int32).reshape(2, 3)
    assert gather_token_ids(ids) is ids


def test_numa_binding_reads_the_kfd_and_pci_topology(tmp_path):
    """crispy_amd.launch.bind_rank_to_gpu_numa against a fake sysfs: two CPU nodes, two GPUs on different NUMA nodes;
    the rank is pinned to the CPUs its GPU's PCI device lists (intersected with what the process may use), nothing is
    pinned when the topology is missing or a *_VISIBLE_DEVICES variable re-maps the devices, and nothing ever raises."""
    from crispy_amd import launch
    kfd, pci = tmp_path / "kfd", tmp_path / "pci"
    for n, (simd, loc) in enumerate([(0, 0), (0, 0), (1024, 0x0500), (1024, 0x8508)]):      # bus 05 dev 0; bus 85 dev 1
        d = kfd / str(n)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nlocation_id {loc}\ndomain 0\n")
    assert launch.gpu_pci_addresses(str(kfd)) == ["0000:05:00.0", "0000:85:01.0"]
    mine = sorted(os.sched_getaffinity(0))
    for addr, cpus, node in (("0000:05:00.0", f"{mine[0]}", 0), ("0000:85:01.0", f"{mine[-1]}", 1)):
        d = pci / addr
        d.mkdir(parents=True)
        (d / "local_cpulist").write_text(cpus + "\n")
        (d / "numa_node").write_text(f"{node}\n")
    before = os.sched_getaffinity(0)
    try:
        got = launch.bind_rank_to_gpu_numa(1, 2, str(pci), str(kfd))
        assert got == {"gpu_pci": "0000:85:01.0", "numa_node": 1, "cpus": 1, "first_cpu": mine[-1]}
        assert os.sched_getaffinity(0) == {mine[-1]}
    finally:
        os.sched_setaffinity(0, before)
    assert launch.bind_rank_to_gpu_numa(5, 8, str(pci), str(kfd)) is None          # no such GPU
    assert launch.bind_rank_to_gpu_numa(0, 1, str(tmp_path / "nope"), str(tmp_path / "nope")) is None
    os.environ["HIP_VISIBLE_DEVICES"] = "1"
    try:
        assert launch.bind_rank_to_gpu_numa(0, 2, str(pci), str(kfd)) is None
    finally:
        del os.environ["HIP_VISIBLE_DEVICES"]
    assert launch._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


# ---- bench.py's own launcher (crispy_amd/launch.py): `python bench.py --gpus 2` with no WORLD_SIZE starts 2 ranks ----
import json
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_gpus2_spawns_two_ranks_and_relays_rank0_json():
    """The launcher path of `bench.py --gpus N` (gloo dry run: rank env plumbing, barrier, reductions, JSON relay)."""
    r = _bench(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--streams", "10", "--frames", "7"])
    assert r.returncode == 0, r.stderr[-2000:]
    last = [ln for ln in r.stdout.splitlines() if ln.strip()][-1]
    line = json.loads(last)                                   # rank 0's JSON line is the LAST stdout line
    assert line["n_gpus"] == 2                                # as the process group reports it, not as --gpus says
    assert len(line["per_rank_ms"]) == 2 and all(ms > 0 for ms in line["per_rank_ms"])
    assert line["frames_total"] == 2 * 10 * 7 * 3             # weak scaling: both ranks' frames are summed
    assert line["ms_per_step"] * 3 >= max(line["per_rank_ms"]) - 1e-6   # max over ranks
    assert line["data"] == "dry-run" and line["value"] is None


def test_bench_launcher_fails_when_a_rank_fails():
    r = _bench(["--gpus", "2", "--dry-run", "--steps", "2"], {"CRISPY_BENCH_DRY_FAIL_RANK": "1"})
    assert r.returncode == 3
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())      # no JSON line from a failed job
    assert "rank 1 exited with 3" in r.stderr


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = _bench(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "3", "RANK": "0"}, drop=())
    assert r.returncode == 2 and "refusing" in r.stderr


def test_bench_under_an_external_launcher_is_a_rank_not_a_parent():
    """What the driver does: torch.distributed.run starts the ranks; bench.py must not spawn again."""
    from crispy_amd.launch import free_port
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    jl = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(jl) == 1 and json.loads(jl[0])["n_gpus"] == 2


def test_rank_env_contents():
    from crispy_amd.launch import rank_env
    e = rank_env(3, 8, 12345, base={})
    assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"], e["MASTER_PORT"]) == \
        ("3", "3", "8", "127.0.0.1", "12345")
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
