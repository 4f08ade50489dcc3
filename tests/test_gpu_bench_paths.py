"""bench.py's other code paths on the ONE GPU of the test box, each as a fresh child process:
  * the RCCL path with a single rank (CRISPY_BENCH_FORCE_DIST=1: process group init over "nccl" = RCCL, barriers, the
    all-reduces of the job statistics, the per-rank all-gather, and -- cfg 5 -- the all_gather_into_tensor of the token
    ids), which the driver's N = 1 run never enters;
  * the host-fed flavour of cfg 2 (page-locked host buffers through crispy_rn_process);
  * the cfg 4 workload with two pipelines in flight, against its serial form (same token checksum)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    last = [ln for ln in r.stdout.splitlines() if ln.strip()][-1]          # the JSON line is the LAST stdout line
    return json.loads(last)


QUICK = ["--no-asr", "--no-latency", "--no-live-traffic", "--no-cpu-baseline", "--no-cfg45", "--no-host-fed", "--sustain-seconds", "0"]


def test_cfg2_through_rccl_with_one_rank():
    from crispy_amd.launch import free_port
    plain = _bench(["--steps", "2", "--warmup", "1", *QUICK])
    assert plain["process_group"] is None and plain["n_gpus"] == 1
    line = _bench(["--steps", "2", "--warmup", "1", *QUICK], {"CRISPY_BENCH_FORCE_DIST": "1", "MASTER_PORT": str(free_port())})
    assert line["process_group"] == "nccl (RCCL)"
    assert line["n_gpus"] == 1 and len(line["per_rank_ms"]) == 1           # as RCCL's process group reports it
    assert line["config"]["output_finite"] and line["value"] > 8000       # the north-star floor, by a wide margin
    assert line["roofline"]["kernel"] == "rn_frame_kernel" and 0 < line["roofline"]["frac"] < 1
    # RCCL initialised must not cost the step its stream concurrency (NOTEBOOK.md section 5: hardware queues)
    assert line["ms_per_step"] < 1.25 * plain["ms_per_step"], (line["ms_per_step"], plain["ms_per_step"])


def test_cfg5_gathers_the_token_ids_over_rccl():
    from crispy_amd.launch import free_port
    args = ["--workload", "cfg5", "--clips", "8", "--steps", "1", "--warmup", "1", "--new-tokens", "4"]
    solo = _bench(args)
    line = _bench(args, {"CRISPY_BENCH_FORCE_DIST": "1", "MASTER_PORT": str(free_port())})
    assert line["process_group"] == "nccl (RCCL)" and line["n_gpus"] == 1
    t = line["transcript_ids"]
    assert t["gathered_with"] == "all_gather_into_tensor" and t["shape"] == [8, 4]
    assert t["checksum"] == solo["transcript_ids"]["checksum"] and t["rank0_first_clip"] == solo["transcript_ids"]["rank0_first_clip"]


def test_cfg2_host_fed_flavour():
    line = _bench(["--steps", "2", "--warmup", "1", "--host-fed", "--streams", "1024", "--frames", "50", *QUICK])
    assert "HOST-FED" in line["metric"] and "host-resident" in line["data"]
    assert line["config"]["output_finite"] and line["value"] > 8000
    # ... and as a leg of the default line: f32 and int16 sample transport beside the HBM-resident value
    quick = [a for a in QUICK if a != "--no-host-fed"]
    line = _bench(["--steps", "2", "--warmup", "1", *quick])
    hf = line["host_fed"]
    assert "error" not in hf, hf
    assert hf["f32"]["bytes_per_sample"] == 4 and hf["s16"]["bytes_per_sample"] == 2
    assert 8000 < hf["value"] < line["value"] and hf["pcie_gbps_each_way"] > 5
    print(f"host-fed: f32 {hf['f32']['value']:.0f} streams at {hf['f32']['pcie_gbps_each_way']:.1f} GB/s each way, "
          f"s16 {hf['s16']['value']:.0f} at {hf['s16']['pcie_gbps_each_way']:.1f} = {hf['s16_over_f32']:.2f} x")
    assert hf["s16_over_f32"] > 1.3, hf


def test_cfg4_two_pipelines_in_flight_equal_the_serial_form():
    args = ["--workload", "cfg4", "--pipe-streams", "128", "--steps", "2", "--warmup", "1", "--new-tokens", "8"]
    serial = _bench([*args, "--pipe-depth", "1"])
    piped = _bench([*args, "--pipe-depth", "2"])
    for line in (serial, piped):
        assert line["config"]["new_tokens"] == 8 and line["config"]["precision_mode"] == 1 and line["value"] > 500
    assert piped["config"]["tokens_checksum"] == serial["config"]["tokens_checksum"]
    assert set(piped["config"]["serial_stage_ms"]) == {"denoise", "resample", "logmel", "encoder", "decode"}
