"""Turns gpurun_out/<tag>_pmc.json (raw sums written by tools/collect_profiles.sh) and the kernel-trace statistics of
the same run into the committed summary profiles/<tag>_pmc.json that bench.py reads.

    python tools/summarize_pmc.py r01d
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
raw = json.load(open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc.json")))
ks = raw["kernels"]
fk = ks["rn_frame_kernel"]
p = {k: round(v, 3) for k, v in fk["per_stream_frame"].items()}

# bench launches 100 frames over 4096 streams as 3 + 4 + 5 + 7 + 10 + 5 x 12 + 11 (11 launches; 3 + 8 + 7 x 12 + 5 = 10 up to r03b): the average duration of a frame-kernel
# launch and the stream-frames it covers give the VALU issue fraction.
avg_ns, calls = None, None
stats = os.path.join(ROOT, "gpurun_out", f"{tag}_kernel_stats.csv")
if os.path.exists(stats):
    for r in csv.DictReader(open(stats)):
        if "rn_frame_kernel" in r["Name"]:
            avg_ns, calls = float(r["AverageNs"]), int(r["Calls"])
frames_per_launch = 100.0 / (11.0 if tag >= "r03c" else 10.0)   # 3 + 4 + 5 + 7 + 10 + 5 x 12 + 11 from r03c on
issue = None
if avg_ns:
    issue = p["SQ_ACTIVE_INST_VALU"] * 4 * 4096 * frames_per_launch / (1024 * avg_ns * 1e-9 * 2.4e9)

out = {
    "command": f"tools/collect_profiles.sh {tag}  (one `rocprofv3 --pmc <group> --output-format csv -- python "
               "tools/pmc_frame.py` run per counter group, no trace domains), then tools/summarize_pmc.py",
    "config": {"streams": raw["streams"], "frames_per_call": raw["frames_per_call"], "calls": raw["calls"],
               "stream_frames": raw["stream_frames"], "launches": "3 + 8 + 12 + 2 frames per call (ramp-up sub-chunks)"},
    "units": "per_stream_frame = counter summed over all rn_frame_kernel dispatches / stream-frames. FETCH_SIZE / "
             "WRITE_SIZE in KiB; gfx950 correction: FETCH_SIZE reports half the bytes read (calibrated in "
             "r01_v2_pmc_hbm.json on rn_highpass_kernel, whose byte counts are known exactly; WRITE_SIZE exact). "
             "SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count in units of 4 cycles.",
    "rn_frame_kernel": {
        "dispatches": fk["dispatches"],
        "per_stream_frame": p,
        "hbm_bytes_per_stream_frame": round((2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024, 1),
        "algorithmic_bytes_per_stream_frame": 3840,
        "valu_active_quads_per_stream_frame": round(p["SQ_ACTIVE_INST_VALU"], 2),
        "insts_per_stream_frame": {"valu": round(p["SQ_INSTS_VALU"]), "salu": round(p["SQ_INSTS_SALU"]),
                                   "lds": round(p["SQ_INSTS_LDS"]), "vmem_rd": round(p["SQ_INSTS_VMEM_RD"]),
                                   "vmem_wr": round(p["SQ_INSTS_VMEM_WR"], 1)},
        "wave_wait_fraction": round(p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"], 3),
        # SQ_LDS_BANK_CONFLICT counts cycles, SQ_ACTIVE_INST_LDS quad-cycles (MI355X_MICROARCH.md): summaries up to r03b
        # divided one by the other without the factor 4 and overstated the conflicts four-fold
        "lds_bank_conflict_fraction_of_lds_active": round(p["SQ_LDS_BANK_CONFLICT"] / (4.0 * p["SQ_ACTIVE_INST_LDS"]), 3),
        "lds_pipe_busy_fraction": round(p["SQ_ACTIVE_INST_LDS"] * 4 * 16 / ((avg_ns or 1) * 1e-9 * 2.16e9 / frames_per_launch), 3) if avg_ns else None,
        "valu_issue_fraction": {
            "definition": "SQ_ACTIVE_INST_VALU * 4 cycles * stream-frames / (1024 SIMDs * kernel time * 2.4 GHz), "
                          f"kernel time from {tag}_bench_kernel_stats.csv "
                          f"({(avg_ns or 0) / 1e6:.3f} ms avg per launch of {frames_per_launch:.2f} frames x 4096 streams)",
            "value": None if issue is None else round(issue, 3)},
    },
}
for k in ("rn_highpass_kernel", "rn_roll_history_kernel"):
    if k in ks:
        out[k] = {"total": {c: v for c, v in ks[k]["total"].items() if c in ("FETCH_SIZE", "WRITE_SIZE")},
                  "dispatches": ks[k]["dispatches"]}
dst = os.path.join(ROOT, "profiles", f"{tag}_pmc.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in out["rn_frame_kernel"].items() if k != "per_stream_frame"}, indent=1))
