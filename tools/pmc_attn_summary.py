"""Counters of attn_enc_h_kernel from the passes of tools/pmc_attn_enc.sh, per launch and per 32-key tile of a wave, next
to the issue model of NOTEBOOK 9.10 (480 cycles per tile and wave).  python tools/pmc_attn_summary.py DIR B > json

Units (MI355X_MICROARCH.md, cycle constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count QUAD-cycles (x 4 =
shader cycles); SQ_VALU_MFMA_BUSY_CYCLES counts cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections, csv, glob, json, sys

d, B = sys.argv[1], int(sys.argv[2])
QUAD = ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_VMEM", "SQ_ACTIVE_INST_ANY",
        "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_VMEM")
vals = collections.defaultdict(list)
dur = []
for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
    per_dispatch = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "attn_enc_h_kernel" not in r["Kernel_Name"]:
            continue
        per_dispatch[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for did, cs in per_dispatch.items():
        for c, v in cs.items():
            vals[c].append(v)
for f in sorted(glob.glob(d + "/p*/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "attn_enc_h_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
T, heads = 1500, 6
waves = B * heads * ((T + 127) // 128) * 4
tiles = (T + 31) // 32
wave_tiles = waves * tiles
per_launch = {c: sum(v) / len(v) for c, v in vals.items()}
cyc = {c: (v * 4 if c in QUAD else v) for c, v in per_launch.items()}
out = {
    "kernel": "attn_enc_h_kernel<false>", "clips": B, "heads": heads, "keys": T, "waves_per_launch": waves, "tiles_per_wave": tiles,
    "launches_sampled": {c: len(v) for c, v in vals.items()},
    "duration_us_under_the_profiler": {"mean": sum(dur) / max(len(dur), 1), "min": min(dur) if dur else None, "n": len(dur)},
    "per_launch_raw": per_launch,
    "shader_cycles_per_tile_and_wave": {c: round(v / wave_tiles, 1) for c, v in cyc.items() if c.startswith("SQ_") and "LDS_" not in c[:7] and c != "SQ_WAVES"},
    "lds_array_cycles_per_tile_and_wave": {c: round(per_launch[c] / wave_tiles, 2) for c in ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT") if c in per_launch},
    "issue_model_cycles_per_tile_and_wave": 480,
}
if "GRBM_GUI_ACTIVE" in per_launch and dur:
    t = sum(dur) / len(dur)
    out["effective_clock_ghz"] = round(per_launch["GRBM_GUI_ACTIVE"] / 8 / (t * 1e-6) / 1e9, 3)
    out["effective_clock_note"] = "GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time under the profiler (reads high on dispatches shorter than ~0.3 ms)"
if "SQ_WAVE_CYCLES" in cyc:
    simds = 256 * 4
    out["simd_cycles_per_launch_if_balanced"] = round(cyc["SQ_WAVE_CYCLES"] / 4 / simds)      # four waves share a SIMD
print(json.dumps(out, indent=1))
