#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace statistics of the default bench and PMC passes over the RNNoise
# frame kernel.  Counter passes are separate runs with no trace domains, as the pool requires.
#   tools/collect_profiles.sh <tag>        -> gpurun_out/<tag>_*  (+ <tag>_pmc.json summary)
set -u
tag=${1:-prof}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
# the driver's own step counts, so that (average launch) x (launches per step) of the dominant kernel can be held against ms_per_step
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --no-latency --no-cfg45 --no-host-fed --sustain-seconds 0 > $out/${tag}_trace.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $out/${tag}_pmc_$name -- python3 tools/pmc_frame.py > $out/${tag}_pmc_$name.log 2>&1
done
python - "$out" "$tag" <<'PY'
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
B, T, CALLS = 4096, 25, 2            # tools/pmc_frame.py: 2 calls of 25 frames over 4096 streams
acc = {}
for f in glob.glob(os.path.join(out, f"{tag}_pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rn_" not in k:
            continue
        short = "rn_frame_kernel" if "rn_frame_kernel" in k else ("rn_highpass_kernel" if "highpass" in k else ("rn_roll_history_kernel" if "roll" in k else k[:40]))
        d = acc.setdefault(short, {}).setdefault(r["Counter_Name"], [])
        d.append(float(r["Counter_Value"]))
# totals over all dispatches of the run (launch sizes differ: 3, 8, 14 frames per call), and per stream-frame
n_sf = B * T * CALLS
summ = {}
for k, cs in acc.items():
    summ[k] = {"dispatches": max(len(v) for v in cs.values()), "total": {c: sum(v) for c, v in cs.items()},
               "per_stream_frame": {c: sum(v) / n_sf for c, v in cs.items()}}
json.dump({"streams": B, "frames_per_call": T, "calls": CALLS, "stream_frames": n_sf, "kernels": summ},
          open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps(summ.get("rn_frame_kernel", {}).get("per_stream_frame", {}), indent=1))
PY
f=$(find $out/${tag}_trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats.csv && head -12 "$f"
