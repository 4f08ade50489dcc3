#!/bin/bash
# Developer tool, run ON THE GPU BOX: LDS activity / bank conflicts / instruction counts of rn_frame_kernel per STAGE,
# from diagnostic builds that end the frame at stamp k (-DRN_STOP_AFTER=k, built into crispy_amd/csrc/build/variants/
# lib_stop<k>.so): counter(k) - counter(k-1) is stage k's share.  One rocprofv3 --pmc pass per build and counter group.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=$PWD/gpurun_out/lds_stage
mkdir -p $out
for k in 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 full; do
  lib=$PWD/crispy_amd/csrc/build/variants/lib_stop$k.so
  [ "$k" = full ] && lib=$PWD/crispy_amd/libcrispy_hip.so
  i=0
  for grp in "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    CRISPY_HIP_LIB=$lib rocprofv3 --pmc $grp --output-format csv -d $out/s${k}_$i -- python3 tools/pmc_frame.py > $out/s${k}_$i.log 2>&1 || echo "stage $k pass $i failed"
  done
  echo "stage $k done"
done
python3 - "$out" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
names = ["0 downsample+lpc+fir", "1 pack+coarse xcorr", "2 Syy prefix+top2", "3 fine search", "4 remove_doubling",
         "5 X window+fft+post", "6 band Ex", "7 P window+fft+post", "8 band Ep/Exp", "9 features",
         "10 dense+vad gru", "11 noise gru", "12 denoise gru+out", "13 pitch filter+gains", "14 inverse fft", "15 OLA+store"]
n_sf = 4096 * 25 * 2
tot = {}
for k in list(range(15)) + ["full"]:
    acc = {}
    for f in glob.glob(os.path.join(out, f"s{k}_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "rn_frame_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    tot[k] = {c: v / n_sf for c, v in acc.items()}
cs = ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"]
print("per stream-frame; LDS-active / conflicts / waits / wave cycles in quad-cycles")
print(f"{'stage':26s} " + " ".join(f"{c[3:]:>18s}" for c in cs))
prev = {c: 0.0 for c in cs}
for i, k in enumerate(list(range(15)) + ["full"]):
    cur = tot.get(k, {})
    print(f"{names[i]:26s} " + " ".join(f"{cur.get(c, 0) - prev[c]:18.1f}" for c in cs))
    prev = {c: cur.get(c, 0) for c in cs}
print(f"{'total':26s} " + " ".join(f"{prev[c]:18.1f}" for c in cs))
PY
