"""Developer tool: groups a rocprofv3 --kernel-trace CSV by (kernel, grid, workgroup size) and prints call counts and
median / min / total durations -- the per-launch view of a decode step.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dec -- python tools/prof_decode.py
    python tools/trace_by_grid.py gpurun_out/dec
"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    wg = int(r["Workgroup_Size_X"])
    key = (k.split("(")[0][-34:], int(r["Grid_Size_X"]) // wg, r["Grid_Size_Y"], wg)
    d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:16]:
    v.sort()
    print(k, len(v), "med %.1f" % v[len(v) // 2], "min %.1f" % v[0], "sum %.0f us" % sum(v))
