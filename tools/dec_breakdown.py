"""Per-kernel breakdown of a decode call from a rocprofv3 kernel-trace CSV: aggregates by kernel name over the LAST call."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict(); tot = 0.0
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = r["Kernel_Name"]
    key = n.split("(")[0][-60:] + " g=" + r["Grid_Size_X"] + "x" + r["Grid_Size_Y"] + "x" + r["Grid_Size_Z"] + " wg=" + r["Workgroup_Size_X"]
    agg.setdefault(key, []).append(d); tot += d
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{len(v):5d} x {sum(v)/len(v):8.1f} us = {sum(v):9.1f}  {k}")
print("sum of kernels", round(tot, 1), "us; span", (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3)
