"""Developer tool: per-kernel breakdown of the LAST mode-1 encoder pass in a rocprofv3 --kernel-trace CSV.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/bench_whisper.py ; python tools/enc_breakdown.py DIR"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "layernorm_kernel<" in n]
end, start = idx[-1], idx[-2] + 1
agg, tot = collections.OrderedDict(), 0.0
for r in rows[start:end + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = (r["Kernel_Name"][:r["Kernel_Name"].index("(crispy")] if "(crispy" in r["Kernel_Name"] else r["Kernel_Name"].split("(")[0])[-48:] + " grid=" + r["Grid_Size_X"] + "x" + r["Grid_Size_Y"] + "x" + r["Grid_Size_Z"]
    agg.setdefault(key, []).append(d); tot += d
for k, v in agg.items():
    print(f"{len(v):3d} x {sum(v)/len(v):8.1f} us = {sum(v):8.1f}  {k}")
print("sum of kernels", round(tot, 1), "us; span", (int(rows[end]["End_Timestamp"]) - int(rows[start]["Start_Timestamp"])) / 1e3)
