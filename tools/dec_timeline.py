"""Developer tool: the kernels of the LAST decode step of a rocprofv3 kernel-trace CSV, in order, with start offsets and gaps."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "vocab" in r["Kernel_Name"]]
a, b = idx[-3] + 1, idx[-2] + 1
t0 = int(rows[a]["Start_Timestamp"]); prev_end = None
for r in rows[a:b + 3]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:8.1f} us  +{gap:5.1f} gap  {(e - s) / 1e3:6.1f} us  {r['Kernel_Name'].split('(')[0][-50:]} g={r['Grid_Size_X']}x{r['Grid_Size_Y']} wg={r['Workgroup_Size_X']}")
    prev_end = e
