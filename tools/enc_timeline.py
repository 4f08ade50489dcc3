"""Developer tool: the kernels of the LAST encoder pass of a rocprofv3 kernel-trace CSV, in launch order, one line each."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "layernorm_kernel<" in n]
end, start = idx[-1], idx[-2] + 1
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:end + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    n = n[:n.index("(crispy")] if "(crispy" in n else n.split("(")[0]
    print(f"{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  {n[-44:]} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
