"""Per-kernel totals AND busy spans of a rocprofv3 kernel-trace CSV: for every kernel name the number of dispatches, their summed
duration and the length of the union of their intervals (kernels on two streams overlap: the union says how long the chain was)."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-70:]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
for k, iv in sorted(by.items(), key=lambda kv: -sum(b - a for a, b in kv[1])):
    iv.sort()
    union = 0; cs, ce = iv[0]
    for a, b in iv[1:]:
        if a > ce: union += ce - cs; cs, ce = a, b
        else: ce = max(ce, b)
    union += ce - cs
    print(f"{len(iv):7d} x {sum(b - a for a, b in iv) / len(iv) / 1e3:9.1f} us  sum {sum(b - a for a, b in iv) / 1e6:9.2f} ms  busy {union / 1e6:9.2f} ms  {k}")
print(f"trace span {(t1 - t0) / 1e6:.2f} ms")
