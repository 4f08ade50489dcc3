"""Average rocprofv3 --pmc counter_collection values per kernel / grid: python tools/pmc_sum.py DIR [name-regex]."""
import csv, glob, sys, collections, re
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"(gemm_hd_kernel<\d>|attn_enc_h|layernorm_h|attn_dec_x16|gemm_skinny\w*<[^>]*>|vocab_f16)")
for f in sorted(glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        m = pat.search(r["Kernel_Name"])
        if not m: continue
        k = m.group(0) + " g" + r["Grid_Size"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    print("==", f.split("/")[-2])
    for k, d in agg.items():
        print("  ", k, {c: f"{v / cnt[(k, c)]:.3g}" for c, v in d.items()})
