"""per-kernel mean of each counter from a rocprofv3 --pmc counter_collection csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"].split("(")[0][:50]
    if pat not in n:
        continue
    agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in agg.items():
    print(n)
    for c, v in d.items():
        v = sorted(v)
        print(f"   {c:36s} n={len(v):3d} median {v[len(v)//2]:.4g}")
