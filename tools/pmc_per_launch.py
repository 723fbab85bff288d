"""every launch of the conv kernels in a rocprofv3 --pmc csv: kernel, grid, counter value (FETCH_SIZE is in units of
32 B... see MI355X guide: reported in KB after the tool's own scaling; printed as is and as MB with the gfx950 x2 for FETCH)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
name = sys.argv[2]
seen = {}
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    if "conv_" not in n or r["Counter_Name"] != name:
        continue
    seen.setdefault((n, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "")), []).append(float(r["Counter_Value"]))
for (n, g), v in seen.items():
    print(f"   in dispatch order (MB): {[round(q * (2 if name == 'FETCH_SIZE' else 1) / 1024) for q in v]}")
    v = sorted(v)
    med = v[len(v) // 2]
    mb = med * (2 if name == "FETCH_SIZE" else 1) / 1024
    print(f"{name:11s} {n:40s} grid {g:>8s} n={len(v):2d} median {med:12.0f} KB -> {mb:8.1f} MB")
