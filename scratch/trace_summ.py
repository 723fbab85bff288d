"""median duration per (kernel, grid) from a rocprofv3 kernel trace csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].split("(")[0]
    if pat not in n:
        continue
    key = (n[:60], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v = sorted(v)
    print(k, len(v), "median us %.1f  min %.1f" % (v[len(v) // 2], v[0]))
