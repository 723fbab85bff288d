"""profiles/rNN_traffic.json from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py"""
import csv, json, sys, collections
fetch_csv, write_csv, out = sys.argv[1:4]
def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        agg[n].append(float(r["Counter_Value"]))
    return agg
f = per_kernel(fetch_csv, "FETCH_SIZE")
w = per_kernel(write_csv, "WRITE_SIZE")
res = {}
for n in sorted(f, key=lambda k: -sum(f[k])):
    if n not in w or len(f[n]) < 6:
        continue
    fb = sum(f[n]) / len(f[n]) * 1024 * 2          # KB; gfx950 tallies 128-B requests at 64 B
    wb = sum(w[n]) / len(w[n]) * 1024
    res[n] = {"launches": len(f[n]), "fetch_bytes_per_launch": int(fb), "write_bytes_per_launch": int(wb),
              "traffic_bytes_per_launch": int(fb + wb)}
json.dump({"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `bench.py --steps 3 --warmup 2`; "
           "counters are KB per dispatch; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); memory-side "
           "requests of the L2, Infinity-Cache hits included", "kernels": res}, open(out, "w"), indent=1)
for n, d in list(res.items())[:12]:
    print(f"{n:40s} {d['launches']:4d}  fetch {d['fetch_bytes_per_launch']/1e6:8.1f} MB  write {d['write_bytes_per_launch']/1e6:8.1f} MB")
