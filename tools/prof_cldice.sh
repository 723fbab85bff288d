#!/bin/bash
# prof_cldice.sh <tag>: rocprofv3 kernel statistics and L2 memory-side traffic (FETCH_SIZE / WRITE_SIZE passes, each in a
# run of its own) of 10 clDice loss calls at batch 32, 256 x 256 -> gpurun_out/prof_cldice_<tag>/
tag=$1; out=$GRAFT_REPO_ROOT/gpurun_out/prof_cldice_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rp -o cl -- python3 $GRAFT_REPO_ROOT/tools/cldice_loop.py 10 > $out/loop.log 2> $out/rocprof.err || exit 1
f=$(find $out/rp -name "*kernel_stats.csv" | head -1); cp $f $out/cldice_kernel_stats.csv; rm -rf $out/rp
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $out/pmc_$pass -o p -- python3 $GRAFT_REPO_ROOT/tools/cldice_loop.py 10 > $out/pmc_$pass.log 2> $out/pmc_$pass.err || exit 1
  f=$(find $out/pmc_$pass -name "*counter_collection.csv" | head -1); cp $f $out/pmc_$pass.csv; rm -rf $out/pmc_$pass
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, json, collections
out = "$out"
calls = 10
stats = {r["Name"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open(out + "/cldice_kernel_stats.csv"))}
def agg(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return d
f, w = agg(out + "/pmc_FETCH_SIZE.csv", "FETCH_SIZE"), agg(out + "/pmc_WRITE_SIZE.csv", "WRITE_SIZE")
res, tot_b, tot_us = {}, 0.0, 0.0
for n, r in stats.items():
    if n not in f:
        continue
    fb = sum(f[n]) * 1024 * 2 / calls          # KB per dispatch; gfx950 tallies 128-B requests at 64 B
    wb = sum(w.get(n, [0])) * 1024 / calls
    us = float(r["TotalDurationNs"]) / 1e3 / calls
    res[n] = {"launches_per_call": int(r["Calls"]) / calls, "us_per_call": round(us, 1), "avg_us": round(float(r["AverageNs"]) / 1e3, 1),
              "fetch_bytes_per_call": int(fb), "write_bytes_per_call": int(wb)}
    if "skel" in n or "morph" in n or "cldice" in n:
        tot_b += fb + wb; tot_us += us
    print(f"{n[:50]:50s} {res[n]['launches_per_call']:5.1f}/call {us:7.1f} us/call  fetch {fb/1e6:7.1f} MB  write {wb/1e6:7.1f} MB")
json.dump({"_note": "10 clDice loss calls (forward of both skeletons + backward), batch 32, 256 x 256; rocprofv3 --kernel-trace "
           "--stats, and --pmc FETCH_SIZE / WRITE_SIZE in separate runs (KB per dispatch; FETCH_SIZE doubled: gfx950 tallies "
           "128-B requests at 64 B); L2 memory-side requests, Infinity-Cache hits included",
           "kernels": res, "loss_kernels_traffic_bytes_per_call": int(tot_b), "loss_kernels_us_per_call": round(tot_us, 1)},
          open(out + "/cldice_traffic.json", "w"), indent=1)
print("clDice kernels: %.1f MB per call, %.1f us of kernel time per call" % (tot_b / 1e6, tot_us))
PY
