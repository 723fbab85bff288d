#!/bin/bash
# prof_lunext.sh <tag>: rocprofv3 kernel stats of the LUNeXt + clDice bench -> gpurun_out/prof_<tag>/lunext_kernel_stats.csv
tag=$1; out=gpurun_out/prof_$tag; mkdir -p $out
timeout -k 10 300 python bench.py --workload lunext --steps 20 --warmup 5 > $out/bench_lunext.json 2> $out/bench_lunext.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rpl -o lunext -- python3 $GRAFT_REPO_ROOT/bench.py --workload lunext --steps 16 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_lunext_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/rocprof_lunext.err || exit 1
cd $GRAFT_REPO_ROOT
f=$(find $out/rpl -name "*kernel_stats.csv" | head -1); cp $f $out/lunext_kernel_stats.csv
rm -rf $out/rpl
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/lunext_kernel_stats.csv")))
steps=20; tot=0
for r in rows:
    ms=float(r['TotalDurationNs'])/1e6/steps; tot+=ms
    if ms>0.03: print(f"{r['Name'][:58]:58s} calls/step {int(r['Calls'])/steps:5.1f} ms/step {ms:6.3f} avg_us {float(r['AverageNs'])/1e3:7.1f}")
print('sum ms/step', round(tot,3), 'kernels/step', sum(int(r['Calls']) for r in rows)/steps)
PY
