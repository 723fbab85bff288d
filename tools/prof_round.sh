#!/bin/bash
# prof_round.sh <tag>: bench line + rocprofv3 kernel stats of the same command -> gpurun_out/prof_<tag>/
tag=$1; out=gpurun_out/prof_$tag; mkdir -p $out
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench_unet.json 2> $out/bench_unet.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp -o unet -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-second-workload > $GRAFT_REPO_ROOT/$out/bench_unet_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/rocprof.err || exit 1
cd $GRAFT_REPO_ROOT
f=$(find $out/rp -name "*kernel_stats.csv" | head -1); cp $f $out/unet_kernel_stats.csv
find $out/rp -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $out/rp -name "*.db" -delete
head -30 $out/unet_kernel_stats.csv | cut -c1-150
python -c "import json; d=json.load(open('$out/bench_unet.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernels'])"
