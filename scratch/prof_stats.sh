#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_now
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o unet --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
cat $O/bench_under_rocprof.json | cut -c1-300
