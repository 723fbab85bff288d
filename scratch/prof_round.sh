#!/bin/bash
# rocprofv3 passes for the committed profile summaries (run on the GPU box from the repo root)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o unet --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o unet --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/write -o unet --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/write.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_lunext -o lunext --output-format csv -- python $R/bench.py --workload lunext --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_lunext_under_rocprof.json 2> $O/stats_lunext.err
cd $R
timeout -k 10 300 python bench.py --workload lunext > $O/bench_lunext.json 2> $O/bench_lunext.err
timeout -k 10 300 python bench.py > $O/bench_unet.json 2> $O/bench_unet.err
cat $O/bench_lunext.json $O/bench_unet.json
ls $O/*
