#!/bin/bash
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/selfreg; mkdir -p $O
timeout -k 10 200 python $GRAFT_REPO_ROOT/scratch/selfreg_bench.py > $O/bench.json 2> $O/bench.err
cat $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o sr --output-format csv -- python $GRAFT_REPO_ROOT/scratch/selfreg_bench.py > $O/bench_prof.json 2> $O/prof.err
echo done
