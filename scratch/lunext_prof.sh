#!/bin/bash
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/lunext; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o lunext --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload lunext --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_prof.json 2> $O/stats.err
echo done
