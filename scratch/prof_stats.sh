#!/bin/bash
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o unet --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/err.log
echo done
