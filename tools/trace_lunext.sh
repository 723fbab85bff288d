#!/bin/bash
# trace_lunext.sh <tag>: kernel timeline of one LUNeXt + clDice training step -> gpurun_out/trace_<tag>/timeline2.txt
tag=${1:-lx}
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$tag; mkdir -p $out; rm -rf $out/rp
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/rp -o t -- python3 $GRAFT_REPO_ROOT/bench.py --workload lunext --steps 6 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/err.log || exit 1
cd $GRAFT_REPO_ROOT
f=$(find $out/rp -name "*kernel_trace.csv" | head -1)
TIMELINE_FIRST_KERNEL=nchw_to_c8_kernel python tools/timeline.py $f full 6 > $out/timeline2.txt
tail -1 $out/timeline2.txt
rm -rf $out/rp
