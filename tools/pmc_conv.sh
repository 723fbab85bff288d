#!/bin/bash
# pmc_conv.sh <layers> <mode>: HBM-side traffic (FETCH_SIZE, WRITE_SIZE; separate passes) of tools/bench_conv.py launches
layers=${1:-conv1_2}; mode=${2:-fwd}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/rp
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out/rp -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py $layers $mode 5 > $out/$c.log 2>&1 || exit 1
  f=$(find $out/rp -name "*counter_collection.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/pmc_per_launch.py $f $c
done
rm -rf $out/rp
