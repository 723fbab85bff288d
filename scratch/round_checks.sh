#!/bin/bash
# full GPU test suite, then the PMC pass for MFMA utilisation and the larger bench configurations
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/checks; mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 180 --timeout-method thread > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 200 python bench.py --size 512 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_512.json 2> $O/bench_512.err
cat $O/bench_512.json
timeout -k 10 200 python bench.py --size 1024 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_1024.json 2> $O/bench_1024.err
cat $O/bench_1024.json
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/mfma -o unet --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/mfma.err
ls $O/mfma
