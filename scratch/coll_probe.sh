#!/bin/bash
set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_dp_two_ranks_gpu.py tests/test_unet_gpu.py tests/test_lunext_gpu.py tests/test_selfreg_gpu.py -x -q --timeout 150 --timeout-method thread > gpurun_out/coll_tests.log 2>&1 || { tail -40 gpurun_out/coll_tests.log; exit 1; }
tail -2 gpurun_out/coll_tests.log
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null > gpurun_out/coll_out.txt; echo "$1 stdout lines: $(wc -l < gpurun_out/coll_out.txt)"; python3 -c "import json,sys;d=json.loads(open('gpurun_out/coll_out.txt').read());print('$1',d['value'],d['ms_per_step'],d['roofline']['achieved'])"; }
run "no group"
export UNET_AMD_FORCE_COLLECTIVES=1
run "group   "
