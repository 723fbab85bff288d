#!/bin/bash
set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u -m pytest tests/test_dp_two_ranks_gpu.py -x -q --timeout 150 --timeout-method thread > gpurun_out/dp_test.log 2>&1 || { tail -40 gpurun_out/dp_test.log; exit 1; }; tail -2 gpurun_out/dp_test.log
export UNET_AMD_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for i in 1 2; do
(cd scratch/base_tree && timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null) | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('base (collectives on)',d['value'],d['ms_per_step'])"
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('new  (collectives on)',d['value'],d['ms_per_step'])"
done
unset UNET_AMD_FORCE_COLLECTIVES
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('new  (no group)      ',d['value'],d['ms_per_step'])"
