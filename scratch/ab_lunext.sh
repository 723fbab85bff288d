#!/bin/bash
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_lunext; mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py tests/test_lunext_gpu.py -x -q --timeout 120 --timeout-method thread > $O/test.log 2>&1 || { tail -30 $O/test.log; exit 1; }
tail -1 $O/test.log
for i in 1 2 3; do
UNET_HIP_LIB=scratch/libbase.so timeout -k 10 200 python bench.py --workload lunext --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('base',d['value'],d['ms_per_step'])"
timeout -k 10 200 python bench.py --workload lunext --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('new ',d['value'],d['ms_per_step'])"
done
