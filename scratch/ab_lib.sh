#!/bin/bash
# current library vs scratch/libbase.so: kernel + U-Net tests, then the training step of both, alternating
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_lib; mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_lunext_gpu.py -x -q --timeout 120 --timeout-method thread > $O/test.log 2>&1 || { tail -30 $O/test.log; exit 1; }
tail -1 $O/test.log
for i in 1 2 3; do
UNET_HIP_LIB=scratch/libbase.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('base',d['value'],d['ms_per_step'],d['roofline']['achieved'])"
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('new ',d['value'],d['ms_per_step'],d['roofline']['achieved'])"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats -o unet --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/prof.err
echo profiled
