#!/bin/bash
# current library vs scratch/libbase.so: kernel tests, conv microbench, training step
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_base; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -x -q --timeout 120 --timeout-method thread > $O/test.log 2>&1 || { tail -30 $O/test.log; exit 1; }
tail -1 $O/test.log
for i in 1 2; do
UNET_HIP_LIB=scratch/libbase.so timeout -k 10 120 python scratch/bench_conv.py all fwdws 20 > $O/b_$i.log 2>&1
timeout -k 10 120 python scratch/bench_conv.py all fwdws 20 > $O/n_$i.log 2>&1
done
paste <(grep TFLOP $O/b_1.log) <(grep TFLOP $O/n_1.log) <(grep TFLOP $O/b_2.log) <(grep TFLOP $O/n_2.log) | awk '{print $1, $5, $11, $17, $23}'
for i in 1 2; do
UNET_HIP_LIB=scratch/libbase.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('base',d['value'],d['ms_per_step'],d['roofline']['achieved'])"
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('new ',d['value'],d['ms_per_step'],d['roofline']['achieved'])"
done
