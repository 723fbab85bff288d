#!/bin/bash
# wide (4-wave) kernel vs band kernel on the same device
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_wide; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "conv_gemm or dgrad" > $O/test.log 2>&1 || { tail -30 $O/test.log; exit 1; }
tail -3 $O/test.log
SH=all
for i in 1 2; do
UNET_HIP_LIB=scratch/libnowide.so timeout -k 10 120 python scratch/bench_conv.py $SH fwdws 20 > $O/band_$i.log 2>&1
timeout -k 10 120 python scratch/bench_conv.py $SH fwdws 20 > $O/wide_$i.log 2>&1
done
for f in band_1 wide_1 band_2 wide_2; do echo == $f; grep TFLOP $O/$f.log; done
