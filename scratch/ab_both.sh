#!/bin/bash
set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_lunext_gpu.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -2
bash scratch/ab_tree.sh ""
bash scratch/ab_tree.sh "--workload lunext"
