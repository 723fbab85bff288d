#!/bin/bash
# one-stream vs two-stream U-Net step + tests
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_unet_gpu.py tests/test_fullsize_gpu.py -x -q --timeout 180 --timeout-method thread 2>&1 | tail -2
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$1',d['value'],d['ms_per_step'],d['config']['loss_last_step'])"; }
for i in 1 2; do
export UNET_AMD_WGRAD_STREAM=0; run one_stream
export UNET_AMD_WGRAD_STREAM=1; run two_streams
done
