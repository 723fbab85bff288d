#!/bin/bash
out=gpurun_out/ab_prio; mkdir -p $out
run() { # tag prio hwq
  UNET_AMD_SIDE_PRIORITY=$2 GPU_MAX_HW_QUEUES=$3 timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-workload > $out/$1.json 2> $out/$1.err || return 1
  python -c "import json,sys; d=json.load(open('$out/$1.json')); print('$1', d['ms_per_step'], d['value'])"
}
run s_prio0_q4 0 4 && run s_priohi_q4 -1 4 && run s_prio0_q8 0 8 && run s_priohi_q8 -1 8 && run s_prio0_q4_b 0 4 && run s_priohi_q4_b -1 4
