#!/bin/bash
# training step (batch 32, 256^2) with a one-rank RCCL group: second stream priority x hardware queue count
out=gpurun_out/ab_prio; mkdir -p $out
export UNET_AMD_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1
run() { # tag prio hwq8
  UNET_AMD_SIDE_PRIORITY=$2 UNET_AMD_BENCH_HWQ8=$3 timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-second-workload > $out/$1.json 2> $out/$1.err || return 1
  python -c "import json,sys; d=json.load(open('$out/$1.json')); print('$1', d['ms_per_step'], d['value'])"
}
run prio0_q4 0 0 && run priohi_q8 -1 1 && run prio0_q8 0 1 && run priohi_q4 -1 0 && run prio0_q4_b 0 0 && run priohi_q8_b -1 1
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-workload > $out/single.json 2> $out/single.err && python -c "import json; d=json.load(open('$out/single.json')); print('single prio0', d['ms_per_step'])"
UNET_AMD_SIDE_PRIORITY=-1 timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-workload > $out/single_hi.json 2> $out/single_hi.err && python -c "import json; d=json.load(open('$out/single_hi.json')); print('single priohi', d['ms_per_step'])"
