#!/bin/bash
# ab_vals.sh VAR rounds val1 val2 ...: bench.py under VAR=val_i in turn, `rounds` times, in one call ("-" = empty value)
var=$1; n=$2; shift 2
out=gpurun_out/ab_env; mkdir -p $out
for i in $(seq 1 $n); do
  for v in "$@"; do
    val=$v; [ "$v" = "-" ] && val=""
    env $var=$val timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-workload > $out/r.json 2> $out/r.err || exit 1
    python -c "import json; d=json.load(open('$out/r.json')); print('$var=$v', d['ms_per_step'], {k: (v['TFLOP/s'], v['ms_per_step']) for k, v in d['kernels'].items()})"
  done
done
