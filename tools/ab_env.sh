#!/bin/bash
# ab_env.sh VAR valA valB [rounds]: bench.py under VAR=valA / VAR=valB alternately in one call
var=$1; a=$2; b=$3; n=${4:-2}
out=gpurun_out/ab_env; mkdir -p $out
for i in $(seq 1 $n); do
  for v in $a $b; do
    env $var=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-workload > $out/r.json 2> $out/r.err || exit 1
    python -c "import json; d=json.load(open('$out/r.json')); print('$var=$v', d['ms_per_step'], {k: (v['TFLOP/s'], v['ms_per_step']) for k, v in d['kernels'].items()})"
  done
done
