#!/bin/bash
# ab_lunext.sh VAR rounds val1 val2 ...: the LUNeXt + clDice workload of bench.py under VAR=val_i in turn, in one call
var=$1; n=$2; shift 2
out=gpurun_out/ab_lunext; mkdir -p $out
for i in $(seq 1 $n); do
  for v in "$@"; do
    env $var=$v timeout -k 10 200 python bench.py --workload lunext --steps 20 --warmup 5 --no-cpu-baseline > $out/r.json 2> $out/r.err || exit 1
    python -c "import json; d=json.load(open('$out/r.json')); print('$var=$v', d['ms_per_step'], d['value'], d['roofline'].get('ms_per_call'))"
  done
done
