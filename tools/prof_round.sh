#!/bin/bash
# prof_round.sh <tag>: bench line + rocprofv3 kernel stats of the same command -> gpurun_out/prof_<tag>/
tag=$1; out=gpurun_out/prof_$tag; mkdir -p $out
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench_unet.json 2> $out/bench_unet.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp -o unet -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-second-workload > $GRAFT_REPO_ROOT/$out/bench_unet_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/rocprof.err || exit 1
cd $GRAFT_REPO_ROOT
f=$(find $out/rp -name "*kernel_stats.csv" | head -1); cp $f $out/unet_kernel_stats.csv
find $out/rp -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $out/rp -name "*.db" -delete
head -30 $out/unet_kernel_stats.csv | cut -c1-150
python -c "import json; d=json.load(open('$out/bench_unet.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernels'])"
# ---- PMC passes (each in a run of its own): L2 memory-side traffic and MFMA busy cycles per kernel
cd /tmp
for pass in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 400 rocprofv3 --pmc $pass --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_$name -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-second-workload > $GRAFT_REPO_ROOT/$out/pmc_$name.json 2> $GRAFT_REPO_ROOT/$out/pmc_$name.err || exit 1
  f=$(find $GRAFT_REPO_ROOT/$out/pmc_$name -name "*counter_collection.csv" | head -1)
  cp $f $GRAFT_REPO_ROOT/$out/pmc_$name.csv; rm -rf $GRAFT_REPO_ROOT/$out/pmc_$name
done
cd $GRAFT_REPO_ROOT
python tools/traffic_json.py $out/pmc_FETCH_SIZE.csv $out/pmc_WRITE_SIZE.csv $out/traffic.json
python tools/pmc_summ.py $out/pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv conv_ > $out/pmc_mfma_summary.txt; cat $out/pmc_mfma_summary.txt | head -40
