#!/bin/bash
# prof_dw.sh <tag>: L1->L2 and L2->memory read counters of the depthwise-block backward kernels in tools/bench_dw.py
# (one counter pass per run) -> gpurun_out/prof_dw_<tag>/
tag=$1; out=$GRAFT_REPO_ROOT/gpurun_out/prof_dw_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for pass in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $out/pmc_$name -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_dw.py > $out/pmc_$name.log 2> $out/pmc_$name.err || exit 1
  f=$(find $out/pmc_$name -name "*counter_collection.csv" | head -1); cp $f $out/pmc_$name.csv; rm -rf $out/pmc_$name
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, collections, glob
for path in sorted(glob.glob("$out/pmc_*.csv")):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "dwblock" in k or "bn_res_relu_bwd_apply" in k:
            d[(k, r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in sorted(d.items()):
        print(k, f"n={len(v)} mean={sum(v) / len(v):.4g}")
PY
