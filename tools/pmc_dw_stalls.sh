#!/bin/bash
# pmc_dw_stalls.sh <tag> [script] [kernel substring]: where the waves of the depthwise-block backward (default:
# tools/bench_dw.py, dwblock_bwd_strip) spend their cycles, from two counter passes of eight and seven SQ counters
# -> gpurun_out/pmc_dw_stalls_<tag>/
tag=$1; script=${2:-tools/bench_dw.py}; kern=${3:-dwblock_bwd_strip}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_dw_stalls_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
            "SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $out/p$i -o p -- python3 $GRAFT_REPO_ROOT/$script > $out/p$i.log 2> $out/p$i.err || exit 1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1); cp $f $out/p$i.csv; rm -rf $out/p$i
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, collections, glob
d = collections.defaultdict(list)
for path in sorted(glob.glob("$out/p*.csv")):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "$kern" in k:
            d[(k[-24:] + " " + r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
grids = sorted({g for g, _ in d})
names = sorted({n for _, n in d})
for g in grids:
    print("grid", g, " ".join(f"{n}={sum(d[(g, n)]) / len(d[(g, n)]):.4g}" for n in names if (g, n) in d))
PY
