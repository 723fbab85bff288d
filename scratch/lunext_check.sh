#!/bin/bash
set -e
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/lunext; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_lunext_gpu.py tests/test_kernels_gpu.py tests/test_unet_gpu.py -x -q -s --timeout 180 --timeout-method thread > $O/test.log 2>&1 || { tail -40 $O/test.log; exit 1; }; grep -E "passed|failed|cosine|worst" $O/test.log | cut -c1-200

timeout -k 10 300 python bench.py --workload lunext --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o lunext --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload lunext --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> $O/stats.err
head -12 $O/stats/lunext_kernel_stats.csv | cut -c1-60,200-400
python3 - <<'PY'
import csv, collections, os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/lunext/stats/lunext_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if n.startswith('dwconv') or 'dwconv' in n[:30]:
        d[(n.split('(')[0][:28], r.get('Grid_Size_X') or r.get('Grid_Size'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()):
    print(k, len(v), f"{sum(v)/len(v):.1f} us")
PY
