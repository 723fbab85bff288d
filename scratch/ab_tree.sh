#!/bin/bash
# this tree against scratch/base_tree (an export of the last commit, built): $1 = bench arguments
set -e
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
(cd scratch/base_tree && timeout -k 10 200 python bench.py $1 --no-cpu-baseline --steps 30 2>/dev/null) | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('base',d['value'],d['ms_per_step'])"
timeout -k 10 200 python bench.py $1 --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('new ',d['value'],d['ms_per_step'])"
done
