#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ky in 0 1; do
  export UH_KY_INNER=$ky
  echo "== ky_inner=$ky"
  timeout -k 10 100 python $R/scratch/bench_conv.py all fwd 2>&1 | grep -v amdgpu.ids || exit 1
  for shp in conv2_2 conv_up2_1 conv1_2 conv4_2; do
    timeout -s KILL 100 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/ky$ky -o $shp --output-format csv -- python $R/scratch/bench_conv.py $shp fwd 5 > /dev/null 2>&1 || exit 1
    python $R/scratch/pmc_summ.py $R/gpurun_out/ky$ky/${shp}_counter_collection.csv conv_band | sed "s/^/$shp /"
  done
done
