#!/bin/bash
# usage: tools/pmc.sh <tag> "<COUNTERS pass 1>" "<COUNTERS pass 2>" ...   (env knobs pass through)
# One rocprofv3 --pmc pass per argument (no trace options: gpurun refuses the combination).
tag=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs -d $repo/gpurun_out/pmc_${tag}_$i -o out -- python $repo/tools/one_reduce.py > /dev/null 2>&1
  db=$(ls $repo/gpurun_out/pmc_${tag}_$i/*.db $repo/gpurun_out/pmc_${tag}_$i/*/*.db 2>/dev/null | head -1)
  python $repo/profiles/rocprof_summary.py $db | grep -i "reduce_fused" | sed "s/^/[$tag] /"
done
