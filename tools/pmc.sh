#!/bin/bash
# usage: tools/pmc.sh <tag> <script.py> <kernel-name-grep> "<COUNTERS pass 1>" "<COUNTERS pass 2>" ...
# One rocprofv3 --pmc pass per counter list (no trace options: gpurun refuses the combination).
tag=$1; script=$2; pat=$3; shift 3
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs -d /tmp/pmc_${tag}_$i -o out -- python $repo/$script > /dev/null 2>&1
  db=$(find /tmp/pmc_${tag}_$i -name '*.db' | head -1)
  python $repo/profiles/rocprof_summary.py $db | grep -i "$pat" | sed "s/^/[$tag] /"
done
