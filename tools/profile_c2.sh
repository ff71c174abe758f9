#!/bin/bash
# usage: tools/profile_c2.sh <tag>   -> gpurun_out/<tag>_{bench.json,rocprofv3.txt}
# kernel trace + separate FETCH_SIZE / WRITE_SIZE PMC passes of the default bench command
tag=$1
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p $out
python bench.py --steps 50 --warmup 5 --no-configs > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
cmd="python $repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs"
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $cmd > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -o f -- $cmd > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_w -o w -- $cmd > /dev/null 2>&1
{
  echo "# rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $cmd"
  python $repo/profiles/rocprof_summary.py $(find /tmp/prof_kt -name '*.db' | head -1) $(find /tmp/prof_f -name '*.db' | head -1) $(find /tmp/prof_w -name '*.db' | head -1)
} > $out/${tag}_rocprofv3.txt 2>&1
