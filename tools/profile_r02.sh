#!/bin/bash
# usage: tools/profile_r02.sh <tag>   -> gpurun_out/<tag>_{bench.json,rocprofv3.txt}
# The default bench line (C2 + configs), a kernel trace of the same command, and separate
# FETCH_SIZE / WRITE_SIZE PMC passes of the C2 part (no trace options with --pmc).
tag=$1
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p $out
python bench.py --steps 50 --warmup 5 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
full="python $repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify"
c2="python $repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs"
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $full > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -o f -- $c2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_w -o w -- $c2 > /dev/null 2>&1
VIPS_HIP_FUSED_DEBUG=4 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fp -o fp -- $c2 > /dev/null 2>&1
{
  echo "# rocprofv3 --kernel-trace --stats -- $full"
  echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $c2"
  echo "# last table: FETCH_SIZE of the same command with VIPS_HIP_FUSED_DEBUG=4 (nt instead of plain loads)"
  python $repo/profiles/rocprof_summary.py $(find /tmp/prof_kt -name '*.db' | head -1) $(find /tmp/prof_f -name '*.db' | head -1) $(find /tmp/prof_w -name '*.db' | head -1) $(find /tmp/prof_fp -name '*.db' | head -1)
} > $out/${tag}_rocprofv3.txt 2>&1
