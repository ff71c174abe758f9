#!/bin/bash
# usage: tools/profile_c4.sh <tag> -> gpurun_out/<tag>_c4_pmc.txt
# HBM traffic of the C4 kernels against their algorithmic bytes: separate rocprofv3 --pmc passes
# (no trace options) over `bench.py --config c4` with 64 images.
tag=$1
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cmd="python $repo/bench.py --config c4 --images 64 --steps 1 --warmup 1 --no-cpu-baseline --no-verify"
i=0
{
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs -d /tmp/pmc_${tag}_$i -o out -- $cmd > /dev/null 2>&1
  echo "# rocprofv3 --pmc $ctrs -- $cmd"
  python $repo/profiles/rocprof_summary.py $(find /tmp/pmc_${tag}_$i -name '*.db' | head -1) | grep -i "resize_stream\|sharpen_fused\|counter"
done
} > $out/${tag}_c4_pmc.txt 2>&1
