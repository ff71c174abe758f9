#!/bin/bash
# usage: tools/profile_c3.sh <tag> -> gpurun_out/<tag>_c3_pmc.txt
# PMC passes (no trace options) over the C3 bench: HBM traffic against the algorithmic bytes and
# the VALU instruction count of convsep_stream (SQ_INSTS_VALU / (32768^2 * 3 / 64) = instructions
# per element).
tag=$1
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
c3="python $repo/bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-verify"
i=0
{
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs -d /tmp/pmc_${tag}_$i -o out -- $c3 > /dev/null 2>&1
  echo "# rocprofv3 --pmc $ctrs -- $c3"
  python $repo/profiles/rocprof_summary.py $(find /tmp/pmc_${tag}_$i -name '*.db' | head -1) | grep -i "convsep_stream\|counter"
done
} > $out/${tag}_c3_pmc.txt 2>&1
