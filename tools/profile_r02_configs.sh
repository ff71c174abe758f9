#!/bin/bash
# usage: tools/profile_r02_configs.sh <tag> -> gpurun_out/<tag>_pmc.txt
# PMC passes (no trace options) over the C3 and C5-slab benches: HBM traffic against the
# algorithmic bytes, and where the waves' time goes.
tag=$1
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
c3="python $repo/bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-verify"
c5="python $repo/bench.py --config c5slab --steps 2 --warmup 1 --no-cpu-baseline --no-verify"
i=0
{
for cmd in "$c3" "$c5"; do
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $ctrs -d /tmp/pmc_${tag}_$i -o out -- $cmd > /dev/null 2>&1
    echo "# rocprofv3 --pmc $ctrs -- $cmd"
    python $repo/profiles/rocprof_summary.py $(find /tmp/pmc_${tag}_$i -name '*.db' | head -1) | grep -i "convsep_stream\|convf_rows\|counter"
  done
done
} > $out/${tag}_pmc.txt 2>&1
