#!/bin/bash
# Round 5, first GPU call (nothing of round 4's last changes -- the integer horizontal pass and the
# staggered epilogue of convsep_stream -- has been through the whole suite or a counter pass: the round
# ran out of GPU minutes with them validated on their own files and on C3's whole-size parity rows).
#   1. the whole GPU suite (no -x)
#   2. the full default bench
#   3. C3 under the counters, both forms of the epilogue and the integer pass off: SQ_WAIT_ANY /
#      SQ_WAVE_CYCLES should fall with the staggered form if NOTES R4.10's reading is right
#   4. the kernel trace of C3 (average duration against the bench's events)
tag=${1:-r05a}
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -14) > gpurun_out/${tag}_tests.txt
tail -3 gpurun_out/${tag}_tests.txt
(timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo rc=$?
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
c3="python $repo/bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-verify"
i=0
{
for env in "VIPS_HIP_STREAM_EPI=1" "VIPS_HIP_STREAM_EPI=0" "VIPS_HIP_STREAM_INT=0"; do
  for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    env $env rocprofv3 --pmc $ctrs -d /tmp/pmc_${tag}_$i -o out -- $c3 > /dev/null 2>&1
    echo "# $env rocprofv3 --pmc $ctrs -- $c3"
    python $repo/profiles/rocprof_summary.py $(find /tmp/pmc_${tag}_$i -name '*.db' | head -1) | grep -i "convsep_stream\|counter"
  done
done
} > $repo/gpurun_out/${tag}_c3_pmc.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_${tag} -o kt -- $c3 > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $c3"; python $repo/profiles/rocprof_summary.py $(find /tmp/kt_${tag} -name '*.db' | head -1) | grep -v "^$" | head -20; } > $repo/gpurun_out/${tag}_c3_rocprofv3.txt 2>&1
