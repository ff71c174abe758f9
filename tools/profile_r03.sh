#!/bin/bash
# usage: tools/profile_r03.sh <tag>  ->  gpurun_out/<tag>_{c2_rocprofv3,c2_pmc,c3_pmc,c4_pmc,c5_pmc}.txt
# Round 3: a kernel trace of the driver's own C2 command, then separate rocprofv3 --pmc passes (no
# trace options: gpurun refuses the combination) -- FETCH_SIZE / WRITE_SIZE of every config's
# dominant kernel against its algorithmic bytes, and where the waves' time goes.
tag=$1
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
sum="python $repo/profiles/rocprof_summary.py"
db() { find $1 -name '*.db' | head -1; }

c2="python $repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-verify"
rocprofv3 --kernel-trace --stats -d /tmp/p3_kt -o kt -- $c2 > /dev/null 2>&1
{
  echo "# rocprofv3 --kernel-trace --stats -- $c2"
  $sum $(db /tmp/p3_kt) | grep -v "^$" | head -14
} > $out/${tag}_c2_rocprofv3.txt 2>&1

one="python $repo/tools/one_reduce.py"
i=0
{
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  TUNE_LAUNCHES=8 rocprofv3 --pmc $ctrs -d /tmp/p3_c2_$i -o out -- $one > /dev/null 2>&1
  echo "# TUNE_LAUNCHES=8 rocprofv3 --pmc $ctrs -- $one"
  $sum $(db /tmp/p3_c2_$i) | grep -i "reduce_fused\|counter"
done
} > $out/${tag}_c2_pmc.txt 2>&1

for cfg in c3 c4 c5slab; do
  case $cfg in
    c3) cmd="python $repo/bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-verify"; pat="convsep_stream";;
    c4) cmd="python $repo/bench.py --config c4 --images 256 --steps 1 --warmup 1 --no-cpu-baseline --no-verify"; pat="resize_stream\|sharpen_fused";;
    c5slab) cmd="python $repo/bench.py --config c5slab --steps 2 --warmup 1 --no-cpu-baseline --no-verify"; pat="convf_rows";;
  esac
  {
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" ; do
    i=$((i+1))
    rocprofv3 --pmc $ctrs -d /tmp/p3_${cfg}_$i -o out -- $cmd > /dev/null 2>&1
    echo "# rocprofv3 --pmc $ctrs -- $cmd"
    $sum $(db /tmp/p3_${cfg}_$i) | grep -i "$pat\|counter"
  done
  } > $out/${tag}_${cfg}_pmc.txt 2>&1
done
