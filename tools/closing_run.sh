#!/bin/bash
# A round's closing run on one box (usage: gpurun -- 'bash tools/closing_run.sh <tag>'; outputs under
# gpurun_out/<tag>_*, the ones worth keeping are copied into profiles/):
#   1. the whole GPU suite (no -x)
#   2. C2: the two PMC passes that stamp profiles/traffic.json (by the kernel's machine-code hash) + a kernel
#      trace of the driver's command
#   3. C3: counters + kernel trace of the shipped build, on both inputs (integers / float proper)
#   4. the full default bench (roofline.traffic non-null, C3/C4/C5 scalars inside roofline, summary last)
tag=${1:-closing}
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -14) > gpurun_out/${tag}_tests.txt
tail -3 gpurun_out/${tag}_tests.txt
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
sum="python $repo/profiles/rocprof_summary.py"
db() { find $1 -name '*.db' | head -1; }
for c in FETCH_SIZE WRITE_SIZE; do
  TUNE_LAUNCHES=8 rocprofv3 --pmc $c -d /tmp/pa_$c -o out -- python $repo/tools/one_reduce.py > /dev/null 2>&1
  echo "# TUNE_LAUNCHES=8 rocprofv3 --pmc $c -- python tools/one_reduce.py"
  $sum $(db /tmp/pa_$c) | grep -i "reduce_fused\|counter"
done > $repo/gpurun_out/${tag}_c2_pmc.txt 2>&1
drv="python $repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-verify"
rocprofv3 --kernel-trace --stats -d /tmp/pa_kt -o kt -- $drv > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $drv"; $sum $(db /tmp/pa_kt) | grep -v "^$" | head -8; } > $repo/gpurun_out/${tag}_c2_rocprofv3.txt 2>&1
c3="python $repo/bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-verify"
i=0
{
for input in int float; do
  for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    VIPS_BENCH_C3_INPUT=$input rocprofv3 --pmc $ctrs -d /tmp/pa_c3_$i -o out -- $c3 > /dev/null 2>&1
    echo "# VIPS_BENCH_C3_INPUT=$input rocprofv3 --pmc $ctrs -- $c3"
    $sum $(db /tmp/pa_c3_$i) | grep -i "convsep_stream\|counter"
  done
done
} > $repo/gpurun_out/${tag}_c3_pmc.txt 2>&1
{
for input in int float; do
  VIPS_BENCH_C3_INPUT=$input timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pa_c3kt_$input -o kt -- $c3 > /dev/null 2>&1
  echo "# VIPS_BENCH_C3_INPUT=$input rocprofv3 --kernel-trace --stats -- $c3"
  $sum $(db /tmp/pa_c3kt_$input) | grep -v "^$" | head -12
done
} > $repo/gpurun_out/${tag}_c3_rocprofv3.txt 2>&1
ops="python $repo/bench.py --config ops --steps 10 --warmup 3 --no-cpu-baseline --no-verify"
rocprofv3 --kernel-trace --stats -d /tmp/pa_ops -o kt -- $ops > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $ops"; $sum $(db /tmp/pa_ops) | grep -v "^$" | head -70; } > $repo/gpurun_out/${tag}_ops_rocprofv3.txt 2>&1
cd $repo
cut -c1-170 gpurun_out/${tag}_c2_pmc.txt
cut -c1-170 gpurun_out/${tag}_c3_pmc.txt
(timeout 100 python tools/fuzz_gpu.py 40 91 resize; timeout 100 python tools/fuzz_gpu.py 30 92 thumb) 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_fuzz.txt; cat gpurun_out/${tag}_fuzz.txt
(timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 | tail -1) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo rc=$?; cp gpurun_out/bench_full.json gpurun_out/${tag}_bench_full.json
python - <<PY
import json
l=json.load(open("gpurun_out/${tag}_bench.json"))
print({k:l[k] for k in ("ms_per_step","value")}, l["roofline"]["frac"], l["roofline"]["traffic"])
print(json.dumps(l["roofline"]))
PY
