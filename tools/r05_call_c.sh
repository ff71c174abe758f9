#!/bin/bash
# Round 5, call C: the banded-matrix vertical reduce (reduce_band.hip) -- parity files, the reduce entries of the
# ops table with the kernel on / off, counters.
tag=${1:-r05h}
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_reduce_band_gpu.py tests/test_reducev8_gpu.py tests/test_reduceh_u8_gpu.py -q -m gpu 2>&1 | tail -8) > gpurun_out/${tag}_tests.txt
tail -4 gpurun_out/${tag}_tests.txt
ops="python bench.py --config ops --ops reduce_rgb_7.3,reducev_8 --steps 20 --warmup 3 --no-cpu-baseline"
show() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in l['ops']: print('  %-22s %.4f ms frac %.3f %s %s %s' % (e['name'], e['ms'], e['frac'], e['kernel'], (e.get('parity') or {}).get('bit_exact'), {k: v['mean_ms'] for k, v in e['kernels'].items()}))
"; }
{
for env in ${VARIANTS:-"VIPS_HIP_REDUCE_BAND=1" "VIPS_HIP_REDUCE_BAND=0"}; do
  echo "# $env $ops"
  env $env $ops 2>/dev/null | show
done
} > gpurun_out/${tag}_ops.txt 2>&1
cat gpurun_out/${tag}_ops.txt
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
one="python $repo/bench.py --config ops --ops reduce_rgb_7.3 --steps 5 --warmup 2 --no-cpu-baseline --no-verify"
i=0
{
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs -d /tmp/pc_$i -o out -- $one > /dev/null 2>&1
  echo "# rocprofv3 --pmc $ctrs -- $one"
  python $repo/profiles/rocprof_summary.py $(find /tmp/pc_$i -name '*.db' | head -1) | grep -i "reduce\|counter"
done
rocprofv3 --kernel-trace --stats -d /tmp/pc_kt -o kt -- $one > /dev/null 2>&1
echo "# rocprofv3 --kernel-trace --stats -- $one"
python $repo/profiles/rocprof_summary.py $(find /tmp/pc_kt -name '*.db' | head -1) | grep -v "^$" | head -6
} > $repo/gpurun_out/${tag}_pmc.txt 2>&1
cut -c1-170 $repo/gpurun_out/${tag}_pmc.txt
