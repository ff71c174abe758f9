#!/bin/bash
# Round 5, call B: the matrix-core separable convolution (conv_u8_mfma.hip) -- parity file, then the
# gaussblur entries of the ops table with the kernel on / off and for blocks per CU / segment lengths.
tag=${1:-r05b}
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_conv_u8_gpu.py -q -m gpu 2>&1 | tail -8) > gpurun_out/${tag}_tests.txt
tail -4 gpurun_out/${tag}_tests.txt
ops="python bench.py --config ops --ops gaussblur_s8_u8,gaussblur_s2_u8 --steps 20 --warmup 3 --no-cpu-baseline"
show() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in l['ops']: print('  %-22s %.4f ms frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], e['kernel'], (e.get('parity') or {}).get('bit_exact')))
"; }
{
for env in ${VARIANTS:-"VIPS_HIP_CONV_U8_MFMA=1" "VIPS_HIP_CONV_MFMA_PER_CU=4" "VIPS_HIP_CONV_MFMA_PER_CU=6" "VIPS_HIP_CONV_MFMA_SEG=3" "VIPS_HIP_CONV_MFMA_SEG=4" "VIPS_HIP_CONV_MFMA_SEG=6" "VIPS_HIP_CONV_MFMA_SEG=8" "VIPS_HIP_CONV_MFMA_SEG=12"}; do
  echo "# $env $ops"
  env $env $ops 2>/dev/null | show
done
} > gpurun_out/${tag}_ops.txt 2>&1
cat gpurun_out/${tag}_ops.txt
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
one="python $repo/bench.py --config ops --ops gaussblur_s8_u8,gaussblur_s2_u8 --steps 5 --warmup 2 --no-cpu-baseline --no-verify"
i=0
{
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs -d /tmp/pb_$i -o out -- $one > /dev/null 2>&1
  echo "# rocprofv3 --pmc $ctrs -- $one"
  python $repo/profiles/rocprof_summary.py $(find /tmp/pb_$i -name '*.db' | head -1) | grep -i "conv_u8\|counter"
done
rocprofv3 --kernel-trace --stats -d /tmp/pb_kt -o kt -- $one > /dev/null 2>&1
echo "# rocprofv3 --kernel-trace --stats -- $one"
python $repo/profiles/rocprof_summary.py $(find /tmp/pb_kt -name '*.db' | head -1) | grep -v "^$" | head -6
} > $repo/gpurun_out/${tag}_pmc.txt 2>&1
cut -c1-160 $repo/gpurun_out/${tag}_pmc.txt
