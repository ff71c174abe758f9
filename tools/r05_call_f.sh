#!/bin/bash
# Round 5, call F: the all-in-LDS sharpen (sharpen_quad_u8) -- parity, the ops entry, C4 with CU splits
tag=${1:-r05p}
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_conv_colour_gpu.py -q -m gpu -k "sharpen" 2>&1 | tail -3) > gpurun_out/${tag}_tests.txt
tail -2 gpurun_out/${tag}_tests.txt
show() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in l.get('ops', []): print('  %-28s %.4f ms frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], e['kernel'], (e.get('parity') or {}).get('bit_exact')))
if 'config' in l and 'ops' not in l: print('  c4 %.3f ms/step frac %.4f per image %s kernels %s parity %s' % (l['ms_per_step'], l['roofline']['frac'], l['config'].get('ms_per_image'), l['roofline'].get('kernels'), (l.get('parity') or {}).get('bit_exact')))
"; }
{
for env in ${VARIANTS:-"A=1" "VIPS_HIP_NO_SHARPEN_QUAD=1" "VIPS_HIP_SHARPEN_QUAD_GRID=512"}; do
  echo "# $env ops sharpen_u8"
  env $env python bench.py --config ops --ops sharpen_u8 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | show
done
for env in ${C4VARIANTS:-"A=1" "VIPS_HIP_NO_SHARPEN_QUAD=1"}; do
  echo "# $env c4 256 images"
  env $env python bench.py --config c4 --images 256 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | show
done
} > gpurun_out/${tag}_ops.txt 2>&1
cat gpurun_out/${tag}_ops.txt
