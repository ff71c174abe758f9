#!/bin/bash
# Round 5, call E: the colour kernels with the cube-root table in cbrt_quad.h's form -- parity file, ops entries
tag=${1:-r05l}
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_conv_colour_gpu.py -q -m gpu -x -k "colour or lab or Lab or srgb or sharpen" 2>&1 | tail -5) > gpurun_out/${tag}_tests.txt
tail -3 gpurun_out/${tag}_tests.txt
ops="python bench.py --config ops --ops colourspace_srgb,sharpen --steps 10 --warmup 3 --no-cpu-baseline"
show() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in l['ops']: print('  %-28s %.4f ms frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], e['kernel'], (e.get('parity') or {}).get('bit_exact')))
"; }
{
for env in ${VARIANTS:-"A=1" "VIPS_HIP_NO_CBRT_QUAD=1"}; do
  echo "# $env $ops"
  env $env $ops 2>/dev/null | show
done
} > gpurun_out/${tag}_ops.txt 2>&1
cat gpurun_out/${tag}_ops.txt
