# round 4, GPU call F: the streaming ushort convolution
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_conv_u16_gpu.py tests/test_conv_colour_gpu.py -m gpu -q -k "u16 or ushort or conv" 2>&1 | tail -8 ) > gpurun_out/r04g_tests.txt 2>&1
tail -4 gpurun_out/r04g_tests.txt
for seg in default 16 48; do
  if [ $seg = default ]; then unset VIPS_HIP_CONV_U16_SEG; else export VIPS_HIP_CONV_U16_SEG=$seg; fi
  echo "== seg $seg"
  timeout 300 python bench.py --config ops --ops u16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in d['ops']: print('  %-28s %.4f ms  frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], e['kernels'], e.get('parity')))"
done > gpurun_out/r04g_ops.txt 2>&1
cat gpurun_out/r04g_ops.txt
