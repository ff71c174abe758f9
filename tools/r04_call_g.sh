# round 4, GPU call G: conv_u8_sep with the rings in registers against LDS; conv_u16 after the cached check
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_conv_u8_gpu.py tests/test_conv_u16_gpu.py -m gpu -q 2>&1 | tail -6 ) > gpurun_out/r04h_tests.txt 2>&1
tail -3 gpurun_out/r04h_tests.txt
for ring in regs lds; do
  export VIPS_HIP_CONV_U8_RING=$ring
  echo "== ring $ring"
  timeout 300 python bench.py --config ops --ops gaussblur_s,convi_3x3_u16,convi_5x5_u16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in d['ops']: print('  %-28s %.4f ms  frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], e['kernels'], e.get('parity')))"
done > gpurun_out/r04h_ops.txt 2>&1
cat gpurun_out/r04h_ops.txt
