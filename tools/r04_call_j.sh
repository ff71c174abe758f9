# round 4, GPU call J: XYZ2Lab's cube-root table through the single-precision form (colour_lab_lds_kernel)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_conv_colour_gpu.py -m gpu -q -k "colour or lab or Lab or route" 2>&1 | tail -4 ) > gpurun_out/r04k_tests.txt 2>&1
tail -2 gpurun_out/r04k_tests.txt
for form in f32 f64; do
  if [ $form = f64 ]; then export VIPS_HIP_CBRT_F64=1; else unset VIPS_HIP_CBRT_F64; fi
  echo "== cbrt form $form"
  timeout 200 python bench.py --config ops --ops colourspace_srgb --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in d['ops']: print('  %-28s %.4f ms  frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], e['kernels'], e.get('parity')))"
done > gpurun_out/r04k_ops.txt 2>&1
cat gpurun_out/r04k_ops.txt
