mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_u8_gpu.py -x -q > gpurun_out/r04_conv_u8_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04_conv_u8_tests.txt; tail -3 gpurun_out/r04_conv_u8_tests.txt
for seg in default 6; do
  if [ $seg = default ]; then unset VIPS_HIP_CONV_U8_SEG; else export VIPS_HIP_CONV_U8_SEG=$seg; fi
  echo "== seg $seg"
  timeout 300 python bench.py --config ops --ops convi_3x3_u8,convi_5x5_u8,gaussblur_s2_u8,gaussblur_s8_u8 --steps 10 --warmup 3 --no-verify --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in d['ops']: print('  %-20s %.4f ms  frac %.3f' % (e['name'], e['ms'], e['frac']))"
done
