mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_resample16_gpu.py -x -q > gpurun_out/r04_r16_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04_r16_tests.txt; tail -4 gpurun_out/r04_r16_tests.txt
timeout 900 python -m pytest tests/test_resample_gpu.py -x -q -k "format or ushort or u16 or matrix" > gpurun_out/r04_r16_tests_b.txt 2>&1; echo "rc=$?" >> gpurun_out/r04_r16_tests_b.txt; tail -3 gpurun_out/r04_r16_tests_b.txt
for seg in default 16 64; do
  if [ $seg = default ]; then unset VIPS_HIP_R16_SEG; else export VIPS_HIP_R16_SEG=$seg; fi
  echo "== seg $seg"
  timeout 300 python bench.py --config ops --ops rgba16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in d['ops']: print('  %-20s %.4f ms  frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], e['kernels'], e.get('parity')))"
done
