# round 4, GPU call D: the module ring tests with memory inputs, conv_u8 with B-dword loads
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_module_stream.py tests/test_conv_u8_gpu.py -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r04d_tests1.txt 2>&1
tail -4 gpurun_out/r04d_tests1.txt
timeout 400 python bench.py --config ops --ops convi,gaussblur --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in d['ops']: print('  %-28s %.4f ms  frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], e['kernels'], e.get('parity')))" > gpurun_out/r04d_ops.txt 2>&1
cat gpurun_out/r04d_ops.txt
