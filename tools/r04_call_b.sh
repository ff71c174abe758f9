# round 4, GPU call B (second go): what call A's -x cut off (module output side, shrinkh_u8, the module), the
# packed reduceh, the unrolled issue-rate probe, and the reduce / shrink entries of the ops table
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_module_stream.py tests/test_shrinkh_u8_gpu.py tests/test_reduceh_u8_gpu.py tests/test_module.py -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r04c_tests1.txt 2>&1
tail -4 gpurun_out/r04c_tests1.txt
( timeout 900 python -m pytest tests/test_resample_gpu.py -m gpu -q -x -k "reduce or shrink or golden or format" 2>&1 | tail -6 ) > gpurun_out/r04c_tests2.txt 2>&1
tail -3 gpurun_out/r04c_tests2.txt
( timeout 120 tools/valu_probe2 ) > gpurun_out/r04c_valu_probe2.txt 2>&1
cat gpurun_out/r04c_valu_probe2.txt
timeout 400 python bench.py --config ops --ops reduce,shrink --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in d['ops']: print('  %-28s %.4f ms  frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], e['kernels'], e.get('parity')))" > gpurun_out/r04c_ops.txt 2>&1
cat gpurun_out/r04c_ops.txt
