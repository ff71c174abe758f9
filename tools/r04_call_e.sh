# round 4, GPU call E: the rest of the GPU suite (the closing run's -x stopped at test_module.py),
# the streaming uchar reducev, the module end to end after the condition-variable split
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_module.py tests/test_module_stream.py tests/test_multidevice_gpu.py tests/test_reduceh_u8_gpu.py tests/test_reducev8_gpu.py tests/test_resample16_gpu.py tests/test_resample_gpu.py tests/test_sharding.py tests/test_shrinkh_u8_gpu.py tests/test_threads_gpu.py tests/test_vfile.py tests/test_x80.py tests/test_zz_jpeg.py tests/test_zz_vector.py -m gpu -q 2>&1 | tail -12 ) > gpurun_out/r04e_tests.txt 2>&1
tail -5 gpurun_out/r04e_tests.txt
( timeout 300 python tools/module_e2e.py 2>/dev/null | grep -E "ms_module|ms_builtin|bit_exact|h2d_ms_pageable" ) > gpurun_out/r04e_module_e2e.txt 2>&1
cat gpurun_out/r04e_module_e2e.txt
timeout 400 python bench.py --config ops --ops reduce --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in d['ops']: print('  %-28s %.4f ms  frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], e['kernels'], e.get('parity')))" > gpurun_out/r04e_ops.txt 2>&1
cat gpurun_out/r04e_ops.txt
