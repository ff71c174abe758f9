# round 4, GPU call A: the new module output side, shrinkh_u8, the float quotient of XYZ2Lab, the C3
# phase order, and the issue-rate probe -- tests first, then timings
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_module_stream.py tests/test_shrinkh_u8_gpu.py tests/test_module.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r04a_tests1.txt 2>&1
tail -3 gpurun_out/r04a_tests1.txt
( timeout 900 python -m pytest tests/test_conv_colour_gpu.py tests/test_full_size_gpu.py -m gpu -x -q -k "colour or c3 or C3 or sharpen or lab" 2>&1 | tail -6 ) > gpurun_out/r04a_tests2.txt 2>&1
tail -3 gpurun_out/r04a_tests2.txt
( timeout 120 tools/valu_probe2 ) > gpurun_out/r04a_valu_probe2.txt 2>&1
cat gpurun_out/r04a_valu_probe2.txt
for order in 0 1; do
  echo "== VIPS_HIP_STREAM_ORDER=$order"
  VIPS_HIP_STREAM_ORDER=$order timeout 300 python bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  c3 ms_per_step', d.get('ms_per_step'), d.get('roofline'))"
done > gpurun_out/r04a_c3_order.txt 2>&1
cat gpurun_out/r04a_c3_order.txt
timeout 400 python bench.py --config ops --ops shrink,colourspace,sharpen --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in d['ops']: print('  %-28s %.4f ms  frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], e['kernels'], e.get('parity')))" > gpurun_out/r04a_ops.txt 2>&1
cat gpurun_out/r04a_ops.txt
