# round 4, GPU call I: the streaming uchar reducev with its new segment rule (tests + the op), C2 alone on this box
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_reducev8_gpu.py tests/test_resample16_gpu.py -m gpu -q 2>&1 | tail -4 ) > gpurun_out/r04j_tests.txt 2>&1
tail -2 gpurun_out/r04j_tests.txt
timeout 200 python bench.py --config ops --ops reduce_rgb,reduce_rgba16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in d['ops']: print('  %-28s %.4f ms  frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], e['kernels'], e.get('parity')))" > gpurun_out/r04j_ops.txt 2>&1
cat gpurun_out/r04j_ops.txt
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  c2', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_cold'), d['clock_ramp']['ms_per_step'], d.get('parity'))" > gpurun_out/r04j_c2.txt 2>&1
cat gpurun_out/r04j_c2.txt
