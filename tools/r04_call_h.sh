# round 4, GPU call H: segment height of the streaming uchar reducev (2 160 waves per launch at the default)
mkdir -p gpurun_out
export TMPDIR=/tmp
for seg in default 6 9 13 18; do
  if [ $seg = default ]; then unset VIPS_HIP_R16_SEG; else export VIPS_HIP_R16_SEG=$seg; fi
  echo "== seg $seg"
  timeout 200 python bench.py --config ops --ops reduce_rgb_7.3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in d['ops']: print('  %-28s %.4f ms  frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], e['kernels'], e.get('parity')))"
done > gpurun_out/r04i_ops.txt 2>&1
cat gpurun_out/r04i_ops.txt
