mkdir -p gpurun_out; rm -f gpurun_out/r04_ops2.txt
for seg in default 4 8 12 16 24 48; do
  if [ $seg = default ]; then unset VIPS_HIP_CONV_U8_SEG; else export VIPS_HIP_CONV_U8_SEG=$seg; fi
  echo "== seg $seg" >> gpurun_out/r04_ops2.txt
  timeout 300 python bench.py --config ops --ops convi_3x3_u8,convi_5x5_u8,gaussblur_s2_u8,gaussblur_s8_u8 --steps 10 --warmup 3 --no-verify --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in d['ops']: print('  %-20s %.4f ms  frac %.3f' % (e['name'], e['ms'], e['frac']))" >> gpurun_out/r04_ops2.txt
done
cat gpurun_out/r04_ops2.txt
