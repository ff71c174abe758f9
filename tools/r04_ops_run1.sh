mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_u8_gpu.py -x -q > gpurun_out/r04_conv_u8_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04_conv_u8_tests.txt; tail -4 gpurun_out/r04_conv_u8_tests.txt
timeout 600 python bench.py --config ops --ops convi,gaussblur --steps 5 --warmup 2 > gpurun_out/r04_ops1.json 2> gpurun_out/r04_ops1.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_ops1.json').read().strip().splitlines()[-1])
for e in d['ops']:
    print("%-28s ms %8.4f frac %.3f  %-30s par %s" % (e['name'], e['ms'], e['frac'], (e['kernel'] or '')[:30], e.get('parity',{}).get('bit_exact')))
PY
