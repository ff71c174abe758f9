mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_conv_colour_gpu.py -x -q -k "colour or golden or c3 or gaussblur or convsep or c5" > gpurun_out/r04_colour_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04_colour_tests.txt; tail -4 gpurun_out/r04_colour_tests.txt
timeout 600 python bench.py --config ops --ops colourspace --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in d['ops']: print('  %-28s %.4f ms  frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], e['kernels'], e.get('parity')))"
