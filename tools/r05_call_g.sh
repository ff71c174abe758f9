show() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in l['ops']: print('  %-22s %.4f ms frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], (e.get('parity') or {}).get('bit_exact'), {k: v['mean_ms'] for k, v in e['kernels'].items()}))
"; }
for env in "A=1" "VIPS_HIP_NO_RESIZE_STREAMG=1" "VIPS_HIP_NO_RESIZE_STREAM=1 VIPS_HIP_NO_RESIZE_TAIL=1"; do echo "# $env"; env $env python bench.py --config ops --ops resize_rgb_to_1000,thumbnail_500 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | show; done
