show() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in l['ops']: print('  %-22s %.4f ms frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], (e.get('parity') or {}).get('bit_exact'), {k: v['mean_ms'] for k, v in e['kernels'].items()}))
"; }
for env in "A=1" "VIPS_HIP_REDUCEV8_OHT=24" "VIPS_HIP_REDUCEV8_OHT=32" "VIPS_HIP_REDUCEV8_OHT=48" "VIPS_HIP_REDUCEV8_OHT=64"; do echo "# $env"; env $env python bench.py --config ops --ops reducev_8,reduce_rgb_8 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | show; done
