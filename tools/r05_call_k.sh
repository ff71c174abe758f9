# the rows neighbouring blocks share: blocks of rows per wave (VIPS_HIP_BAND_GPW) and alternating directions
show() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in l['ops']: print('  %-22s %.4f ms frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], (e.get('parity') or {}).get('bit_exact'), {k: v['mean_ms'] for k, v in e['kernels'].items()}))
"; }
for env in "VIPS_HIP_BAND_GPW=1 VIPS_HIP_BAND_NO_ALTERNATE=1" "VIPS_HIP_BAND_GPW=1" "VIPS_HIP_BAND_GPW=2 VIPS_HIP_BAND_NO_ALTERNATE=1" "VIPS_HIP_BAND_GPW=2" "VIPS_HIP_BAND_GPW=4" "A=1"; do echo "# $env"; env $env python bench.py --config ops --ops resize_rgb_to_1000,thumbnail_500,reduce_rgb_7.3 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | show; done
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
sum="python $repo/profiles/rocprof_summary.py"
db() { find $1 -name '*.db' | head -1; }
for c in FETCH_SIZE; do
  rocprofv3 --pmc $c -d /tmp/pk_$c -o out -- python $repo/bench.py --config ops --ops resize_rgb_to_1000,reduce_rgb_7.3 --steps 5 --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2>&1
  echo "# rocprofv3 --pmc $c -- bench.py --config ops --ops resize_rgb_to_1000,reduce_rgb_7.3"
  $sum $(db /tmp/pk_$c) | grep -i "_band\|counter" | cut -c1-200
done
