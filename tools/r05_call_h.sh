# the band resize chain (shrinkv + reducev fused on the matrix cores, shrinkh, reduceh): parity, A/B, fuzz
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_reduce_band_gpu.py tests/test_resample_gpu.py -x -q -m gpu 2>&1 | tail -5
show() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in l['ops']: print('  %-22s %.4f ms frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], (e.get('parity') or {}).get('bit_exact'), {k: v['mean_ms'] for k, v in e['kernels'].items()}))
"; }
for env in "A=1" "VIPS_HIP_NO_RESIZE_BAND=1"; do echo "# $env"; env $env python bench.py --config ops --ops resize_rgb_to_1000,thumbnail_500 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | show; done
VIPS_HIP_RESIZE_BAND_MIN=0 timeout 200 python tools/fuzz_gpu.py 90 77 resize 2>&1 | tail -3
VIPS_HIP_RESIZE_BAND_MIN=0 timeout 200 python tools/fuzz_gpu.py 60 78 thumb 2>&1 | tail -3
