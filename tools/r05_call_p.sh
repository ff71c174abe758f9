# the ushort separable convolution on the matrix cores: parity, time (default / the vector-ALU kernel)
timeout 900 python -m pytest tests/test_conv_u8_gpu.py -x -q -m gpu 2>&1 | tail -3
show() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in l['ops']: print('  %-22s %.4f ms frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], (e.get('parity') or {}).get('bit_exact'), {k: v['mean_ms'] for k, v in e['kernels'].items()}))
"; }
for env in "A=1" "VIPS_HIP_NO_CONV_U16_MFMA=1"; do echo "# $env"; env $env python bench.py --config ops --ops gaussblur_s2_u16,gaussblur_s8_u16,gaussblur_s2_u8,gaussblur_s8_u8 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | show; done
VIPS_HIP_CONV_MFMA_DEBUG=1 python tools/u16blur_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
