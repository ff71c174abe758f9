# the vertical band kernel: one chunk of rows in flight a wave (3 blocks a CU) against a ring of three (2 blocks)
mkdir -p gpurun_out
show() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in l['ops']: print('  %-22s %.4f ms frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], (e.get('parity') or {}).get('bit_exact'), {k: v['mean_ms'] for k, v in e['kernels'].items()}))
"; }
for env in "A=1" "VIPS_HIP_BAND_RING=1" "A=2" "VIPS_HIP_BAND_RING=1"; do echo "# $env"; env $env python bench.py --config ops --ops resize_rgb_to_1000,thumbnail_500 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | show; done
VIPS_HIP_BAND_RING=1 timeout 600 python -m pytest tests/test_reduce_band_gpu.py -x -q -m gpu -k resize_band 2>&1 | tail -3
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
sum="python $repo/profiles/rocprof_summary.py"
db() { find $1 -name '*.db' | head -1; }
i=0
for env in "A=1" "VIPS_HIP_BAND_RING=1"; do
  i=$((i+1))
  env $env rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d /tmp/pi_$i -o out -- python $repo/bench.py --config ops --ops resize_rgb_to_1000 --steps 5 --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2>&1
  echo "# $env rocprofv3 --pmc SQ_* -- bench.py --config ops --ops resize_rgb_to_1000"
  $sum $(db /tmp/pi_$i) | grep -i "shrinkv_reducev\|counter" | cut -c1-200
done
