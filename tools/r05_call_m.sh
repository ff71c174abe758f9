# reducev_u8_mfma with every other row of tiles bottom-up: parity, A/B, traffic; and C4's traffic
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_resample_gpu.py tests/test_reducev8_gpu.py tests/test_reduce_band_gpu.py -x -q -m gpu 2>&1 | tail -3
show() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in l['ops']: print('  %-22s %.4f ms frac %.3f %s %s' % (e['name'], e['ms'], e['frac'], (e.get('parity') or {}).get('bit_exact'), {k: v['mean_ms'] for k, v in e['kernels'].items()}))
"; }
for env in "VIPS_HIP_BAND_NO_ALTERNATE=1" "A=1" "VIPS_HIP_BAND_NO_ALTERNATE=1" "A=1"; do echo "# $env"; env $env python bench.py --config ops --ops reducev_8,reduce_rgb_8,reduce_rgba16_8 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | show; done
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
sum="python $repo/profiles/rocprof_summary.py"
db() { find $1 -name '*.db' | head -1; }
{
rocprofv3 --pmc FETCH_SIZE -d /tmp/pm_1 -o out -- python $repo/bench.py --config ops --ops reducev_8,reduce_rgb_8 --steps 5 --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2>&1
echo "# rocprofv3 --pmc FETCH_SIZE -- bench.py --config ops --ops reducev_8,reduce_rgb_8"
$sum $(db /tmp/pm_1) | grep -i "vh::\|counter" | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c -d /tmp/pm_c4$c -o out -- python $repo/bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2>&1
echo "# rocprofv3 --pmc $c -- bench.py --config c4 --steps 2 --warmup 1"
$sum $(db /tmp/pm_c4$c) | grep -i "vh::\|counter" | cut -c1-200
done
} > $repo/gpurun_out/r05m_traffic.txt 2>&1
cat $repo/gpurun_out/r05m_traffic.txt
