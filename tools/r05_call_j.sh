# HBM traffic of the vertical band kernel (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate passes)
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
sum="python $repo/profiles/rocprof_summary.py"
db() { find $1 -name '*.db' | head -1; }
for c in FETCH_SIZE WRITE_SIZE "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $c | cut -c1-8)
  rocprofv3 --pmc $c -d /tmp/pj_$n -o out -- python $repo/bench.py --config ops --ops resize_rgb_to_1000,reduce_rgb_7.3 --steps 5 --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2>&1
  echo "# rocprofv3 --pmc $c -- bench.py --config ops --ops resize_rgb_to_1000,reduce_rgb_7.3"
  $sum $(db /tmp/pj_$n) | grep -i "_band\|counter" | cut -c1-200
done
