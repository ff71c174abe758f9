# HBM bytes fetched / written by every kernel of the ops table (one rocprofv3 --pmc pass each)
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
sum="python $repo/profiles/rocprof_summary.py"
db() { find $1 -name '*.db' | head -1; }
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/pl_$c -o out -- python $repo/bench.py --config ops --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2>&1
  echo "# rocprofv3 --pmc $c -- bench.py --config ops --steps 3 --warmup 1"
  $sum $(db /tmp/pl_$c) | cut -c1-220
done > $repo/gpurun_out/r05l_ops_traffic.txt 2>&1
cat $repo/gpurun_out/r05l_ops_traffic.txt
