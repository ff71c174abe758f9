#!/bin/bash
# The round's closing run on one box: the whole GPU suite, the PMC passes that stamp
# profiles/traffic.json, a kernel trace of the driver's command and the full default bench.
tag=${1:-r03z}
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -q -m gpu -x --durations=5 2>&1 | tail -12) > gpurun_out/${tag}_tests.txt
cat gpurun_out/${tag}_tests.txt
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  TUNE_LAUNCHES=8 rocprofv3 --pmc $c -d /tmp/pz_$c -o out -- python $repo/tools/one_reduce.py > /dev/null 2>&1
  echo "# TUNE_LAUNCHES=8 rocprofv3 --pmc $c -- python tools/one_reduce.py"
  python $repo/profiles/rocprof_summary.py $(find /tmp/pz_$c -name '*.db' | head -1) | grep -i "reduce_fused\|counter"
done > $repo/gpurun_out/${tag}_c2_pmc.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/pz_kt -o kt -- python $repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-verify > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-verify"; python $repo/profiles/rocprof_summary.py $(find /tmp/pz_kt -name '*.db' | head -1) | grep -v "^$" | head -8; } > $repo/gpurun_out/${tag}_c2_rocprofv3.txt 2>&1
cd $repo
cat gpurun_out/${tag}_c2_pmc.txt | cut -c1-170
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo rc=$?
python - <<PY
import json
l=json.load(open("gpurun_out/${tag}_bench.json"))
print({k:l[k] for k in ("ms_per_step","value")}, l["roofline"]["frac"], l["roofline"]["traffic"], l["clock_ramp"]["ms_per_step"])
for c in l["configs"]:
    print(c["name"], {k:v for k,v in c.items() if k in ("ms","frac","frac_hbm","ms_per_image","ms_module_whole_image","ms_module_strips_512m","ms_builtin_reduce")})
PY
