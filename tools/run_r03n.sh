mkdir -p gpurun_out
(timeout 400 python -m pytest tests/test_module.py tests/test_multidevice_gpu.py -x -q -m gpu 2>&1 | tail -3) > gpurun_out/r03n_tests.txt
cat gpurun_out/r03n_tests.txt
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  TUNE_LAUNCHES=8 rocprofv3 --pmc $c -d /tmp/p3n_$c -o out -- python $repo/tools/one_reduce.py > /dev/null 2>&1
  echo "# TUNE_LAUNCHES=8 rocprofv3 --pmc $c -- python tools/one_reduce.py"
  python $repo/profiles/rocprof_summary.py $(find /tmp/p3n_$c -name '*.db' | head -1) | grep -i "reduce_fused\|counter"
done > $repo/gpurun_out/r03n_c2_pmc.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p3n_kt -o kt -- python $repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-verify > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-verify"; python $repo/profiles/rocprof_summary.py $(find /tmp/p3n_kt -name '*.db' | head -1) | grep -v "^$" | head -8; } > $repo/gpurun_out/r03n_c2_rocprofv3.txt 2>&1
cd $repo
cat gpurun_out/r03n_c2_pmc.txt gpurun_out/r03n_c2_rocprofv3.txt | cut -c1-170
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/r03n_bench.json 2> gpurun_out/r03n_bench.err; echo rc=$?
python - <<PY
import json
l=json.load(open("gpurun_out/r03n_bench.json"))
print({k:l[k] for k in ("ms_per_step","value")}, l["roofline"]["frac"], l["roofline"]["traffic"], l["clock_ramp"]["ms_per_step"])
for c in l["configs"]:
    print(c["name"], {k:v for k,v in c.items() if k in ("ms","frac","frac_hbm","ms_per_image","ms_module_whole_image","ms_module_strips_512m","ms_builtin_reduce")})
PY
