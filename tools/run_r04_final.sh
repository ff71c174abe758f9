#!/bin/bash
# Round 4's closing run on one box: the whole GPU suite, the PMC passes that stamp
# profiles/traffic.json, a kernel trace of the driver's command, a kernel trace of the ops table
# (the per-function rocprof summary VERDICT round 3 asked for), the C3 instruction counters, the
# issue-rate probe and the full default bench.
tag=${1:-r04z}
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu -x --durations=5 2>&1 | tail -12) > gpurun_out/${tag}_tests.txt
cat gpurun_out/${tag}_tests.txt
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
sum="python $repo/profiles/rocprof_summary.py"
db() { find $1 -name '*.db' | head -1; }
for c in FETCH_SIZE WRITE_SIZE; do
  TUNE_LAUNCHES=8 rocprofv3 --pmc $c -d /tmp/pz_$c -o out -- python $repo/tools/one_reduce.py > /dev/null 2>&1
  echo "# TUNE_LAUNCHES=8 rocprofv3 --pmc $c -- python tools/one_reduce.py"
  $sum $(db /tmp/pz_$c) | grep -i "reduce_fused\|counter"
done > $repo/gpurun_out/${tag}_c2_pmc.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/pz_kt -o kt -- python $repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-verify > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-verify"; $sum $(db /tmp/pz_kt) | grep -v "^$" | head -8; } > $repo/gpurun_out/${tag}_c2_rocprofv3.txt 2>&1
ops="python $repo/bench.py --config ops --steps 10 --warmup 3 --no-cpu-baseline --no-verify"
rocprofv3 --kernel-trace --stats -d /tmp/pz_ops -o kt -- $ops > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $ops"; $sum $(db /tmp/pz_ops) | grep -v "^$" | head -60; } > $repo/gpurun_out/${tag}_ops_rocprofv3.txt 2>&1
c3="python $repo/bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-verify"
{
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  rocprofv3 --pmc $ctrs -d /tmp/pz_c3_${ctrs%% *} -o out -- $c3 > /dev/null 2>&1
  echo "# rocprofv3 --pmc $ctrs -- $c3"
  $sum $(db /tmp/pz_c3_${ctrs%% *}) | grep -i "convsep_stream\|counter"
done
} > $repo/gpurun_out/${tag}_c3_pmc.txt 2>&1
cd $repo
( timeout 120 tools/valu_probe2 ) > gpurun_out/${tag}_valu_probe2.txt 2>&1
cat gpurun_out/${tag}_c2_pmc.txt | cut -c1-170
cat gpurun_out/${tag}_c3_pmc.txt | cut -c1-170
(timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo rc=$?
python - <<PY
import json
l=json.load(open("gpurun_out/${tag}_bench.json"))
print({k:l[k] for k in ("ms_per_step","value")}, l["roofline"]["frac"], l["roofline"]["traffic"], l["clock_ramp"]["ms_per_step"])
for c in l["configs"]:
    print(c["name"], {k:v for k,v in c.items() if k in ("ms","frac","frac_hbm","ms_per_image","ms_module_whole_image","ms_module_strips_512m","ms_builtin_reduce")})
for e in l.get("ops", []):
    print("  %-28s %.4f ms  frac %.3f %s" % (e["name"], e["ms"], e["frac"], (e.get("parity") or {}).get("bit_exact")))
PY
