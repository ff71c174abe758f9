#!/bin/bash
# Round 4's closing run on the final tree: the whole GPU suite (no -x: every failure is listed), a
# kernel trace of the ops table, the full default bench.  (PMC passes and the C2 / C3 traces of the
# same kernels: tools/run_r04_final.sh, profiles/r04f_*.)
tag=${1:-r04z}
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -14) > gpurun_out/${tag}_tests.txt
tail -3 gpurun_out/${tag}_tests.txt
(timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo rc=$?
python - <<PY
import json
l=json.load(open("gpurun_out/${tag}_bench.json"))
print({k:l[k] for k in ("ms_per_step","value")}, l["roofline"]["frac"], l["roofline"]["traffic"], l["clock_ramp"]["ms_per_step"])
for c in l["configs"]:
    print(c["name"], {k:v for k,v in c.items() if k in ("ms","frac","frac_hbm","frac_of_fp64_stream","ms_per_image","ms_module_whole_image","ms_module_strips_512m","ms_builtin_reduce")})
for e in l.get("ops", []):
    print("  %-28s %.4f ms  frac %.3f %s" % (e["name"], e["ms"], e["frac"], (e.get("parity") or {}).get("bit_exact")))
PY
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
ops="python $repo/bench.py --config ops --steps 10 --warmup 3 --no-cpu-baseline --no-verify"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pz_ops -o kt -- $ops > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $ops"; python $repo/profiles/rocprof_summary.py $(find /tmp/pz_ops -name '*.db' | head -1) | grep -v "^$" | head -60; } > $repo/gpurun_out/${tag}_ops_rocprofv3.txt 2>&1
