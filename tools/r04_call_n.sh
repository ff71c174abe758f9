# round 4: the integer horizontal pass of convsep_stream on the device -- its parity file, then C3 with
# the pass off and on (whole-size parity rows against the compiled reference inside bench.py)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 90 python -m pytest tests/test_convsep_int_gpu.py -x -q 2>&1 | tail -6 ) > gpurun_out/r04n_tests.txt 2>&1
tail -3 gpurun_out/r04n_tests.txt
for m in 0 1; do
  echo "== VIPS_HIP_STREAM_INT=$m"
  VIPS_HIP_STREAM_INT=$m timeout 80 python bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d.get('parity'), d.get('config'))"
done > gpurun_out/r04n_c3.txt 2>&1
cat gpurun_out/r04n_c3.txt
