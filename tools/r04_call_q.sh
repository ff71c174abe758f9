# round 4, last seconds of GPU time: the experimental forms of C3's epilogue (whole-size parity rows inside bench.py)
mkdir -p gpurun_out
export TMPDIR=/tmp
for f in 2 4 3 1; do
  echo "== VIPS_HIP_STREAM_EPI=$f"
  VIPS_HIP_STREAM_EPI=$f timeout 20 python bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['parity']['bit_exact'], d['parity']['max_ulp'])"
done > gpurun_out/r04q_c3.txt 2>&1
cat gpurun_out/r04q_c3.txt
