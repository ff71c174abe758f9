# round 4: where the integer horizontal pass of C3 spends its time -- issue rates of its instructions
# (tools/valu_probe3) and two timing builds of the kernel (no window test / no dot products)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 60 tools/valu_probe3 > gpurun_out/r04o_valu_probe3.txt 2>&1
cat gpurun_out/r04o_valu_probe3.txt
run() {
  echo "== $*"
  env "$@" timeout 80 python bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-verify 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
}
{
run VIPS_HIP_STREAM_INT=1
run VIPS_HIP_STREAM_INT=1 VIPS_HIP_STREAM_VAR=1
run VIPS_HIP_STREAM_INT=1 VIPS_HIP_STREAM_VAR=2
run VIPS_HIP_STREAM_INT=0
} > gpurun_out/r04o_c3.txt 2>&1
cat gpurun_out/r04o_c3.txt
