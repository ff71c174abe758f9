# round 4: the colour epilogue of C3 with its table reads batched (VAR 3) and staggered over the waves
# of a SIMD (VAR 4): parity of the fused blur + colourspace cases under each build, then the timings
# (whole-size parity rows against the compiled reference inside bench.py)
mkdir -p gpurun_out
export TMPDIR=/tmp
K='convsep or gaussblur or c3_pipeline'
{
echo "== default"; timeout 100 python -m pytest tests/test_convsep_int_gpu.py tests/test_conv_colour_gpu.py -k "$K or integer_image or almost or float_image" -x -q 2>&1 | tail -3
for v in 3 4 5 6; do
  echo "== VIPS_HIP_STREAM_VAR=$v"
  VIPS_HIP_STREAM_VAR=$v timeout 60 python -m pytest tests/test_convsep_int_gpu.py tests/test_conv_colour_gpu.py -k "integer_image_blur_colourspace or gaussblur_colourspace_fused or c3_pipeline" -x -q 2>&1 | tail -3
done
} > gpurun_out/r04p_tests.txt 2>&1
cat gpurun_out/r04p_tests.txt
run() {
  echo "== $*"
  env "$@" timeout 80 python bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['parity']['bit_exact'], d['parity']['max_ulp'])"
}
{
run VIPS_HIP_STREAM_VAR=0
run VIPS_HIP_STREAM_VAR=3
run VIPS_HIP_STREAM_VAR=4
run VIPS_HIP_STREAM_VAR=5
run VIPS_HIP_STREAM_VAR=6
run VIPS_HIP_STREAM_VAR=4 VIPS_HIP_STREAM_INT=0
} > gpurun_out/r04p_c3.txt 2>&1
cat gpurun_out/r04p_c3.txt
