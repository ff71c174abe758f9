mkdir -p gpurun_out; rm -f gpurun_out/r04_c4_tune3.txt
unset VIPS_HIP_RESIZE_SHARPEN
for cfg in "A=1" "VIPS_HIP_STREAM_BLOCKS=3328" "VIPS_HIP_STREAM_BLOCKS=4160" "VIPS_HIP_STREAM_BLOCKS=4992" "VIPS_HIP_STREAM_BLOCKS=5824" "VIPS_HIP_STREAM_BLOCKS=6656" "VIPS_HIP_STREAM_BLOCKS=9984"; do
  echo "== $cfg" >> gpurun_out/r04_c4_tune3.txt
  env $cfg timeout 300 python bench.py --config c4 --images 512 --steps 3 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels'])" >> gpurun_out/r04_c4_tune3.txt 2>&1
done
cat gpurun_out/r04_c4_tune3.txt
