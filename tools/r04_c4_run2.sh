set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_resample_gpu.py -x -q -k "resize_sharpen or batch" > gpurun_out/r04_c4_tests2.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r04_c4_tests2.txt
tail -5 gpurun_out/r04_c4_tests2.txt
rm -f gpurun_out/r04_c4_tune2.txt
for cfg in "A=1" "VIPS_HIP_NO_RESIZE_SHARPEN=1" "VIPS_HIP_RSH_TW=71" "VIPS_HIP_RSH_TW=77" "VIPS_HIP_STREAM_BLOCKS=4096" "VIPS_HIP_STREAM_BLOCKS=8192" "VIPS_HIP_STREAM_WINDOW=12" "VIPS_HIP_STREAM_WINDOW=15"; do
  echo "== $cfg" >> gpurun_out/r04_c4_tune2.txt
  env $cfg timeout 300 python bench.py --config c4 --images 256 --steps 3 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels'])" >> gpurun_out/r04_c4_tune2.txt 2>&1
done
cat gpurun_out/r04_c4_tune2.txt
