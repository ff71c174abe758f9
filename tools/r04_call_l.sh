# round 4, GPU call L: C4 through the one-kernel resize + sharpen with the cube-root table's single- and
# double-precision forms, beside the shipped two-kernel pair (256 images; thumbnails checked against the
# compiled reference in each run)
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "A=1" "VIPS_HIP_RESIZE_SHARPEN=1" "VIPS_HIP_RESIZE_SHARPEN=1 VIPS_HIP_CBRT_F64=1"; do
  echo "== $cfg"
  env $cfg timeout 200 python bench.py --config c4 --images 256 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels'], d.get('parity'))"
done > gpurun_out/r04m_c4.txt 2>&1
cat gpurun_out/r04m_c4.txt
