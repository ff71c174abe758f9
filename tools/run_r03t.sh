#!/bin/bash
# C4 at the whole batch (1024 images): where the time between the chunks' kernels goes (a kernel
# trace read launch by launch), the sharpen partition's size, then the round's closing run.
mkdir -p gpurun_out
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pt_c4 -o kt -- python $repo/bench.py --config c4 --images 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-verify > /tmp/pt_c4.log 2>&1
{ echo "# rocprofv3 --kernel-trace -- python bench.py --config c4 --images 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-verify"; tail -1 /tmp/pt_c4.log | cut -c1-400; python $repo/tools/kernel_timeline.py $(find /tmp/pt_c4 -name '*.db' | head -1) resize_stream,sharpen_fused 34; } > $repo/gpurun_out/r03t_c4_timeline.txt 2>&1
cd $repo
cat gpurun_out/r03t_c4_timeline.txt
for share in 48 56 64 80; do
  VIPS_HIP_BATCH_SHARPEN_CUS=$share C4_IMAGES=512 timeout 300 python tools/time_c4.py "share$share:" 2>&1 | tail -1
done | tee gpurun_out/r03t_c4_share.txt
bash tools/run_r03_final.sh r03t
