#!/bin/bash
# usage: tools/profile_configs.sh <tag>  -> gpurun_out/<tag>_configs_rocprofv3.txt
# rocprofv3 kernel trace of tools/bench_configs.py (C3, C4, C5 kernels)
tag=$1
repo=$(pwd)
mkdir -p $repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg -o kt -- python $repo/tools/bench_configs.py > /tmp/cfg.log 2>&1
{
  echo "# rocprofv3 --kernel-trace --stats -- python tools/bench_configs.py   (C3 32768^2x3 f32, C4 8192^2x3 u8, C5 16384x2048 u16)"
  python $repo/profiles/rocprof_summary.py $(find /tmp/prof_cfg -name '*.db' | head -1) | grep -v "at::native" | head -40
} > $repo/gpurun_out/${tag}_configs_rocprofv3.txt 2>&1
