#!/bin/bash
# What do the L2's memory-side counters say about the resize kernel's output stores (1.5 % of its
# bytes, 7-20 % of its time) -- against the same stores in the load-skeleton probe, which are free?
repo=$(pwd)
out=$repo/gpurun_out/${1:-r03}_c4_store_pmc.txt
hipcc --offload-arch=gfx950 -O3 tools/c4_load_probe.hip -o /tmp/c4lp 2>/dev/null
cd /tmp && export TMPDIR=/tmp
sum="python $repo/profiles/rocprof_summary.py"
i=0
for ctrs in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" \
            "TCC_WRITE_sum TCC_WRITE_SECTORS_sum TCC_NORMAL_WRITEBACK_sum TCC_TAG_STALL_sum" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WR_UNCACHED_32B_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  for v in "stores:" "nostore:VIPS_HIP_STREAM_DEBUG=2"; do
    ROUNDS=1 rocprofv3 --pmc $ctrs -d /tmp/ps_${i}_${v%%:*} -o out -- python $repo/tools/time_resize64.py "$v" > /dev/null 2>&1
    echo "# rocprofv3 --pmc $ctrs -- python tools/time_resize64.py $v"
    $sum $(find /tmp/ps_${i}_${v%%:*} -name '*.db' | head -1) | grep -i "resize_stream"
  done
  rocprofv3 --pmc $ctrs -d /tmp/ps_${i}_probe -o out -- /tmp/c4lp > /dev/null 2>&1
  echo "# rocprofv3 --pmc $ctrs -- c4_load_probe (walk<512,4,8,false,6,STORE,JIT>: the 6th argument is the store mode)"
  $sum $(find /tmp/ps_${i}_probe -name '*.db' | head -1) | grep "walk<512, 4, 8, false, 6, [012]"
done > $out 2>&1
cut -c1-175 $out
