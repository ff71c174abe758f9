repo=$(pwd)
out=$repo/gpurun_out/r03x2_c4_pmc.txt
cd /tmp && export TMPDIR=/tmp
cmd="python $repo/bench.py --config c4 --images 256 --steps 1 --warmup 1 --no-cpu-baseline --no-verify"
i=0
{
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs -d /tmp/px_$i -o out -- $cmd > /dev/null 2>&1
  echo "# rocprofv3 --pmc $ctrs -- $cmd"
  python $repo/profiles/rocprof_summary.py $(find /tmp/px_$i -name '*.db' | head -1) | grep -i "resize_stream\|sharpen_fused\|counter"
done
} > $out 2>&1
cut -c1-170 $out
