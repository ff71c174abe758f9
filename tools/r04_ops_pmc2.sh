# round 4: HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE) and wave counters of the round's new single-call kernels
mkdir -p gpurun_out
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
( OPS_ENTRY=shrinkh_4 bash tools/pmc.sh shrinkh tools/ops_pmc_case.py shrinkh_u8 "FETCH_SIZE" "WRITE_SIZE" "$SQ"
  OPS_ENTRY=reduceh_8 bash tools/pmc.sh reduceh tools/ops_pmc_case.py reduceh_u8p "FETCH_SIZE" "WRITE_SIZE" "$SQ"
  OPS_ENTRY=convi_3x3_u16 bash tools/pmc.sh c16 tools/ops_pmc_case.py conv_u16 "FETCH_SIZE" "WRITE_SIZE" "$SQ"
  OPS_ENTRY=reduce_rgb_7.3 bash tools/pmc.sh rv8 tools/ops_pmc_case.py reducev8 "FETCH_SIZE" "WRITE_SIZE" "$SQ" ) > gpurun_out/r04_ops_pmc2.txt 2>&1
cat gpurun_out/r04_ops_pmc2.txt | cut -c1-170
