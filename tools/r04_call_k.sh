# round 4, last GPU call: smoke(), the module's GPU tests and the streaming-kernel files on the final tree
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > gpurun_out/r04l_smoke.txt 2>&1
cat gpurun_out/r04l_smoke.txt
( timeout 600 python -m pytest tests/test_module_stream.py tests/test_module.py tests/test_shrinkh_u8_gpu.py tests/test_reduceh_u8_gpu.py tests/test_reducev8_gpu.py tests/test_conv_u16_gpu.py tests/test_conv_u8_gpu.py -m gpu -q 2>&1 | tail -4 ) > gpurun_out/r04l_tests.txt 2>&1
tail -2 gpurun_out/r04l_tests.txt
