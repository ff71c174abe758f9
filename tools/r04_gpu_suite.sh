mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r04_gpu_suite.txt 2>&1; echo "rc=$?" >> gpurun_out/r04_gpu_suite.txt; tail -8 gpurun_out/r04_gpu_suite.txt
