#!/bin/bash
# The GPU parity suite on the CPU: `pytest -m gpu` against libvipship_emul.so (every kernel file of the product
# compiled for host fibers, tests/emul) under the mock HIP runtime.  Left out: the libvips module's files (the
# plugin loads the real libvipship.so), the device-sized cases, the sweeps of the matrix-core kernel's variants
# (a minute each on fibers).  What fails here for a reason that is not a kernel: tests that need torch CUDA
# tensors (tests/test_sharding.py's GPU cases, test_multidevice_gpu's module case).
#   usage: tools/emul_gpu_suite.sh [log]      (8 cores: about 8 minutes)
log=${1:-/tmp/emul_gpu_suite.log}
cd "$(dirname "$0")/.." || exit 1
python -c "from tests.test_host_glue_mock import _build_mock; import sys; sys.exit(0 if _build_mock() else 1)" || exit 1
make -C tests/emul > /dev/null || exit 1
LD_PRELOAD=$PWD/tests/mock_hip/_build/libmockhip.so VIPS_HIP_LIBRARY=$PWD/tests/emul/_build/libvipship_emul.so \
  python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 \
  --ignore=tests/test_module.py --ignore=tests/test_module_stream.py --ignore=tests/test_full_size_gpu.py \
  -k "not mfma_variants and not region_windows and not any_bands and not c2_full and not c2_quarter" > "$log" 2>&1
tail -12 "$log"
