#!/bin/bash
# Host-side memory / UB check without a GPU: build libvipship.so with AddressSanitizer and
# UndefinedBehaviorSanitizer on the HOST code (device code untouched), then run the whole
# `-m gpu` test-suite against the mock HIP runtime of tests/mock_hip (kernels do nothing).
# Every test "fails" on its pixel comparison; what matters is the sanitizer log, which must
# stay empty.  usage: tools/asan_mock.sh [pytest args...]   (output: /tmp/vips_hip_asan/)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT=/tmp/vips_hip_asan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
mkdir -p "$OUT"
make -j"$(nproc)" -C "$ROOT/libvips_amd/csrc" OBJDIR="$OUT/obj" OUT="$OUT/libvipship.so" \
	CXXFLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden \
 -Wall -Wno-unused-function -I$ROOT/include -I$ROOT/libvips_amd/csrc \
 -fsanitize=address,undefined -fno-gpu-sanitize -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
 -DVIPS_HIP_HAVE_JPEGLIB -idirafter /opt/conda/include" >/dev/null
python -c "import sys; sys.path.insert(0, '$ROOT'); import tests.test_host_glue_mock as m; assert m._build_mock()"
rm -f "$OUT"/san.log*
cd /tmp
ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:log_path=$OUT/san.log \
UBSAN_OPTIONS=print_stacktrace=1:log_path=$OUT/san.log \
VIPS_HIP_LIBRARY=$OUT/libvipship.so \
LD_PRELOAD=$RT:$ROOT/tests/mock_hip/_build/libmockhip.so \
	python -m pytest "$ROOT/tests" -m gpu -q --tb=no -p no:cacheprovider -k "not Module and not module" "$@" | tail -1 || true
if ls "$OUT"/san.log* >/dev/null 2>&1; then
	echo "SANITIZER REPORTS:"
	cat "$OUT"/san.log* | grep -E "ERROR|SUMMARY|runtime error" | sort | uniq -c
	exit 1
fi
echo "sanitizers: clean"
