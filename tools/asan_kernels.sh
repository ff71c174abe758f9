#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the KERNEL code: the product's .hip files compiled for
# host fibers (tests/emul) with the sanitizers on, linked with the product's host objects, and the GPU parity
# files run against that library under the mock HIP runtime.  An out-of-bounds read or write of an image, a
# table or the block's dynamic LDS, a misaligned access, a signed overflow, a shift out of range in a kernel
# shows here; the GPU would not say.  (float-cast-overflow is off: the device's conversions saturate and turn NaN
# into 0, and the kernels say where they count on it.)  usage: [TESTS='files'] tools/asan_kernels.sh [pytest args...]   (output: /tmp/vips_hip_kasan/)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT=/tmp/vips_hip_kasan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
CXX=/opt/rocm/lib/llvm/bin/clang++
EMUL=$ROOT/tests/emul
CSRC=$ROOT/libvips_amd/csrc
mkdir -p "$OUT"
make -C "$CSRC" >/dev/null
FLAGS="-std=c++17 -O1 -g -ffp-contract=off -fPIC -D__HIP_PLATFORM_AMD__ -DEMUL_EXACT_LDS -I$EMUL -I$CSRC -I$ROOT/include -I/opt/rocm/include \
 -Wno-unused-function -fsanitize=address,undefined -fno-sanitize=float-cast-overflow -fno-omit-frame-pointer -Wno-pass-failed"
pids=()
for src in "$EMUL"/*_emul.cpp "$EMUL"/emul.cpp; do
	obj="$OUT/$(basename "$src" .cpp).o"
	if [ ! -e "$obj" ] || [ "$src" -nt "$obj" ] || [ -n "$(find "$EMUL" "$CSRC" -maxdepth 1 \( -name '*.h' -o -name '*.hip' \) -newer "$obj" | head -1)" ]; then
		$CXX $FLAGS -c "$src" -o "$obj" &
		pids+=($!)
	fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
REPLACED=$(for s in "$EMUL"/*_emul.cpp; do echo "$CSRC/_obj/$(basename "$s" _emul.cpp).hip.o"; done)
PROD=$(ls "$CSRC"/_obj/*.o | grep -v -F "$REPLACED")
TORCH_LIB=$(python3 -c "import os, torch; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
g++ -shared -fPIC -o "$OUT/libvipship_kasan.so" "$OUT"/*.o $PROD -L"$TORCH_LIB" -l:libamdhip64.so -Wl,-rpath,/opt/rocm/lib -lpthread -ldl
python -c "import sys; sys.path.insert(0, '$ROOT'); import tests.test_host_glue_mock as m; assert m._build_mock()"
rm -f "$OUT"/san.log*
cd "$ROOT"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:detect_stack_use_after_return=0:log_path=$OUT/san.log \
UBSAN_OPTIONS=print_stacktrace=1:log_path=$OUT/san.log \
VIPS_HIP_LIBRARY=$OUT/libvipship_kasan.so \
LD_PRELOAD=$RT:$ROOT/tests/mock_hip/_build/libmockhip.so \
	python -m pytest ${TESTS:-tests} -m gpu -q --tb=no -p no:cacheprovider \
	--ignore=tests/test_module.py --ignore=tests/test_module_stream.py --ignore=tests/test_full_size_gpu.py \
	--ignore=tests/test_sharding.py --ignore=tests/test_multidevice_gpu.py --ignore=tests/test_threads_gpu.py \
	-k "not mfma_variants and not region_windows and not any_bands and not c2_full and not c2_quarter" "$@" | tail -3 || true
# (the one line ASan writes about makecontext / swapcontext is not a report)
if cat "$OUT"/san.log* 2>/dev/null | grep -qE "ERROR|SUMMARY|runtime error"; then
	echo "SANITIZER REPORTS:"
	cat "$OUT"/san.log* | grep -E "ERROR|SUMMARY|runtime error" | sort | uniq -c | sort -rn | head -40
	exit 1
fi
echo "sanitizers: clean"
