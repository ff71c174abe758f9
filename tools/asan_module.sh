#!/bin/bash
# The libvips-side module (host/vips_hip_module.c: producer thread, host ring, generate) under
# AddressSanitizer + UBSan without a GPU: the module is rebuilt with clang's shared sanitizer
# runtime and drives tests/test_module_stream.py's child program against the mock HIP runtime with
# the convolution kernel on host fibers (real pixels, every access pattern, the injected failure).
# usage: tools/asan_module.sh        (output: /tmp/vips_hip_modasan/)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT=/tmp/vips_hip_modasan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
mkdir -p "$OUT"
python -c "import sys; sys.path.insert(0, '$ROOT'); import tests.test_host_glue_mock as m, tests.test_emul_resize_sharpen as e; assert m._build_mock() and e._build_emul()"
/opt/rocm/lib/llvm/bin/clang -std=gnu99 -O1 -g -fPIC -shared -fsanitize=address,undefined -shared-libsan -fno-omit-frame-pointer \
	-Wno-unused-function -o "$OUT/vips-hip.so" "$ROOT/host/vips_hip_module.c" -I"$ROOT/include" -I"$ROOT/oracle/_ref/gen" \
	-I/root/reference/libvips/include -I/opt/conda/include/glib-2.0 -I/opt/conda/lib/glib-2.0/include \
	-L"$ROOT/oracle/_ref/lib" -lvips -L"$ROOT/libvips_amd/lib" -lvipship -Wl,-rpath,"$ROOT/oracle/_ref/lib" \
	-Wl,-rpath,"$ROOT/libvips_amd/lib" -L/opt/conda/lib -Wl,-rpath,/opt/conda/lib -lgobject-2.0 -lgmodule-2.0 -lglib-2.0
python - "$ROOT" "$OUT" <<'PY'
import re, sys
root, out = sys.argv[1], sys.argv[2]
src = open(root + "/tests/test_module_stream.py").read()
child = re.search(r"CHILD = r'''(.*?)'''", src, re.S).group(1) % {"root": root}
child = child.replace("Ref.load_module()", "helpers.MODULE_LIB = %r\nRef.load_module()" % (out + "/vips-hip.so"))
open(out + "/child.py", "w").write(child)
PY
rm -f "$OUT"/san.log*
cd /tmp
ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:log_path=$OUT/san.log UBSAN_OPTIONS=print_stacktrace=1:log_path=$OUT/san.log \
LD_PRELOAD=$RT:$ROOT/tests/mock_hip/_build/libmockhip.so:$ROOT/tests/emul/_build/libvipship_emul.so \
	python "$OUT/child.py" | tail -1
if ls "$OUT"/san.log* >/dev/null 2>&1; then
	echo "SANITIZER REPORTS:"
	cat "$OUT"/san.log* | grep -E "ERROR|SUMMARY|runtime error" | sort | uniq -c
	exit 1
fi
echo "sanitizers: clean"
