#!/bin/bash
# TEST INFRASTRUCTURE ONLY -- builds the *reference itself* (libvips 8.19.0, scalar C
# paths: no Highway, no ORC) from the sources where they lie under /root/reference
# into oracle/_ref/.  Nothing is copied into the repo; oracle/_ref/ is git-ignored
# but travels to the GPU box with the gpurun snapshot.
#
# Recipe follows SURVEY.md section 8(c).  The reference's own build system (meson)
# is NOT used: we compile every libvips/**/*.c|cpp except deprecated/, module/ and
# libnsgif/test directly with gcc/g++ against the glib in /opt/conda.
#
# Usage: oracle/build_ref.sh            (no-op when /root/reference is absent)
set -euo pipefail

REF=${VIPS_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
GLIB_PREFIX=${GLIB_PREFIX:-/opt/conda}
JOBS=${JOBS:-$(nproc)}

if [ ! -d "$REF/libvips" ]; then
	echo "build_ref: $REF not present, keeping prebuilt oracle/_ref" >&2
	exit 0
fi

if [ -f "$OUT/lib/libvips.so" ] && [ -z "${FORCE:-}" ]; then
	echo "build_ref: $OUT/lib/libvips.so already built (FORCE=1 to rebuild)" >&2
	exit 0
fi

mkdir -p "$OUT/gen/vips" "$OUT/obj" "$OUT/lib" "$OUT/bin"
GEN="$OUT/gen"
INC="$REF/libvips/include"

# 1. enumtypes.{h,c}: the 16 enum headers named in libvips/include/vips/meson.build:57-74
ENUM_HDRS=""
for h in resample memory create foreign arithmetic conversion util image colour \
	operation convolution morphology draw basic object region; do
	ENUM_HDRS="$ENUM_HDRS $INC/vips/$h.h"
done
"$GLIB_PREFIX/bin/glib-mkenums" --template "$INC/vips/enumtypes.h.in" $ENUM_HDRS \
	>"$GEN/vips/enumtypes.h" 2>/dev/null
"$GLIB_PREFIX/bin/glib-mkenums" --template "$INC/vips/enumtypes.c.in" $ENUM_HDRS \
	>"$GEN/enumtypes.c" 2>/dev/null

# 2. vipsmarshal.{h,c}  (libvips/iofuncs/meson.build:51-55)
"$GLIB_PREFIX/bin/glib-genmarshal" --prefix=vips --header \
	"$REF/libvips/iofuncs/vipsmarshal.list" >"$GEN/vipsmarshal.h"
"$GLIB_PREFIX/bin/glib-genmarshal" --prefix=vips --body --include-header=vipsmarshal.h \
	"$REF/libvips/iofuncs/vipsmarshal.list" >"$GEN/vipsmarshal.c"

# 3. version.h  (meson.build:2,27-29)
sed -e 's/@VIPS_VERSION@/8.19.0/' -e 's/@VIPS_VERSION_STRING@/8.19.0/' \
	-e 's/@VIPS_MAJOR_VERSION@/8/' -e 's/@VIPS_MINOR_VERSION@/19/' \
	-e 's/@VIPS_MICRO_VERSION@/0/' -e 's/@LIBRARY_CURRENT@/63/' \
	-e 's/@LIBRARY_REVISION@/0/' -e 's/@LIBRARY_AGE@/21/' \
	-e 's/@VIPS_CONFIG@/oracle build: scalar C paths, no hwy, no orc/' \
	-e 's/@VIPS_ENABLE_DEPRECATED@/0/' \
	"$INC/vips/version.h.in" >"$GEN/vips/version.h"

# 4. config.h: no HAVE_HWY / HAVE_ORC / modules / deprecated.  The one codec: the IJG libjpeg that
# ships in $GLIB_PREFIX (jpeglib.h + libjpeg.so), for the shrink-on-load thumbnail path
# (foreign/jpeg2vips.c).  No libexif, no lcms: no auto-rotate, no ICC.
JPEG_DEFINE=""
JPEG_LIB=""
if [ -f "$GLIB_PREFIX/include/jpeglib.h" ] && [ -e "$GLIB_PREFIX/lib/libjpeg.so" ]; then
	JPEG_DEFINE="#define HAVE_JPEG 1"
	JPEG_LIB="-ljpeg"
fi
cat >"$GEN/config.h" <<EOF
#ifndef ORACLE_CONFIG_H
#define ORACLE_CONFIG_H
#define G_LOG_DOMAIN "VIPS"
#define GETTEXT_PACKAGE "vips8.19"
#define VIPS_PREFIX "$OUT"
#define VIPS_LIBDIR "$OUT/lib"
#define VIPS_ICC_DIR "/usr/share/color/icc"
#define VIPS_EXEEXT ""
#define HAVE_UNISTD_H 1
#define HAVE_SYS_FILE_H 1
#define HAVE_SYS_MMAN_H 1
#define HAVE_SYS_PARAM_H 1
#define HAVE_POSIX_MEMALIGN 1
#define HAVE_MEMALIGN 1
#define HAVE_PPM 1
#define HAVE_ANALYZE 1
#define HAVE_RADIANCE 1
$JPEG_DEFINE
#define _VIPS_PUBLIC __attribute__((visibility("default")))
#endif
EOF

CFLAGS_COMMON="-O3 -DHAVE_CONFIG_H -DG_DISABLE_CAST_CHECKS -DG_DISABLE_CHECKS -DG_DISABLE_ASSERT \
 -I$GEN -I$GEN/vips -I$INC -I$REF/libvips -I$GLIB_PREFIX/include/glib-2.0 \
 -I$GLIB_PREFIX/lib/glib-2.0/include -idirafter $GLIB_PREFIX/include -fPIC -w"

# 5. compile
cd "$REF/libvips"
SRCS=$(find . \( -name '*.c' -o -name '*.cpp' \) \
	-not -path './deprecated/*' -not -path './module/*' \
	-not -path './foreign/libnsgif/test/*' | sort)
compile_one() {
	src="$1"
	obj="$OUT/obj/$(echo "${src#./}" | tr '/' '_').o"
	if [ -f "$obj" ] && [ "$obj" -nt "$src" ]; then return 0; fi
	case "$src" in
	*.cpp) g++ -std=c++14 $CFLAGS_COMMON -I"$(dirname "$src")" -c "$src" -o "$obj" ;;
	*) gcc -std=gnu99 $CFLAGS_COMMON -I"$(dirname "$src")" -c "$src" -o "$obj" ;;
	esac
}
export -f compile_one
export OUT CFLAGS_COMMON
echo "$SRCS" | xargs -P "$JOBS" -I{} bash -c 'compile_one "$@"' _ {}
gcc -std=gnu99 $CFLAGS_COMMON -c "$GEN/enumtypes.c" -o "$OUT/obj/gen_enumtypes.o"
gcc -std=gnu99 $CFLAGS_COMMON -c "$GEN/vipsmarshal.c" -o "$OUT/obj/gen_vipsmarshal.o"

# 6. link
g++ -shared "$OUT"/obj/*.o -o "$OUT/lib/libvips.so" \
	-L"$GLIB_PREFIX/lib" -Wl,-rpath,"$GLIB_PREFIX/lib" \
	-lgio-2.0 -lgobject-2.0 -lgmodule-2.0 -lglib-2.0 -lexpat $JPEG_LIB -lm -lpthread

# 7. the reference's own CLI tools, unchanged, for config C1
for t in vips vipsthumbnail vipsheader; do
	gcc -std=gnu99 $CFLAGS_COMMON "$REF/tools/$t.c" -o "$OUT/bin/$t" \
		-L"$OUT/lib" -lvips -Wl,-rpath,"$OUT/lib" \
		-L"$GLIB_PREFIX/lib" -Wl,-rpath,"$GLIB_PREFIX/lib" \
		-lgio-2.0 -lgobject-2.0 -lgmodule-2.0 -lglib-2.0 -lm
done

# headers needed to compile the libvips-side module (host/ in this repo) on a box
# without /root/reference are NOT copied: the module is built here, the .so travels.
rm -rf "$OUT/obj"
echo "build_ref: built $OUT/lib/libvips.so"
