"""libvips_amd -- MI355X-native per-tile pixel pipeline behind the libvips API.

The product is ``libvips_amd/lib/libvipship.so`` (hand-written HIP for gfx950
behind the C ABI of ``include/vips_hip.h``); this package is the thin host-side
mirror used by tests and bench.py.  Importing it requires the built library --
there is no CPU fallback.
"""
from ._ffi import VipsHipError, lib, LIB_PATH, HEADER_PATH  # noqa: F401
from .image import Image, gaussmat, resize_sharpen_batch, FORMATS, KERNELS, PRECISIONS, INTERPRETATIONS  # noqa: F401


def init(device=0):
    from ._ffi import check

    check(lib.vips_hip_init(int(device)))


def vector_set_enabled(enabled):
    """vips_vector_set_enabled: select the arithmetic of a Highway-built libvips for convi on
    uchar images (8-bit mantissas, shared exponent).  Off by default."""
    lib.vips_hip_vector_set_enabled(1 if enabled else 0)


def vector_isenabled():
    return bool(lib.vips_hip_vector_isenabled())


def synchronize():
    from ._ffi import check

    check(lib.vips_hip_synchronize())


def gate_report():
    """{kernel gate name: (launches, total_ms)} since the last gate_reset()."""
    import ctypes

    buf = ctypes.create_string_buffer(1 << 16)
    lib.vips_hip_gate_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, n, ms = line.rsplit(" ", 2)
        out[name] = (int(n), float(ms))
    return out
