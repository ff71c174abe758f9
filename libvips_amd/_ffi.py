"""ctypes binding of include/vips_hip.h (libvips_amd/lib/libvipship.so).

Plumbing only: every signature here is the C ABI's, nothing is computed in
Python.  The library is built by ``__graft_entry__.build()`` (hipcc, gfx950).
There is no CPU fallback: if the shared library is missing, import fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VIPS_HIP_LIBRARY: load another build of the same library (e.g. a sanitizer build, tools/asan_mock.sh)
LIB_PATH = os.environ.get("VIPS_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libvipship.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "vips_hip.h")


class VipsHipError(RuntimeError):
    """Mirrors libvips' error convention: -1 + a message in the error buffer."""


class Region(ctypes.Structure):
    """VipsHipRegion (include/vips_hip.h)."""

    _fields_ = [
        ("data", ctypes.c_void_p),
        ("left", ctypes.c_int),
        ("top", ctypes.c_int),
        ("width", ctypes.c_int),
        ("height", ctypes.c_int),
        ("im_width", ctypes.c_int),
        ("im_height", ctypes.c_int),
        ("bands", ctypes.c_int),
        ("format", ctypes.c_int),
        ("stride", ctypes.c_size_t),
    ]


class VHeader(ctypes.Structure):
    """VipsHipVHeader: the fields of a .v file header (include/vips_hip.h)."""

    _fields_ = [
        ("width", ctypes.c_int),
        ("height", ctypes.c_int),
        ("bands", ctypes.c_int),
        ("format", ctypes.c_int),
        ("coding", ctypes.c_int),
        ("interpretation", ctypes.c_int),
        ("xres", ctypes.c_float),
        ("yres", ctypes.c_float),
        ("xoffset", ctypes.c_int),
        ("yoffset", ctypes.c_int),
        ("msb_first", ctypes.c_int),
        ("data_offset", ctypes.c_longlong),
        ("data_size", ctypes.c_longlong),
    ]


class JpegHeader(ctypes.Structure):
    """VipsHipJpegHeader (include/vips_hip.h)."""

    _fields_ = [
        ("width", ctypes.c_int),
        ("height", ctypes.c_int),
        ("bands", ctypes.c_int),
        ("interpretation", ctypes.c_int),
        ("image_width", ctypes.c_int),
        ("image_height", ctypes.c_int),
        ("orientation", ctypes.c_int),
        ("has_icc", ctypes.c_int),
    ]


def _load():
    # PyTorch-ROCm carries its own HIP runtime (soname libamdhip64.so).  Import it first so
    # libvipship.so, which needs that soname, binds to the SAME runtime: device pointers,
    # streams and events can then be shared with torch, and there is one runtime at exit.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the HIP extension is the product; there is no fallback path)" % LIB_PATH
        )
    return ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)


lib = _load()

c_int, c_double, c_void_p, c_size_t, c_char_p = (
    ctypes.c_int,
    ctypes.c_double,
    ctypes.c_void_p,
    ctypes.c_size_t,
    ctypes.c_char_p,
)
P = ctypes.POINTER
RegionP = P(Region)

_SIGNATURES = {
    # runtime
    "vips_hip_init": (c_int, [c_int]),
    "vips_hip_shutdown": (None, []),
    "vips_hip_device_count": (c_int, []),
    "vips_hip_current_device": (c_int, []),
    "vips_hip_devices": (c_int, [P(c_int), c_int]),
    "vips_hip_error_buffer": (c_char_p, []),
    "vips_hip_error_clear": (None, []),
    "vips_hip_set_stream": (c_int, [c_void_p]),
    "vips_hip_get_stream": (c_void_p, []),
    "vips_hip_synchronize": (c_int, []),
    "vips_hip_malloc": (c_void_p, [c_size_t]),
    "vips_hip_free": (None, [c_void_p]),
    "vips_hip_malloc_host": (c_void_p, [c_size_t]),
    "vips_hip_free_host": (None, [c_void_p]),
    "vips_hip_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t]),
    "vips_hip_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t]),
    "vips_hip_memcpy_h2d_async": (c_int, [c_void_p, c_void_p, c_size_t]),
    "vips_hip_memcpy_d2h_async": (c_int, [c_void_p, c_void_p, c_size_t]),
    "vips_hip_stream_new": (c_void_p, []),
    "vips_hip_stream_free": (None, [c_void_p]),
    "vips_hip_event_synchronize": (c_int, [c_void_p]),
    "vips_hip_stream_wait_event": (c_int, [c_void_p]),
    "vips_hip_memcpy_d2d": (c_int, [c_void_p, c_void_p, c_size_t]),
    "vips_hip_memcpy2d_h2d": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_size_t]),
    "vips_hip_memcpy2d_d2h": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_size_t]),
    "vips_hip_pool_bytes": (c_size_t, []),
    "vips_hip_pool_trim": (None, []),
    "vips_hip_event_new": (c_void_p, []),
    "vips_hip_event_free": (None, [c_void_p]),
    "vips_hip_event_record": (c_int, [c_void_p]),
    "vips_hip_event_elapsed_ms": (c_double, [c_void_p, c_void_p]),
    "vips_hip_gate_enable": (None, [c_int]),
    "vips_hip_gate_reset": (None, []),
    "vips_hip_gate_query": (c_int, [c_char_p, P(c_double)]),
    "vips_hip_gate_report": (c_int, [c_char_p, c_int]),
    # reduce
    "vips_hip_reduce_new": (c_void_p, [c_int, c_double, c_int, c_int, c_double]),
    "vips_hip_reduce_free": (None, [c_void_p]),
    "vips_hip_reduce_get_n_point": (c_int, [c_void_p]),
    "vips_hip_reduce_get_out_size": (c_int, [c_void_p]),
    "vips_hip_reduce_get_offset": (c_double, [c_void_p]),
    "vips_hip_reduce_get_matrixs": (c_int, [c_void_p, c_int, P(ctypes.c_short)]),
    "vips_hip_reduce_get_matrixf": (c_int, [c_void_p, c_int, P(c_double)]),
    "vips_hip_reduce_get_points": (c_int, [c_int, c_double]),
    "vips_hip_reduceh_need": (None, [c_void_p, c_int, c_int, P(c_int), P(c_int)]),
    "vips_hip_reducev_need": (None, [c_void_p, c_int, c_int, P(c_int), P(c_int)]),
    "vips_hip_reduceh_gen": (c_int, [c_void_p, RegionP, RegionP]),
    "vips_hip_reducev_gen": (c_int, [c_void_p, RegionP, RegionP]),
    "vips_hip_reduceh_gen_tiled": (c_int, [c_void_p, RegionP, RegionP, c_int]),
    "vips_hip_reducev_gen_tiled": (c_int, [c_void_p, RegionP, RegionP, c_int]),
    "vips_hip_upsize_gen": (c_int, [RegionP, RegionP, c_double, c_double, c_double, c_double, c_int, c_int]),
    "vips_hip_affine_out_size": (c_int, [c_int, c_double]),
    "vips_hip_zoom_gen": (c_int, [RegionP, RegionP, c_int, c_int]),
    "vips_hip_subsample_gen": (c_int, [RegionP, RegionP, c_int, c_int]),
    "vips_hip_reduce_gen": (c_int, [c_void_p, c_void_p, RegionP, RegionP]),
    "vips_hip_reduce_gen_tiled": (c_int, [c_void_p, c_void_p, RegionP, RegionP, c_int]),
    # shrink
    "vips_hip_shrinkh_gen": (c_int, [c_int, RegionP, RegionP]),
    "vips_hip_shrinkv_gen": (c_int, [c_int, RegionP, RegionP]),
    "vips_hip_shrink_out_size": (c_int, [c_int, c_int, c_int]),
    # convolution
    "vips_hip_conv_new": (c_void_p, [P(c_double), c_int, c_int, c_double, c_double, c_int]),
    "vips_hip_conv_free": (None, [c_void_p]),
    "vips_hip_conv_get_nnz": (c_int, [c_void_p]),
    "vips_hip_conv_get_vector": (c_int, [c_void_p, P(c_int), P(c_int), P(c_int), c_int]),
    "vips_hip_conv_out_format": (c_int, [c_void_p, c_int]),
    "vips_hip_conv_gen": (c_int, [c_void_p, RegionP, RegionP]),
    "vips_hip_vector_set_enabled": (None, [c_int]),
    "vips_hip_vector_isenabled": (c_int, []),
    "vips_hip_set_exact_float": (None, [c_int]),
    "vips_hip_get_exact_float": (c_int, []),
    "vips_hip_conva_new": (c_void_p, [P(c_double), c_int, c_int, c_double, c_double, c_int, c_int]),
    "vips_hip_convasep_new": (c_void_p, [P(c_double), c_int, c_double, c_double, c_int]),
    "vips_hip_conva_free": (None, [c_void_p]),
    "vips_hip_conva_get_lines": (c_int, [c_void_p, P(c_int), P(c_int), c_int]),
    "vips_hip_conva_gen": (c_int, [c_void_p, RegionP, RegionP]),
    "vips_hip_convasep_gen": (c_int, [c_void_p, RegionP, RegionP, c_int]),
    "vips_hip_gaussmat": (c_int, [c_double, c_double, c_int, c_int, P(c_double), c_int, P(c_double)]),
    # colour
    "vips_hip_colour_gen": (c_int, [c_int, RegionP, RegionP]),
    "vips_hip_colour_route_gen": (c_int, [P(c_int), c_int, c_double, RegionP, RegionP]),
    "vips_hip_cast_gen": (c_int, [RegionP, RegionP]),
    "vips_hip_premultiply_gen": (c_int, [RegionP, RegionP, c_double, c_int, c_int]),
    "vips_hip_sharpen_gen": (c_int, [c_void_p, RegionP, RegionP, RegionP]),
    # images
    "vips_hip_image_new": (c_void_p, [c_int, c_int, c_int, c_int, c_int]),
    "vips_hip_image_new_from_memory": (c_void_p, [c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "vips_hip_image_new_from_device": (c_void_p, [c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "vips_hip_image_unref": (None, [c_void_p]),
    "vips_hip_image_unref_many": (None, [P(c_void_p), c_int]),
    "vips_hip_image_write_to_memory": (c_int, [c_void_p, c_void_p]),
    "vips_hip_image_get_data": (c_void_p, [c_void_p]),
    "vips_hip_image_get_device": (c_int, [c_void_p]),
    "vips_hip_image_get_width": (c_int, [c_void_p]),
    "vips_hip_image_get_height": (c_int, [c_void_p]),
    "vips_hip_image_get_bands": (c_int, [c_void_p]),
    "vips_hip_image_get_format": (c_int, [c_void_p]),
    "vips_hip_image_get_interpretation": (c_int, [c_void_p]),
    "vips_hip_image_get_stride": (c_size_t, [c_void_p]),
    "vips_hip_image_region": (None, [c_void_p, RegionP]),
    "vips_hip_set_fatstrip_height": (None, [c_int]),
    # operations
    "vips_hip_reduceh": (c_int, [c_void_p, P(c_void_p), c_double, c_int, c_double]),
    "vips_hip_reducev": (c_int, [c_void_p, P(c_void_p), c_double, c_int, c_double]),
    "vips_hip_reduce": (c_int, [c_void_p, P(c_void_p), c_double, c_double, c_int, c_double]),
    "vips_hip_shrinkh": (c_int, [c_void_p, P(c_void_p), c_int, c_int]),
    "vips_hip_shrinkv": (c_int, [c_void_p, P(c_void_p), c_int, c_int]),
    "vips_hip_shrink": (c_int, [c_void_p, P(c_void_p), c_double, c_double, c_int]),
    "vips_hip_resize": (c_int, [c_void_p, P(c_void_p), c_double, c_double, c_int, c_double]),
    "vips_hip_thumbnail_image": (c_int, [c_void_p, P(c_void_p), c_int, c_int, c_int, c_int]),
    "vips_hip_conv": (c_int, [c_void_p, P(c_void_p), P(c_double), c_int, c_int, c_double, c_double, c_int]),
    "vips_hip_convsep": (c_int, [c_void_p, P(c_void_p), P(c_double), c_int, c_double, c_double, c_int]),
    "vips_hip_vfile_read_header": (c_int, [c_char_p, P(VHeader)]),
    "vips_hip_image_new_from_vfile": (c_void_p, [c_char_p]),
    "vips_hip_image_write_to_vfile": (c_int, [c_void_p, c_char_p]),
    "vips_hip_thumbnail_find_jpegshrink": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "vips_hip_jpeg_read_header": (c_int, [c_char_p, c_int, P(JpegHeader)]),
    "vips_hip_jpeg_read_to_memory": (c_int, [c_char_p, c_int, c_void_p, c_size_t]),
    "vips_hip_image_new_from_jpeg": (c_void_p, [c_char_p, c_int]),
    "vips_hip_thumbnail": (c_int, [c_char_p, P(c_void_p), c_int, c_int, c_int, c_int, c_int]),
    "vips_hip_thumbnail_batch": (c_int, [P(c_char_p), c_int, P(c_void_p), c_char_p, c_int, c_int, c_int, c_int,
                                         c_int, c_int]),
    "vips_hip_thumbnail_image_crop": (c_int, [c_void_p, P(c_void_p), c_int, c_int, c_int, c_int, c_int]),
    "vips_hip_extract_area": (c_int, [c_void_p, P(c_void_p), c_int, c_int, c_int, c_int]),
    "vips_hip_conva": (c_int, [c_void_p, P(c_void_p), P(c_double), c_int, c_int, c_double, c_double, c_int, c_int]),
    "vips_hip_convasep": (c_int, [c_void_p, P(c_void_p), P(c_double), c_int, c_double, c_double, c_int]),
    "vips_hip_gaussblur": (c_int, [c_void_p, P(c_void_p), c_double, c_double, c_int]),
    "vips_hip_sharpen": (c_int, [c_void_p, P(c_void_p), c_double, c_double, c_double, c_double, c_double, c_double]),
    "vips_hip_colourspace": (c_int, [c_void_p, P(c_void_p), c_int]),
    "vips_hip_strips_new": (c_void_p, [c_int, c_int, c_int, c_int, c_int, P(c_int), c_int]),
    "vips_hip_strips_free": (None, [c_void_p]),
    "vips_hip_strips_count": (c_int, [c_void_p]),
    "vips_hip_strips_region": (c_int, [c_void_p, c_int, P(c_int), P(Region), P(Region)]),
    "vips_hip_strips_exchange": (c_int, [c_void_p]),
    "vips_hip_conv_strips": (c_int, [c_void_p, P(c_void_p), P(c_double), c_int, c_int, c_double, c_double, c_int]),
    "vips_hip_resize_sharpen_batch": (c_int, [P(c_void_p), c_int, P(c_void_p), c_double, c_int, c_double,
                                             c_double, c_double, c_double, c_double, c_double, c_double, c_int]),
    "vips_hip_resize_sharpen_batch_queue": (c_int, [P(c_void_p), c_int, P(c_void_p), c_double, c_int, c_double,
                                             c_double, c_double, c_double, c_double, c_double, c_double, c_int]),
    "vips_hip_gaussblur_colourspace": (c_int, [c_void_p, P(c_void_p), c_double, c_double, c_int, c_int]),
    "vips_hip_cast": (c_int, [c_void_p, P(c_void_p), c_int]),
    "vips_hip_premultiply": (c_int, [c_void_p, P(c_void_p), c_int]),
    "vips_hip_unpremultiply": (c_int, [c_void_p, P(c_void_p), c_int]),
}

MISSING = []
for _name, (_res, _args) in _SIGNATURES.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError:
        MISSING.append(_name)
        continue
    _fn.restype = _res
    _fn.argtypes = _args


def error_buffer():
    return lib.vips_hip_error_buffer().decode("utf-8", "replace")


def _raise(what):
    message = error_buffer().strip() or what
    lib.vips_hip_error_clear()
    raise VipsHipError(message)


def check(result, what="vips_hip"):
    """Turn the C convention (non-zero return + error buffer) into an exception."""
    if result != 0:
        _raise(what)
    return result


def check_handle(handle, what="vips_hip"):
    """NULL handle + error buffer -> exception."""
    if not handle:
        _raise(what)
    return handle
