"""Host-side mirror of the reference's image API for the hot path.

``Image`` wraps a device-resident ``VipsHipImage`` and exposes the operations
with the names / argument meaning / error behaviour of the reference's Python
binding (pyvips, which the reference's own test-suite uses:
test/test-suite/test_resample.py, test_convolution.py, test_colour.py), so the
parity tests read like the reference's tests.  Every method is a single call
through the C ABI (include/vips_hip.h); no pixel is touched in Python.
"""
import ctypes
import os

import numpy as np

from . import _ffi
from ._ffi import lib, check, check_handle

# VipsBandFormat (include/vips/image.h:120-133)
FORMATS = {
    "uchar": 0,
    "char": 1,
    "ushort": 2,
    "short": 3,
    "uint": 4,
    "int": 5,
    "float": 6,
    "complex": 7,
    "double": 8,
    "dpcomplex": 9,
}
FORMAT_NAMES = {v: k for k, v in FORMATS.items()}
FORMAT_DTYPES = {
    0: np.uint8,
    1: np.int8,
    2: np.uint16,
    3: np.int16,
    4: np.uint32,
    5: np.int32,
    6: np.float32,
    7: np.complex64,
    8: np.float64,
    9: np.complex128,
}
DTYPE_FORMATS = {np.dtype(v): k for k, v in FORMAT_DTYPES.items()}

# VipsKernel (include/vips/resample.h:41-51)
KERNELS = {
    "nearest": 0,
    "linear": 1,
    "cubic": 2,
    "mitchell": 3,
    "lanczos2": 4,
    "lanczos3": 5,
    "mks2013": 6,
    "mks2021": 7,
}
# VipsPrecision (include/vips/basic.h:106-110)
PRECISIONS = {"integer": 0, "float": 1, "approximate": 2}
# VipsSize (include/vips/resample.h), VipsInteresting (include/vips/conversion.h:97-107)
SIZES = {"both": 0, "up": 1, "down": 2, "force": 3}
INTERESTING = {"none": 0, "centre": 1, "entropy": 2, "attention": 3, "low": 4, "high": 5, "all": 6}
# VipsInterpretation (include/vips/image.h:94-118)
INTERPRETATIONS = {
    "multiband": 0,
    "b-w": 1,
    "xyz": 12,
    "lab": 13,
    "labs": 21,
    "srgb": 22,
    "rgb16": 25,
    "grey16": 26,
    "scrgb": 28,
}
INTERPRETATION_NAMES = {v: k for k, v in INTERPRETATIONS.items()}


def _enum(table, value, what):
    if isinstance(value, str):
        try:
            return table[value.lower()]
        except KeyError:
            raise ValueError("unknown %s %r" % (what, value))
    return int(value)


def _guess_interpretation(bands, fmt):
    # vips_image_new_from_memory defaults (iofuncs/image.c): multiband; the
    # reference's tests set interpretation explicitly via copy().
    return 0


class Image(object):
    """A device-resident image (VipsHipImage)."""

    def __init__(self, handle, keepalive=None):
        check_handle(handle, "image")
        self._h = ctypes.c_void_p(handle)
        self._keepalive = keepalive

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            lib.vips_hip_image_unref(h)
            self._h = None

    # ------------------------------------------------------------ creation
    @classmethod
    def new_from_array(cls, array, interpretation="multiband"):
        """Upload a (height, width[, bands]) numpy array."""
        a = np.ascontiguousarray(array)
        if a.ndim == 2:
            a = a[:, :, None]
        if a.ndim != 3:
            raise ValueError("need a 2D or 3D array")
        fmt = DTYPE_FORMATS[a.dtype]
        h, w, b = a.shape
        handle = lib.vips_hip_image_new_from_memory(
            a.ctypes.data, w, h, b, fmt, _enum(INTERPRETATIONS, interpretation, "interpretation")
        )
        return cls(check_handle(handle))

    @classmethod
    def new_from_file(cls, path):
        """Load a libvips native .v file into HBM (iofuncs/vips.c; pinned double-buffered upload)."""
        return cls(check_handle(lib.vips_hip_image_new_from_vfile(os.fsencode(path))))

    def write_to_file(self, path):
        """Save as a libvips native .v file."""
        check(lib.vips_hip_image_write_to_vfile(self._h, os.fsencode(path)))

    @classmethod
    def new_from_tensor(cls, tensor, interpretation="multiband"):
        """Wrap (no copy) a contiguous (H, W, C) torch CUDA tensor."""
        import torch

        if not tensor.is_cuda or not tensor.is_contiguous():
            raise ValueError("need a contiguous CUDA tensor")
        # The library's own streams are non-blocking: work torch has queued on the tensor
        # (fills, copies) must be complete before a kernel on another stream reads it.
        cur = torch.cuda.current_stream(tensor.device)
        if (lib.vips_hip_get_stream() or 0) != cur.cuda_stream:
            cur.synchronize()
        t = tensor if tensor.dim() == 3 else tensor.unsqueeze(-1)
        np_dtype = np.dtype(str(t.dtype).replace("torch.", ""))
        fmt = DTYPE_FORMATS[np_dtype]
        h, w, b = t.shape
        handle = lib.vips_hip_image_new_from_device(
            t.data_ptr(), w, h, b, fmt, _enum(INTERPRETATIONS, interpretation, "interpretation")
        )
        return cls(check_handle(handle), keepalive=tensor)

    @classmethod
    def new_from_device(cls, ptr, width, height, bands, format, interpretation="multiband", keepalive=None):
        handle = lib.vips_hip_image_new_from_device(
            ptr, width, height, bands, _enum(FORMATS, format, "format"),
            _enum(INTERPRETATIONS, interpretation, "interpretation"),
        )
        return cls(check_handle(handle), keepalive=keepalive)

    # ---------------------------------------------------------- properties
    @property
    def width(self):
        return lib.vips_hip_image_get_width(self._h)

    @property
    def height(self):
        return lib.vips_hip_image_get_height(self._h)

    @property
    def bands(self):
        return lib.vips_hip_image_get_bands(self._h)

    @property
    def format(self):
        return FORMAT_NAMES[lib.vips_hip_image_get_format(self._h)]

    @property
    def interpretation(self):
        v = lib.vips_hip_image_get_interpretation(self._h)
        return INTERPRETATION_NAMES.get(v, v)

    @property
    def data_ptr(self):
        return lib.vips_hip_image_get_data(self._h)

    def region(self):
        r = _ffi.Region()
        lib.vips_hip_image_region(self._h, ctypes.byref(r))
        return r

    def numpy(self):
        """Download: vips_image_write_to_memory (iofuncs/image.c:2901)."""
        fmt = lib.vips_hip_image_get_format(self._h)
        out = np.empty((self.height, self.width, self.bands), dtype=FORMAT_DTYPES[fmt])
        check(lib.vips_hip_image_write_to_memory(self._h, out.ctypes.data))
        return out

    # ---------------------------------------------------------- operations
    def _unary(self, fn, *args):
        out = ctypes.c_void_p()
        check(fn(self._h, ctypes.byref(out), *args))
        return Image(out.value)

    def reduceh(self, hshrink, kernel="lanczos3", gap=0.0):
        return self._unary(lib.vips_hip_reduceh, float(hshrink), _enum(KERNELS, kernel, "kernel"), float(gap))

    def reducev(self, vshrink, kernel="lanczos3", gap=0.0):
        return self._unary(lib.vips_hip_reducev, float(vshrink), _enum(KERNELS, kernel, "kernel"), float(gap))

    def reduce(self, hshrink, vshrink, kernel="lanczos3", gap=0.0):
        return self._unary(
            lib.vips_hip_reduce, float(hshrink), float(vshrink), _enum(KERNELS, kernel, "kernel"), float(gap)
        )

    def shrinkh(self, hshrink, ceil=False):
        return self._unary(lib.vips_hip_shrinkh, int(hshrink), int(bool(ceil)))

    def shrinkv(self, vshrink, ceil=False):
        return self._unary(lib.vips_hip_shrinkv, int(vshrink), int(bool(ceil)))

    def shrink(self, hshrink, vshrink, ceil=False):
        return self._unary(lib.vips_hip_shrink, float(hshrink), float(vshrink), int(bool(ceil)))

    def resize(self, scale, vscale=None, kernel="lanczos3", gap=2.0):
        return self._unary(
            lib.vips_hip_resize,
            float(scale),
            float(vscale) if vscale is not None else -1.0,
            _enum(KERNELS, kernel, "kernel"),
            float(gap),
        )

    @classmethod
    def new_from_jpeg(cls, path, shrink=1):
        """jpegload with shrink-on-load 1/2/4/8 (foreign/jpeg2vips.c): host decode, upload."""
        return cls(check_handle(lib.vips_hip_image_new_from_jpeg(os.fsencode(path), int(shrink))))

    @classmethod
    def thumbnail(cls, path, width, height=None, size="both", linear=False, crop="none"):
        """vips_thumbnail() for JPEG and .v files: shrink-on-load, then the thumbnail_image pipeline."""
        out = ctypes.c_void_p()
        check(lib.vips_hip_thumbnail(os.fsencode(path), ctypes.byref(out), int(width), int(height) if height else 0,
                                     _enum(SIZES, size, "size"), int(bool(linear)), _enum(INTERESTING, crop, "crop")))
        return cls(out.value)

    @classmethod
    def thumbnail_batch(cls, paths, width, height=None, size="both", linear=False, crop="none", threads=8):
        """vips_thumbnail() over many files on `threads` host threads (decode and device work of
        different files overlap).  Returns a list of Images; a failed file gives a VipsHipError
        instance in its place."""
        from ._ffi import VipsHipError

        n = len(paths)
        arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
        outs = (ctypes.c_void_p * n)()
        errors = ctypes.create_string_buffer(256 * max(n, 1))
        r = lib.vips_hip_thumbnail_batch(arr, n, outs, errors, int(width), int(height) if height else 0,
                                         _enum(SIZES, size, "size"), int(bool(linear)),
                                         _enum(INTERESTING, crop, "crop"), int(threads))
        if r < 0:
            check(r)
        result = []
        for i in range(n):
            if outs[i]:
                result.append(cls(outs[i]))
            else:
                result.append(VipsHipError(errors.raw[256 * i:256 * i + 256].split(b"\0", 1)[0].decode()))
        return result

    def thumbnail_image(self, width, height=None, size="both", linear=False, crop="none"):
        return self._unary(lib.vips_hip_thumbnail_image_crop, int(width), int(height) if height else 0,
                           _enum(SIZES, size, "size"), int(bool(linear)), _enum(INTERESTING, crop, "crop"))

    def extract_area(self, left, top, width, height):
        return self._unary(lib.vips_hip_extract_area, int(left), int(top), int(width), int(height))

    crop = extract_area

    @staticmethod
    def _mask(mask):
        m = np.ascontiguousarray(np.asarray(mask, dtype=np.float64))
        if m.ndim == 1:
            m = m[None, :]
        return m

    def conv(self, mask, scale=1.0, offset=0.0, precision="float", layers=5, cluster=1):
        m = self._mask(mask)
        if precision == "approximate":  # conv.c:99-107
            return self.conva(m, scale, offset, layers, cluster)
        return self._unary(
            lib.vips_hip_conv,
            m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
            m.shape[1],
            m.shape[0],
            float(scale),
            float(offset),
            _enum(PRECISIONS, precision, "precision"),
        )

    def conva(self, mask, scale=1.0, offset=0.0, layers=5, cluster=1):
        """vips_conva: approximate integer convolution (conva.c:1231-1280)."""
        m = self._mask(mask)
        return self._unary(
            lib.vips_hip_conva,
            m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
            m.shape[1],
            m.shape[0],
            float(scale),
            float(offset),
            int(layers),
            int(cluster),
        )

    def convasep(self, mask, scale=1.0, offset=0.0, layers=5):
        """vips_convasep: approximate separable integer convolution (convasep.c:775-828)."""
        m = self._mask(mask).reshape(-1)
        return self._unary(
            lib.vips_hip_convasep,
            m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
            m.size,
            float(scale),
            float(offset),
            int(layers),
        )

    def convsep(self, mask, scale=1.0, offset=0.0, precision="float", layers=5):
        m = self._mask(mask).reshape(-1)
        if precision == "approximate":  # convsep.c:81-87
            return self.convasep(m, scale, offset, layers)
        return self._unary(
            lib.vips_hip_convsep,
            m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
            m.size,
            float(scale),
            float(offset),
            _enum(PRECISIONS, precision, "precision"),
        )

    def gaussblur(self, sigma, min_ampl=0.2, precision="integer"):
        return self._unary(
            lib.vips_hip_gaussblur, float(sigma), float(min_ampl), _enum(PRECISIONS, precision, "precision")
        )

    def sharpen(self, sigma=0.5, x1=2.0, y2=10.0, y3=20.0, m1=0.0, m2=3.0):
        return self._unary(
            lib.vips_hip_sharpen, float(sigma), float(x1), float(y2), float(y3), float(m1), float(m2)
        )

    def gaussblur_colourspace(self, sigma, space, min_ampl=0.2, precision="integer"):
        """vips_gaussblur() then vips_colourspace(), one kernel where the image allows it
        (vips_hip_gaussblur_colourspace); the same pixels as ``.gaussblur().colourspace()``."""
        return self._unary(lib.vips_hip_gaussblur_colourspace, float(sigma), float(min_ampl),
                           _enum(PRECISIONS, precision, "precision"), _enum(INTERPRETATIONS, space, "space"))

    def colourspace(self, space):
        return self._unary(lib.vips_hip_colourspace, _enum(INTERPRETATIONS, space, "interpretation"))

    def premultiply(self, uchar=False):
        return self._unary(lib.vips_hip_premultiply, int(bool(uchar)))

    def unpremultiply(self, uchar=False):
        return self._unary(lib.vips_hip_unpremultiply, int(bool(uchar)))

    def cast(self, format):
        return self._unary(lib.vips_hip_cast, _enum(FORMATS, format, "format"))


def resize_sharpen_batch(images, scale, kernel="lanczos3", gap=2.0, sharpen=True, sigma=0.5, x1=2.0, y2=10.0,
                         y3=20.0, m1=0.0, m2=3.0, threads=8, wait=True):
    """BASELINE config 4: ``im.resize(scale).sharpen()`` over a batch of images, ``threads`` images
    in flight on their own streams (vips_hip_resize_sharpen_batch; ``wait=False``: the queued
    form, vips_hip_resize_sharpen_batch_queue).  Returns a list of Images."""
    n = len(images)
    ins = (ctypes.c_void_p * n)(*[im._h for im in images])
    outs = (ctypes.c_void_p * n)()
    fn = lib.vips_hip_resize_sharpen_batch if wait else lib.vips_hip_resize_sharpen_batch_queue
    failed = fn(ins, n, outs, float(scale), _enum(KERNELS, kernel, "kernel"), float(gap),
                float(sigma) if sharpen else -1.0, float(x1), float(y2), float(y3), float(m1), float(m2), int(threads))
    result = [Image(h) if h else None for h in outs]
    if failed:
        check(-1)
    return result


def gaussmat(sigma, min_ampl, separable=False, precision="integer"):
    """vips_gaussmat (create/gaussmat.c:95-167): returns (mask 2D array, scale)."""
    buf = (ctypes.c_double * (10001 * 1))()
    scale = ctypes.c_double()
    # first ask for the width with a generous buffer when separable, else size^2
    n = lib.vips_hip_gaussmat(
        float(sigma), float(min_ampl), 1, _enum(PRECISIONS, precision, "precision"), buf, 10001,
        ctypes.byref(scale),
    )
    if n < 0:
        check(-1)
    if separable:
        return np.array(buf[:n], dtype=np.float64)[None, :], scale.value
    big = (ctypes.c_double * (n * n))()
    n2 = lib.vips_hip_gaussmat(
        float(sigma), float(min_ampl), 0, _enum(PRECISIONS, precision, "precision"), big, n * n,
        ctypes.byref(scale),
    )
    if n2 < 0:
        check(-1)
    return np.array(big[: n * n], dtype=np.float64).reshape(n, n), scale.value
