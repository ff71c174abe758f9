"""Test helpers: synthetic inputs, the reference shim, the oracle port.

TEST INFRASTRUCTURE.  ``Ref`` drives the *compiled reference* (oracle/_ref, built
by oracle/build_ref.sh from /root/reference; the .so travels to the GPU box).
``Port`` drives the plain-C restatement (oracle/port).  Neither is ever imported
by the product package.
"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "lib", "libref_shim.so")
PORT_LIB = os.path.join(ROOT, "oracle", "_build", "liboracle_port.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")
MODULE_LIB = os.path.join(ROOT, "host", "_build", "vips-hip.so")

FORMAT_DTYPES = {
    0: np.uint8, 1: np.int8, 2: np.uint16, 3: np.int16, 4: np.uint32,
    5: np.int32, 6: np.float32, 7: np.complex64, 8: np.float64, 9: np.complex128,
}
DTYPE_FORMATS = {np.dtype(v): k for k, v in FORMAT_DTYPES.items()}


def lcg_bytes(n, seed=12345):
    """s = s*1664525 + 1013904223 (mod 2^32); byte = s >> 24."""
    out = np.empty(n, dtype=np.uint8)
    B = 1 << 16
    M = np.uint64(0xFFFFFFFF)
    a = np.uint64(1664525)
    c = np.uint64(1013904223)
    ak = np.empty(B, dtype=np.uint64)
    ck = np.empty(B, dtype=np.uint64)
    aa = np.uint64(1)
    cc = np.uint64(0)
    for k in range(B):
        aa = (aa * a) & M
        cc = (cc * a + c) & M
        ak[k] = aa
        ck[k] = cc
    s = np.uint64(seed)
    for i in range(0, n, B):
        m = min(B, n - i)
        v = (ak[:m] * s + ck[:m]) & M
        out[i:i + m] = (v >> np.uint64(24)).astype(np.uint8)
        s = v[m - 1]
    return out


def lcg_image(width, height, bands, dtype=np.uint8, seed=12345):
    """Deterministic synthetic image of any band format (full range for ints)."""
    dtype = np.dtype(dtype)
    n = width * height * bands
    if dtype == np.uint8:
        return lcg_bytes(n, seed).reshape(height, width, bands)
    if dtype == np.int8:
        return lcg_bytes(n, seed).view(np.int8).reshape(height, width, bands)
    if dtype.kind in "ui":
        raw = lcg_bytes(n * dtype.itemsize, seed)
        return raw.view(dtype).reshape(height, width, bands)
    if dtype.kind == "f":
        # byte values 0..255 plus a deterministic fraction
        raw = lcg_bytes(n * 2, seed).astype(np.float64)
        v = raw[0::2] + raw[1::2] / 256.0
        return v.astype(dtype).reshape(height, width, bands)
    raise ValueError(dtype)


def checksum(array):
    """SURVEY.md 8(c) golden checksum: sum o[i] * (i mod 251 + 1)."""
    o = np.ascontiguousarray(array).reshape(-1).astype(np.int64)
    idx = (np.arange(o.size, dtype=np.int64) % 251) + 1
    return int((o * idx).sum())


class RefImage(ctypes.Structure):
    _fields_ = [
        ("data", ctypes.c_void_p), ("width", ctypes.c_int), ("height", ctypes.c_int),
        ("bands", ctypes.c_int), ("format", ctypes.c_int), ("interpretation", ctypes.c_int),
    ]


def have_ref():
    return os.path.exists(REF_LIB)


def have_module():
    return os.path.exists(REF_LIB) and os.path.exists(MODULE_LIB)


class Ref(object):
    """The compiled reference (libvips 8.19.0 scalar C paths) through oracle/ref_shim.c."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            lib = ctypes.CDLL(REF_LIB)
            lib.ref_init.argtypes = [ctypes.c_int]
            lib.ref_error.restype = ctypes.c_char_p
            lib.ref_version.restype = ctypes.c_char_p
            lib.ref_free.argtypes = [ctypes.c_void_p]
            lib.ref_run.argtypes = [ctypes.c_char_p, ctypes.POINTER(RefImage), ctypes.c_char_p,
                                    ctypes.POINTER(RefImage)]
            lib.ref_run_mask.argtypes = [ctypes.c_char_p, ctypes.POINTER(RefImage),
                                         ctypes.POINTER(RefImage), ctypes.c_double, ctypes.c_double,
                                         ctypes.c_char_p, ctypes.POINTER(RefImage)]
            lib.ref_create.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(RefImage),
                                       ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
            lib.ref_run_chain.argtypes = [ctypes.c_char_p, ctypes.POINTER(RefImage),
                                          ctypes.POINTER(RefImage)]
            lib.ref_time_chain.argtypes = [ctypes.c_char_p, ctypes.POINTER(RefImage), ctypes.c_int]
            lib.ref_time_chain.restype = ctypes.c_double
            for name in ("ref_col_Lab2XYZ", "ref_col_XYZ2Lab"):
                getattr(lib, name).argtypes = [ctypes.c_float] * 3 + [ctypes.POINTER(ctypes.c_float)] * 3
            if lib.ref_init(0) != 0:
                raise RuntimeError("reference failed to initialise")
            cls._lib = lib
        return cls._lib

    @staticmethod
    def _wrap(array, interpretation=0):
        a = np.ascontiguousarray(array)
        if a.ndim == 2:
            a = a[:, :, None]
        h, w, b = a.shape
        return a, RefImage(a.ctypes.data, w, h, b, DTYPE_FORMATS[a.dtype], interpretation)

    @classmethod
    def _take(cls, out):
        dtype = np.dtype(FORMAT_DTYPES[out.format])
        n = out.width * out.height * out.bands
        buf = (ctypes.c_char * (n * dtype.itemsize)).from_address(out.data)
        a = np.frombuffer(buf, dtype=dtype).reshape(out.height, out.width, out.bands).copy()
        cls.lib().ref_free(out.data)
        return a

    @classmethod
    def _fail(cls, what):
        lib = cls.lib()
        msg = lib.ref_error().decode()
        lib.ref_error_clear()
        raise RuntimeError("%s: %s" % (what, msg))

    @classmethod
    def run(cls, nick, array, args="", interpretation=0):
        a, ri = cls._wrap(array, interpretation)
        ro = RefImage()
        if cls.lib().ref_run(nick.encode(), ctypes.byref(ri), args.encode(), ctypes.byref(ro)) != 0:
            cls._fail(nick)
        return cls._take(ro)

    @classmethod
    def run_interp(cls, nick, array, args="", interpretation=0):
        """As run(), also returning the output's interpretation."""
        a, ri = cls._wrap(array, interpretation)
        ro = RefImage()
        if cls.lib().ref_run(nick.encode(), ctypes.byref(ri), args.encode(), ctypes.byref(ro)) != 0:
            cls._fail(nick)
        interp = ro.interpretation
        return cls._take(ro), interp

    @classmethod
    def run_mask(cls, nick, array, mask, scale=1.0, offset=0.0, args="", interpretation=0):
        a, ri = cls._wrap(array, interpretation)
        m, rm = cls._wrap(np.asarray(mask, dtype=np.float64))
        ro = RefImage()
        if cls.lib().ref_run_mask(nick.encode(), ctypes.byref(ri), ctypes.byref(rm), scale, offset,
                                  args.encode(), ctypes.byref(ro)) != 0:
            cls._fail(nick)
        return cls._take(ro)

    @classmethod
    def create(cls, nick, args=""):
        ro = RefImage()
        scale = ctypes.c_double()
        offset = ctypes.c_double()
        if cls.lib().ref_create(nick.encode(), args.encode(), ctypes.byref(ro), ctypes.byref(scale),
                                ctypes.byref(offset)) != 0:
            cls._fail(nick)
        return cls._take(ro), scale.value, offset.value

    @classmethod
    def run_chain(cls, chain, array, interpretation=0):
        a, ri = cls._wrap(array, interpretation)
        ro = RefImage()
        if cls.lib().ref_run_chain(chain.encode(), ctypes.byref(ri), ctypes.byref(ro)) != 0:
            cls._fail(chain)
        return cls._take(ro)

    @classmethod
    def time_chain(cls, chain, array, repeats=3, interpretation=0, concurrency=0):
        lib = cls.lib()
        if concurrency:
            lib.ref_init(concurrency)
        a, ri = cls._wrap(array, interpretation)
        t = lib.ref_time_chain(chain.encode(), ctypes.byref(ri), repeats)
        if t < 0:
            cls._fail(chain)
        return t

    @classmethod
    def concurrency(cls):
        return cls.lib().ref_concurrency()

    @classmethod
    def build_probe(cls, nick, width, height, bands=4, args="", interpretation=0):
        """Build (never evaluate) @nick on a black uchar image: ((Xsize, Ysize, Bands, BandFmt,
        Type), seconds)."""
        lib = cls.lib()
        lib.ref_build_probe.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]
        header = (ctypes.c_int * 5)()
        seconds = ctypes.c_double()
        if lib.ref_build_probe(nick.encode(), width, height, bands, interpretation, args.encode(), header,
                               ctypes.byref(seconds)) != 0:
            cls._fail(nick)
        return tuple(header), seconds.value

    _module_loaded = False

    @classmethod
    def load_module(cls):
        """Open host/_build/vips-hip.so in the reference libvips (registers the *_hip ops)."""
        if cls._module_loaded:
            return
        lib = cls.lib()
        lib.ref_load_module.argtypes = [ctypes.c_char_p]
        if lib.ref_load_module(MODULE_LIB.encode()) != 0:
            cls._fail("load_module")
        cls._module_loaded = True


def have_port():
    return os.path.exists(PORT_LIB)


KERNELS = {"nearest": 0, "linear": 1, "cubic": 2, "mitchell": 3, "lanczos2": 4, "lanczos3": 5,
           "mks2013": 6, "mks2021": 7}


class Port(object):
    """oracle/port: the plain-C restatement of the hot path (CPU, single thread)."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            lib = ctypes.CDLL(PORT_LIB)
            vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
            lib.port_reduce_get_points.argtypes = [ci, cd]
            lib.port_reduce_make_mask.argtypes = [ctypes.POINTER(cd), ci, ci, cd, cd]
            lib.port_reduce_make_mask.restype = None
            lib.port_reduceh.argtypes = [vp, ci, ci, ci, ci, cd, ci, ci, cd, vp]
            lib.port_reducev.argtypes = [vp, ci, ci, ci, ci, cd, ci, ci, cd, ci, vp]
            lib.port_shrink_out_size.argtypes = [ci, ci, ci]
            lib.port_shrinkh.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp]
            lib.port_shrinkv.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp]
            cls._lib = lib
        return cls._lib

    @staticmethod
    def _prep(array):
        a = np.ascontiguousarray(array)
        if a.ndim == 2:
            a = a[:, :, None]
        return a

    @classmethod
    def shrinkh(cls, array, hshrink, ceil=False):
        a = cls._prep(array)
        h, w, b = a.shape
        ow = cls.lib().port_shrink_out_size(w, hshrink, int(ceil))
        out = np.empty((h, ow, b), dtype=a.dtype)
        cls.lib().port_shrinkh(a.ctypes.data, w, h, b, DTYPE_FORMATS[a.dtype], hshrink, int(ceil),
                               out.ctypes.data)
        return out

    @classmethod
    def shrinkv(cls, array, vshrink, ceil=False):
        a = cls._prep(array)
        h, w, b = a.shape
        oh = cls.lib().port_shrink_out_size(h, vshrink, int(ceil))
        out = np.empty((oh, w, b), dtype=a.dtype)
        cls.lib().port_shrinkv(a.ctypes.data, w, h, b, DTYPE_FORMATS[a.dtype], vshrink, int(ceil),
                               out.ctypes.data)
        return out

    @classmethod
    def _reduce_axis(cls, array, shrink, kernel, gap, vertical, tile):
        """vips_reduceh_build / vips_reducev_build (reduceh.cpp:396-481, reducev.cpp:859-941)."""
        a = cls._prep(array)
        k = KERNELS[kernel] if isinstance(kernel, str) else kernel
        in_size = a.shape[0] if vertical else a.shape[1]
        size = int(in_size / shrink + 0.5)
        extra = size * shrink - in_size
        residual = shrink
        if gap > 0.0 and k != 0:
            int_shrink = max(1, int(np.floor(in_size / size / gap)))
            if int_shrink > 1:
                a = cls.shrinkv(a, int_shrink, True) if vertical else cls.shrinkh(a, int_shrink, True)
                residual /= int_shrink
                extra /= int_shrink
        if residual == 1.0:
            return a.copy()
        h, w, b = a.shape
        fmt = DTYPE_FORMATS[a.dtype]
        if vertical:
            out = np.empty((size, w, b), dtype=a.dtype)
            r = cls.lib().port_reducev(a.ctypes.data, w, h, b, fmt, residual, k, size, extra, tile,
                                       out.ctypes.data)
        else:
            out = np.empty((h, size, b), dtype=a.dtype)
            r = cls.lib().port_reduceh(a.ctypes.data, w, h, b, fmt, residual, k, size, extra,
                                       out.ctypes.data)
        if r != 0:
            raise RuntimeError("port reduce failed")
        return out

    @classmethod
    def reducev(cls, array, vshrink, kernel="lanczos3", gap=0.0, tile=16):
        return cls._reduce_axis(array, vshrink, kernel, gap, True, tile)

    @classmethod
    def reduceh(cls, array, hshrink, kernel="lanczos3", gap=0.0):
        return cls._reduce_axis(array, hshrink, kernel, gap, False, 0)

    @classmethod
    def reduce(cls, array, hshrink, vshrink, kernel="lanczos3", gap=0.0, tile=16):
        """vips_reduce_build (reduce.c:98-121): vertical first."""
        return cls.reduceh(cls.reducev(array, vshrink, kernel, gap, tile), hshrink, kernel, gap)

    @classmethod
    def shrink(cls, array, hshrink, vshrink, ceil=False):
        """vips_shrink_build (shrink.c:77-119)."""
        if int(hshrink) != hshrink or int(vshrink) != vshrink:
            return cls.reduceh(cls.reducev(array, vshrink, "lanczos3", 1.0), hshrink, "lanczos3", 1.0)
        return cls.shrinkh(cls.shrinkv(array, int(vshrink), ceil), int(hshrink), ceil)

    @classmethod
    def affine_scale(cls, array, hscale, vscale, idx=0.0, idy=0.0, interpolate="bicubic", tile_width=0):
        """vips_affine for a pure scale (affine.c:230-620) with nearest / bilinear / bicubic."""
        a = cls._prep(array)
        lib = cls.lib()
        lib.port_affine_out_size.argtypes = [ctypes.c_int, ctypes.c_double]
        lib.port_affine_scale.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        h, w, b = a.shape
        ow = lib.port_affine_out_size(w, hscale)
        oh = lib.port_affine_out_size(h, vscale)
        out = np.empty((oh, ow, b), dtype=a.dtype)
        interp = {"nearest": 0, "bilinear": 1, "bicubic": 2}[interpolate]
        r = lib.port_affine_scale(a.ctypes.data, w, h, b, DTYPE_FORMATS[a.dtype], hscale, vscale, idx, idy,
                                  interp, tile_width, out.ctypes.data)
        if r != 0:
            raise ValueError("port_affine_scale: unsupported format")
        return out

    @classmethod
    def zoom(cls, array, xfac, yfac):
        """vips_zoom (conversion/zoom.c)."""
        a = cls._prep(array)
        return np.ascontiguousarray(np.repeat(np.repeat(a, yfac, axis=0), xfac, axis=1))

    @classmethod
    def resize(cls, array, scale, vscale=None, kernel="lanczos3", gap=2.0, tile=16):
        """vips_resize_build (resize.c:135-329): residual reduce, then any upsizing through
        vips_affine / vips_zoom; kernel nearest shrinks by vips_subsample first."""
        import math

        a = cls._prep(array)
        hscale = scale
        vscale = scale if vscale is None else vscale
        if kernel == "nearest":
            # the int part by vips_subsample (resize.c:165-203; conversion/subsample.c)
            h, w = a.shape[:2]
            if gap < 1.0:
                ih, iv = math.floor(1.0 / hscale), math.floor(1.0 / vscale)
            else:
                tw, th = int(w * hscale + 0.5), int(h * vscale + 0.5)
                if tw == 0 or th == 0:
                    raise ValueError("unsupported: nearest target rounds to zero pixels")
                ih, iv = math.floor(w / tw / gap), math.floor(h / th / gap)
            ih, iv = max(1, int(ih)), max(1, int(iv))
            if ih > 1 or iv > 1:
                a = np.ascontiguousarray(a[::iv, ::ih][:h // iv, :w // ih])
                hscale *= ih
                vscale *= iv
        hscale = max(hscale, 1.0 / a.shape[1])
        vscale = max(vscale, 1.0 / a.shape[0])
        if vscale < 1.0:
            a = cls.reducev(a, 1.0 / vscale, kernel, gap, tile)
        if hscale < 1.0:
            a = cls.reduceh(a, 1.0 / hscale, kernel, gap)
        if hscale > 1.0 or vscale > 1.0:
            interpolate = {"nearest": "nearest", "linear": "bilinear"}.get(kernel, "bicubic")
            idx = 0.0 if kernel == "nearest" else 0.5 * (1.0 - 1.0 / hscale)
            idy = 0.0 if kernel == "nearest" else 0.5 * (1.0 - 1.0 / vscale)
            if kernel == "nearest" and hscale == math.floor(hscale) and vscale == math.floor(vscale):
                a = cls.zoom(a, int(math.floor(hscale)), int(math.floor(vscale)))
            elif hscale > 1.0 and vscale > 1.0:
                a = cls.affine_scale(a, hscale, vscale, idx, idy, interpolate)
            elif hscale > 1.0:
                a = cls.affine_scale(a, hscale, 1.0, idx, idy, interpolate)
            else:
                a = cls.affine_scale(a, 1.0, vscale, idx, idy, interpolate)
        return a


# ------------------------------------------------------------------ conv / colour port

INTERP = {"multiband": 0, "b-w": 1, "xyz": 12, "lab": 13, "labs": 21, "srgb": 22, "rgb16": 25,
          "grey16": 26, "scrgb": 28}


def _port_conv_setup(lib):
    if getattr(lib, "_conv_ready", False):
        return
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    pd = ctypes.POINTER(cd)
    for name in ("port_convi", "port_convf"):
        getattr(lib, name).argtypes = [vp, ci, ci, ci, ci, pd, ci, ci, cd, cd, vp]
    lib.port_gaussmat.argtypes = [cd, cd, ci, ci, pd, pd]
    lib.port_sharpen_lut.argtypes = [cd, cd, cd, cd, cd, vp]
    lib.port_sharpen_lut.restype = None
    lib.port_sharpen_apply.argtypes = [vp, vp, ci, ci, vp, vp]
    lib.port_sharpen_apply.restype = None
    for name in ("port_sRGB2scRGB_8", "port_sRGB2scRGB_16", "port_scRGB2XYZ", "port_XYZ2Lab",
                 "port_Lab2XYZ", "port_XYZ2scRGB", "port_scRGB2sRGB_8", "port_scRGB2sRGB_16",
                 "port_Lab2LabS", "port_LabS2Lab"):
        getattr(lib, name).argtypes = [vp, ci, vp]
        getattr(lib, name).restype = None
    pi = ctypes.POINTER(ci)
    lib.port_conva.argtypes = [vp, ci, ci, ci, ci, pd, ci, ci, cd, cd, ci, ci, vp]
    lib.port_convasep.argtypes = [vp, ci, ci, ci, ci, pd, ci, cd, cd, ci, vp]
    lib.port_conva_decompose.argtypes = [pd, ci, ci, cd, cd, ci, ci, pi, pi, ci]
    lib.port_convasep_decompose.argtypes = [pd, ci, cd, cd, ci, pi, pi, ci]
    lib.port_convi_hwy.argtypes = [vp, ci, ci, ci, pd, ci, ci, cd, cd, vp]
    lib.port_convi_hwy_intize.argtypes = [pd, ci, cd, ctypes.POINTER(ctypes.c_short), pi, pi, pi]
    lib.port_cast.argtypes = [vp, ctypes.c_size_t, ci, ci, vp]
    lib.port_premultiply.argtypes = [vp, ctypes.c_size_t, ci, ci, cd, ci, ci, vp]
    lib._conv_ready = True


class PortCC(object):
    """oracle/port conv + colour, composed the way the reference composes images."""

    @staticmethod
    def lib():
        lib = Port.lib()
        _port_conv_setup(lib)
        return lib

    @staticmethod
    def _mask(mask):
        m = np.ascontiguousarray(np.asarray(mask, dtype=np.float64))
        if m.ndim == 1:
            m = m[None, :]
        return m

    @classmethod
    def conv(cls, array, mask, scale=1.0, offset=0.0, precision="float", layers=5, cluster=1):
        """vips_conv (conv.c:62-118)."""
        if precision == "approximate":
            return cls.conva(array, mask, scale, offset, layers, cluster)
        a = Port._prep(array)
        m = cls._mask(mask)
        h, w, b = a.shape
        fmt = DTYPE_FORMATS[a.dtype]
        pm = m.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        if precision == "integer":
            out = np.empty_like(a)
            r = cls.lib().port_convi(a.ctypes.data, w, h, b, fmt, pm, m.shape[1], m.shape[0], scale, offset,
                                     out.ctypes.data)
        else:
            out = np.empty(a.shape, dtype=np.float64 if a.dtype == np.float64 else np.float32)
            r = cls.lib().port_convf(a.ctypes.data, w, h, b, fmt, pm, m.shape[1], m.shape[0], scale, offset,
                                     out.ctypes.data)
        if r != 0:
            raise RuntimeError("port conv failed")
        return out

    @classmethod
    def convsep(cls, array, mask, scale=1.0, offset=0.0, precision="float", layers=5):
        """vips_convsep (convsep.c:61-118): conv(M) then conv(rot90(M), offset 0)."""
        if precision == "approximate":
            return cls.convasep(array, mask, scale, offset, layers)
        m = cls._mask(mask).reshape(1, -1)
        t = cls.conv(array, m, scale, offset, precision)
        return cls.conv(t, m.reshape(-1, 1), scale, 0.0, precision)

    @classmethod
    def convi_vector(cls, array, mask, scale=1.0, offset=0.0):
        """convi as a Highway-built libvips computes it on uchar (convi.c:925-1120,
        convi_hwy.cpp:264-273; PARITY UNPINNED: no Highway here).  Falls back to the C path
        when the intize refuses the mask, as convi.c:1150-1170 does."""
        a = Port._prep(array)
        m = cls._mask(mask)
        if a.dtype != np.uint8:
            return cls.conv(a, m, scale, offset, "integer")
        h, w, b = a.shape
        out = np.empty_like(a)
        r = cls.lib().port_convi_hwy(a.ctypes.data, w, h, b, m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                     m.shape[1], m.shape[0], scale, offset, out.ctypes.data)
        if r == 1:
            return cls.conv(a, m, scale, offset, "integer")
        return out

    @classmethod
    def convi_vector_intize(cls, mask, scale=1.0):
        """(exp, [(mant, pos)...]) or None when refused."""
        m = cls._mask(mask)
        n = m.size
        mant = (ctypes.c_short * n)()
        pos = (ctypes.c_int * n)()
        nnz, exp = ctypes.c_int(), ctypes.c_int()
        if cls.lib().port_convi_hwy_intize(m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n, scale, mant, pos,
                                           ctypes.byref(nnz), ctypes.byref(exp)) != 0:
            return None
        return exp.value, [(mant[i], pos[i]) for i in range(nnz.value)]

    @classmethod
    def conva(cls, array, mask, scale=1.0, offset=0.0, layers=5, cluster=1):
        """vips_conva (conva.c:1231-1280)."""
        a = Port._prep(array)
        m = cls._mask(mask)
        h, w, b = a.shape
        out = np.empty_like(a)
        r = cls.lib().port_conva(a.ctypes.data, w, h, b, DTYPE_FORMATS[a.dtype],
                                 m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), m.shape[1], m.shape[0],
                                 scale, offset, layers, cluster, out.ctypes.data)
        if r != 0:
            raise RuntimeError("port conva failed")
        return out

    @classmethod
    def convasep(cls, array, mask, scale=1.0, offset=0.0, layers=5):
        """vips_convasep (convasep.c:775-828)."""
        a = Port._prep(array)
        m = cls._mask(mask).reshape(-1)
        h, w, b = a.shape
        out = np.empty_like(a)
        r = cls.lib().port_convasep(a.ctypes.data, w, h, b, DTYPE_FORMATS[a.dtype],
                                    m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), m.size,
                                    scale, offset, layers, out.ctypes.data)
        if r != 0:
            raise RuntimeError("port convasep failed")
        return out

    @classmethod
    def conva_decompose(cls, mask, scale=1.0, offset=0.0, layers=5, cluster=1):
        """(info, hlines, vlines) of the box decomposition (conva.c:676-767)."""
        m = cls._mask(mask)
        info = (ctypes.c_int * 6)()
        lines = (ctypes.c_int * 6000)()
        n = cls.lib().port_conva_decompose(m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), m.shape[1],
                                           m.shape[0], scale, offset, layers, cluster, info, lines, 6000)
        if n < 0:
            raise RuntimeError("port conva decompose failed")
        nh, nv = info[0], info[1]
        flat = list(lines[:n])
        hl = [tuple(flat[2 * i:2 * i + 2]) for i in range(nh)]
        vl = [tuple(flat[2 * nh + 4 * i:2 * nh + 4 * i + 4]) for i in range(nv)]
        return list(info), hl, vl

    @classmethod
    def convasep_decompose(cls, mask, scale=1.0, offset=0.0, layers=5):
        """(info, lines) of the line decomposition (convasep.c:152-330)."""
        m = cls._mask(mask).reshape(-1)
        info = (ctypes.c_int * 4)()
        lines = (ctypes.c_int * 3000)()
        n = cls.lib().port_convasep_decompose(m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), m.size,
                                              scale, offset, layers, info, lines, 3000)
        if n < 0:
            raise RuntimeError("port convasep decompose failed")
        flat = list(lines[:n])
        return list(info), [tuple(flat[3 * i:3 * i + 3]) for i in range(info[0])]

    @classmethod
    def gaussmat(cls, sigma, min_ampl, separable=False, precision="integer"):
        integer = 0 if precision == "float" else 1
        scale = ctypes.c_double()
        n = cls.lib().port_gaussmat(sigma, min_ampl, int(separable), integer, None, ctypes.byref(scale))
        m = np.empty((1 if separable else n, n), dtype=np.float64)
        cls.lib().port_gaussmat(sigma, min_ampl, int(separable), integer,
                                m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.byref(scale))
        return m, scale.value

    @classmethod
    def gaussblur(cls, array, sigma, min_ampl=0.2, precision="integer"):
        """vips_gaussblur (gaussblur.c:71-116)."""
        if sigma < 0.2:
            return Port._prep(array).copy()
        m, scale = cls.gaussmat(sigma, min_ampl, True, precision)
        return cls.convsep(array, m, scale, 0.0, precision)

    @classmethod
    def cast(cls, array, dtype):
        a = Port._prep(array)
        out = np.empty(a.shape, dtype=dtype)
        if cls.lib().port_cast(a.ctypes.data, a.size, DTYPE_FORMATS[a.dtype], DTYPE_FORMATS[np.dtype(dtype)],
                               out.ctypes.data) != 0:
            raise RuntimeError("port cast failed")
        return out

    # one colour step on the first 3 bands; extra bands carried like colour.c:249-296
    _STEPS = {
        "sRGB2scRGB": ("port_sRGB2scRGB_8", np.uint8, np.float32, "scrgb"),
        "sRGB2scRGB16": ("port_sRGB2scRGB_16", np.uint16, np.float32, "scrgb"),
        "scRGB2XYZ": ("port_scRGB2XYZ", np.float32, np.float32, "xyz"),
        "XYZ2Lab": ("port_XYZ2Lab", np.float32, np.float32, "lab"),
        "Lab2XYZ": ("port_Lab2XYZ", np.float32, np.float32, "xyz"),
        "XYZ2scRGB": ("port_XYZ2scRGB", np.float32, np.float32, "scrgb"),
        "scRGB2sRGB": ("port_scRGB2sRGB_8", np.float32, np.uint8, "srgb"),
        "Lab2LabS": ("port_Lab2LabS", np.float32, np.int16, "labs"),
        "LabS2Lab": ("port_LabS2Lab", np.int16, np.float32, "lab"),
    }
    _ROUTES = {
        ("xyz", "lab"): ["XYZ2Lab"], ("xyz", "labs"): ["XYZ2Lab", "Lab2LabS"],
        ("xyz", "scrgb"): ["XYZ2scRGB"], ("xyz", "srgb"): ["XYZ2scRGB", "scRGB2sRGB"],
        ("lab", "xyz"): ["Lab2XYZ"], ("lab", "labs"): ["Lab2LabS"],
        ("lab", "scrgb"): ["Lab2XYZ", "XYZ2scRGB"], ("lab", "srgb"): ["Lab2XYZ", "XYZ2scRGB", "scRGB2sRGB"],
        ("labs", "xyz"): ["LabS2Lab", "Lab2XYZ"], ("labs", "lab"): ["LabS2Lab"],
        ("labs", "scrgb"): ["LabS2Lab", "Lab2XYZ", "XYZ2scRGB"],
        ("labs", "srgb"): ["LabS2Lab", "Lab2XYZ", "XYZ2scRGB", "scRGB2sRGB"],
        ("scrgb", "xyz"): ["scRGB2XYZ"], ("scrgb", "lab"): ["scRGB2XYZ", "XYZ2Lab"],
        ("scrgb", "labs"): ["scRGB2XYZ", "XYZ2Lab", "Lab2LabS"], ("scrgb", "srgb"): ["scRGB2sRGB"],
        ("srgb", "xyz"): ["sRGB2scRGB", "scRGB2XYZ"], ("srgb", "lab"): ["sRGB2scRGB", "scRGB2XYZ", "XYZ2Lab"],
        ("srgb", "labs"): ["sRGB2scRGB", "scRGB2XYZ", "XYZ2Lab", "Lab2LabS"],
        ("srgb", "scrgb"): ["sRGB2scRGB"],
    }
    _IDENTITY = {"xyz": np.float32, "lab": np.float32, "scrgb": np.float32, "srgb": np.uint8,
                 "labs": np.int16}
    _MAX_ALPHA = {"rgb16": 65535.0, "grey16": 65535.0, "scrgb": 1.0}

    @classmethod
    def colour_step(cls, array, step, interp_in):
        fn, tin, tout, interp_out = cls._STEPS[step]
        a = Port._prep(array)
        main = np.ascontiguousarray(cls.cast(a[:, :, :3], tin))  # code/transform build casts
        h, w, _ = main.shape
        out3 = np.empty((h, w, 3), dtype=tout)
        getattr(cls.lib(), fn)(main.ctypes.data, h * w, out3.ctypes.data)
        if a.shape[2] == 3:
            return out3, interp_out
        extra = a[:, :, 3:]
        before = cls._MAX_ALPHA.get(interp_in, 255.0)
        after = cls._MAX_ALPHA.get(interp_out, 255.0)
        if before != after:
            # vips_linear1, LOOP1 (arithmetic/linear.c:213-223): float a1 * (float) p + b1
            extra = np.float32(after / before) * extra.astype(np.float32) + np.float32(0.0)
        extra = cls.cast(np.ascontiguousarray(extra), tout)
        return np.concatenate([out3, extra], axis=2), interp_out

    @classmethod
    def colourspace(cls, array, space, interpretation):
        """vips_colourspace (colourspace.c:551-612) between the spaces of this library."""
        a = Port._prep(array)
        if interpretation == space:
            return cls.cast(a, cls._IDENTITY[space])
        interp = interpretation
        for step in cls._ROUTES[(interpretation, space)]:
            a, interp = cls.colour_step(a, step, interp)
        return a

    @classmethod
    def premultiply(cls, array, interpretation="srgb", uchar=False, inverse=False):
        """vips_premultiply / vips_unpremultiply, alpha = last band, default max_alpha."""
        a = Port._prep(array)
        h, w, b = a.shape
        fast = uchar and a.dtype == np.uint8
        out = np.empty(a.shape, dtype=np.uint8 if fast else np.float32)
        r = cls.lib().port_premultiply(a.ctypes.data, h * w, b, DTYPE_FORMATS[a.dtype],
                                       cls._MAX_ALPHA.get(interpretation, 255.0), int(uchar), int(inverse),
                                       out.ctypes.data)
        if r != 0:
            raise RuntimeError("port premultiply failed")
        return out

    @classmethod
    def thumbnail_image(cls, array, interpretation, width, height=None, size="both", linear=False, crop="none"):
        """vips_thumbnail_image (thumbnail.c:678-1067, shrink :413-467), positional crops, no ICC."""
        space = "scrgb" if linear else "srgb"
        if interpretation == "b-w" and not linear and np.asarray(array).shape[2] == 1:
            a = Port._prep(array)  # B_W is the processing space of one-band images (thumbnail.c:806-820)
        else:
            a = cls.colourspace(array, space, interpretation)
        h, w, _ = a.shape
        height = height or width
        hshrink, vshrink = w / width, h / height
        if size != "force":
            horizontal = (hshrink < vshrink) if crop != "none" else not (hshrink < vshrink)
            if horizontal:
                vshrink = hshrink
            else:
                hshrink = vshrink
        if size == "up":
            hshrink, vshrink = min(1, hshrink), min(1, vshrink)
        elif size == "down":
            hshrink, vshrink = max(1, hshrink), max(1, vshrink)
        hshrink, vshrink = min(hshrink, w), min(vshrink, h)
        fmt = None
        if a.shape[2] > 3 and hshrink != 1.0 and vshrink != 1.0:  # vips_image_hasalpha
            fmt = a.dtype
            a = cls.premultiply(a, space, uchar=(a.dtype == np.uint8))
        out = Port.resize(a, 1.0 / hshrink, 1.0 / vshrink)
        if fmt is not None:
            if fmt == np.uint8:
                out = cls.premultiply(out, space, uchar=True, inverse=True)
            else:
                out = cls.cast(cls.premultiply(out, space, inverse=True), fmt)
        if linear:
            out = cls.colourspace(out, "srgb", "scrgb")
        if crop != "none":  # thumbnail.c:1010-1038 -> smartcrop.c:359-400
            oh, ow, _ = out.shape
            cw, ch = min(width, ow), min(height, oh)
            left, top = {"centre": ((ow - cw) // 2, (oh - ch) // 2), "low": (0, 0),
                         "high": (ow - cw, oh - ch), "all": (0, 0)}[crop]
            if crop == "all":
                cw, ch = ow, oh
            out = np.ascontiguousarray(out[top:top + ch, left:left + cw])
        return out

    @classmethod
    def sharpen(cls, array, interpretation="srgb", sigma=0.5, x1=2.0, y2=10.0, y3=20.0, m1=0.0, m2=3.0):
        """vips_sharpen (sharpen.c:171-302)."""
        labs = cls.colourspace(array, "labs", interpretation)
        labs = np.ascontiguousarray(cls.cast(labs, np.int16))
        m, scale = cls.gaussmat(sigma, 0.1, True, "integer")
        lut = np.empty(65536, dtype=np.int32)
        cls.lib().port_sharpen_lut(x1, y2, y3, m1, m2, lut.ctypes.data)
        L = np.ascontiguousarray(labs[:, :, :1])
        blur = np.ascontiguousarray(cls.convsep(L, m, scale, 0.0, "integer"))
        out = np.empty_like(labs)
        h, w, b = labs.shape
        cls.lib().port_sharpen_apply(labs.ctypes.data, blur.ctypes.data, h * w, b, lut.ctypes.data,
                                     out.ctypes.data)
        return cls.colourspace(out, interpretation, "labs")


# ------------------------------------------------------------------ the .v native format

def write_v(path, array, interpretation=22):
    """Write a libvips native .v file (doc/file-format.md:42-57: 64-byte header + pixels)."""
    import struct

    a = np.ascontiguousarray(array)
    h, w, b = a.shape
    magic = bytes([0xB6, 0xA6, 0xF2, 0x08])  # VIPS_MAGIC_INTEL, always written MSB first (iofuncs/vips.c:427)
    # Bbits (offset 16) is deprecated but still written: sizeof(element) * 8 (iofuncs/vips.c:352)
    header = magic + struct.pack("<iiiiiiiffiiii", w, h, b, a.dtype.itemsize * 8, DTYPE_FORMATS[a.dtype], 0,
                                 interpretation, 1.0, 1.0, 0, 0, 0, 0)
    header = header.ljust(64, b"\0")
    with open(path, "wb") as f:
        f.write(header)
        f.write(a.tobytes())


def read_v(path):
    import struct

    raw = open(path, "rb").read()
    w, h, b, _, fmt, coding, interp = struct.unpack("<iiiiiii", raw[4:32])
    dtype = np.dtype(FORMAT_DTYPES[fmt])
    n = w * h * b * dtype.itemsize
    return np.frombuffer(raw[64:64 + n], dtype=dtype).reshape(h, w, b).copy(), interp


class Background(object):
    """Child processes of the CPU suite that take minutes (a `-m gpu` file on host fibers, the module's strip
    producer under the mock runtime): started early -- tests/conftest.py starts the selected ones when collection
    ends -- and waited for by the test that asserts on them, so that they run beside the rest of the suite
    instead of after it.  Test infrastructure only."""

    jobs = {}

    @classmethod
    def start(cls, name, cmd, env=None, cwd=None):
        import subprocess
        import tempfile

        if name in cls.jobs:
            return
        out = tempfile.TemporaryFile(mode="w+")
        cls.jobs[name] = (subprocess.Popen(cmd, stdout=out, stderr=subprocess.STDOUT, text=True, env=env, cwd=cwd), out)

    @classmethod
    def wait(cls, name, timeout=3000):
        """-> (returncode, everything the child printed); the job is forgotten."""
        import subprocess

        proc, out = cls.jobs.pop(name)
        try:
            proc.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.wait()
            out.close()
            raise
        out.seek(0)
        text = out.read()
        out.close()
        return proc.returncode, text

    @classmethod
    def reap_all(cls):
        """(runs nobody waited for: -x after a failure, a keyboard interrupt)"""
        for proc, out in cls.jobs.values():
            if proc.poll() is None:
                proc.kill()
                proc.wait()
            out.close()
        cls.jobs.clear()
