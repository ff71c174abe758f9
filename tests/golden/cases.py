"""The golden-vector case list shared by make_golden.py and the tests.

Each case names a reference operation (nickname + argument string as accepted by
vips_object_set_from_string) and a synthetic input (tests.helpers.lcg_image).
"""
import numpy as np


def _case(op, args, width, height, bands, dtype, seed, **extra):
    name = "%s|%s|%dx%dx%d|%s|%d" % (op, args, width, height, bands, np.dtype(dtype).name, seed)
    d = dict(name=name, op=op, args=args, width=width, height=height, bands=bands,
             dtype=np.dtype(dtype), seed=seed)
    d.update(extra)
    return d


RESAMPLE_CASES = []

# reduce across kernels / fractional factors / formats (test_resample.py:77-111 walks
# fac in {1, 1.1, 1.5, 1.999} x formats x kernels)
for _dtype in (np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32):
    for _kernel in ("nearest", "linear", "cubic", "mitchell", "lanczos2", "lanczos3", "mks2013", "mks2021"):
        RESAMPLE_CASES.append(_case("reduce", "hshrink=1.5,vshrink=1.999,kernel=%s" % _kernel,
                                    61, 47, 3, _dtype, 11,
                                    call=("reduce", dict(hshrink=1.5, vshrink=1.999, kernel=_kernel))))
for _fac in (1.1, 2.0, 3.7, 8.0):
    for _bands in (1, 3, 4):
        RESAMPLE_CASES.append(_case("reduce", "hshrink=%g,vshrink=%g,kernel=lanczos3" % (_fac, _fac),
                                    131, 97, _bands, np.uint8, 12,
                                    call=("reduce", dict(hshrink=_fac, vshrink=_fac, kernel="lanczos3"))))
# gap > 0: integer pre-shrink + residual reduce (reducev.cpp:895-921)
for _dtype in (np.uint8, np.uint16, np.float32):
    RESAMPLE_CASES.append(_case("reduce", "hshrink=6.3,vshrink=5.1,kernel=lanczos3,gap=2", 257, 203, 3,
                                _dtype, 13,
                                call=("reduce", dict(hshrink=6.3, vshrink=5.1, kernel="lanczos3", gap=2.0))))
# single axis
RESAMPLE_CASES.append(_case("reducev", "vshrink=2.5,kernel=cubic", 40, 301, 4, np.uint8, 14,
                            call=("reducev", dict(vshrink=2.5, kernel="cubic"))))
RESAMPLE_CASES.append(_case("reduceh", "hshrink=2.5,kernel=cubic", 301, 40, 4, np.uint8, 15,
                            call=("reduceh", dict(hshrink=2.5, kernel="cubic"))))
# shrink: every format, both rounding modes (shrinkh.c:78-232, shrinkv.c:158-310)
for _dtype in (np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32, np.float64):
    for _ceil in (False, True):
        RESAMPLE_CASES.append(_case("shrink", "hshrink=3,vshrink=4,ceil=%s" % ("true" if _ceil else "false"),
                                    64, 50, 3, _dtype, 16,
                                    call=("shrink", dict(hshrink=3, vshrink=4, ceil=_ceil))))
RESAMPLE_CASES.append(_case("shrinkh", "hshrink=7", 100, 9, 1, np.uint8, 17,
                            call=("shrinkh", dict(hshrink=7))))
RESAMPLE_CASES.append(_case("shrinkv", "vshrink=7", 9, 100, 1, np.uint8, 18,
                            call=("shrinkv", dict(vshrink=7))))
# non-integer shrink goes through reduce with gap 1 (shrink.c:98-110)
RESAMPLE_CASES.append(_case("shrink", "hshrink=2.5,vshrink=3.5", 120, 90, 3, np.uint8, 19,
                            call=("shrink", dict(hshrink=2.5, vshrink=3.5))))
# resize = the thumbnail pipeline (resize.c:135-329, gap 2): shrinkv, reducev, shrinkh, reduceh
for _bands in (3, 4):
    RESAMPLE_CASES.append(_case("resize", "scale=0.125", 512, 384, _bands, np.uint8, 20,
                                call=("resize", dict(scale=0.125))))
RESAMPLE_CASES.append(_case("resize", "scale=0.3,vscale=0.21,kernel=mitchell", 300, 260, 3, np.uint16, 21,
                            call=("resize", dict(scale=0.3, vscale=0.21, kernel="mitchell"))))
