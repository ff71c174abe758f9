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
# upsizing half of resize (resize.c:230-300): vips_affine with the nearest / bilinear / bicubic
# interpolators, vips_zoom for integral nearest; mixed up / down; every format of the path
for _dtype in (np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32):
    for _kernel in ("linear", "cubic"):
        RESAMPLE_CASES.append(_case("resize", "scale=2.3,kernel=%s" % _kernel, 61, 45, 3, _dtype, 22,
                                    call=("resize", dict(scale=2.3, kernel=_kernel))))
for _args, _kw in (("scale=0.1,kernel=nearest", dict(scale=0.1, kernel="nearest")),
                   ("scale=0.37,vscale=0.21,kernel=nearest", dict(scale=0.37, vscale=0.21, kernel="nearest")),
                   ("scale=0.05,vscale=1.6,kernel=nearest", dict(scale=0.05, vscale=1.6, kernel="nearest"))):
    RESAMPLE_CASES.append(_case("resize", _args, 413, 290, 3, np.uint8, 24, call=("resize", _kw)))
    RESAMPLE_CASES.append(_case("resize", _args, 211, 300, 2, np.int16, 24, call=("resize", _kw)))
for _args, _kw in (("scale=3,vscale=2,kernel=nearest", dict(scale=3.0, vscale=2.0, kernel="nearest")),
                   ("scale=2.5,vscale=1.7,kernel=nearest", dict(scale=2.5, vscale=1.7, kernel="nearest")),
                   ("scale=1.5,vscale=0.7,kernel=lanczos3", dict(scale=1.5, vscale=0.7, kernel="lanczos3")),
                   ("scale=0.6,vscale=1.9,kernel=linear", dict(scale=0.6, vscale=1.9, kernel="linear")),
                   ("scale=4,kernel=lanczos3", dict(scale=4.0, kernel="lanczos3")),
                   ("scale=1.01,vscale=7.3,kernel=mitchell", dict(scale=1.01, vscale=7.3, kernel="mitchell"))):
    for _bands, _dtype in ((4, np.uint8), (1, np.uint16), (2, np.float32)):
        RESAMPLE_CASES.append(_case("resize", _args, 83, 37, _bands, _dtype, 23, call=("resize", _kw)))


# ----------------------------------------------------------------- conv / colour cases
# kind: "mask" (conv/convsep with an explicit mask), "op" (one-input op), each with the
# reference nickname + args and the equivalent Image / PortCC method + kwargs.

_rng = np.random.RandomState(1234)
MASKS = {
    "blur3": (np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], dtype=np.float64), 16.0, 0.0),
    "rand5x7": (np.round(_rng.randn(5, 7) * 3, 3), 2.5, 1.7),
    "sobel": (np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=np.float64), 1.0, 128.0),
    "zeros": (np.zeros((3, 3)), 1.0, 5.0),
    "row5": (np.array([[1.0, 2, 3, 4, 10]]), 20.0, 3.0),
}

INTERP = {"multiband": 0, "b-w": 1, "xyz": 12, "lab": 13, "labs": 21, "srgb": 22, "rgb16": 25,
          "grey16": 26, "scrgb": 28}

CC_CASES = []


def _cc(name, **kw):
    kw["name"] = name
    CC_CASES.append(kw)


for _dtype in (np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32, np.float64):
    for _m in ("blur3", "rand5x7", "sobel"):
        for _prec in ("integer", "float"):
            _cc("conv|%s|%s|%s" % (np.dtype(_dtype).name, _m, _prec), kind="mask", op="conv", mask=_m,
                args="precision=%s" % _prec, method="conv", kwargs=dict(precision=_prec),
                width=53, height=41, bands=3, dtype=np.dtype(_dtype), seed=31, interp="multiband")
_cc("conv|zeros|integer", kind="mask", op="conv", mask="zeros", args="precision=integer", method="conv",
    kwargs=dict(precision="integer"), width=20, height=10, bands=1, dtype=np.dtype(np.uint8), seed=32,
    interp="multiband")
for _prec in ("integer", "float"):
    _cc("convsep|row5|%s" % _prec, kind="mask", op="convsep", mask="row5", args="precision=%s" % _prec,
        method="convsep", kwargs=dict(precision=_prec), width=50, height=40, bands=2,
        dtype=np.dtype(np.uint8), seed=33, interp="multiband")
for _dtype in (np.uint8, np.uint16, np.float32):
    for _prec in ("integer", "float"):
        _cc("gaussblur|%s|%s" % (np.dtype(_dtype).name, _prec), kind="op", op="gaussblur",
            args="sigma=2.5,precision=%s" % _prec, method="gaussblur",
            kwargs=dict(sigma=2.5, precision=_prec), width=80, height=60, bands=3, dtype=np.dtype(_dtype),
            seed=34, interp="multiband")
# the BASELINE config 3 mask: sigma 8 -> 29 taps
_cc("gaussblur|sigma8|float32", kind="op", op="gaussblur", args="sigma=8", method="gaussblur",
    kwargs=dict(sigma=8.0), width=96, height=64, bands=3, dtype=np.dtype(np.float32), seed=35,
    interp="srgb")

_SPACE_INPUT = {"srgb": np.uint8, "scrgb": np.float32, "xyz": np.float32, "lab": np.float32,
                "labs": np.int16}
for _fr in ("srgb", "scrgb", "xyz", "lab", "labs"):
    for _to in ("srgb", "scrgb", "xyz", "lab", "labs"):
        for _bands in (3, 4):
            _cc("colourspace|%s|%s|%d" % (_fr, _to, _bands), kind="op", op="colourspace",
                args="space=%s" % _to, method="colourspace", kwargs=dict(space=_to), width=37, height=29,
                bands=_bands, dtype=np.dtype(_SPACE_INPUT[_fr]), seed=36, interp=_fr, space_input=_fr)
# BASELINE config 3 colour half: float pixels 0..255 tagged sRGB -> Lab
_cc("colourspace|float-srgb|lab", kind="op", op="colourspace", args="space=lab", method="colourspace",
    kwargs=dict(space="lab"), width=64, height=48, bands=3, dtype=np.dtype(np.float32), seed=37,
    interp="srgb")
_cc("colourspace|ushort-srgb|xyz", kind="op", op="colourspace", args="space=xyz", method="colourspace",
    kwargs=dict(space="xyz"), width=64, height=48, bands=3, dtype=np.dtype(np.uint16), seed=38,
    interp="srgb")
for _bands in (3, 4):
    _cc("sharpen|default|%d" % _bands, kind="op", op="sharpen", args="", method="sharpen", kwargs={},
        width=97, height=71, bands=_bands, dtype=np.dtype(np.uint8), seed=39, interp="srgb")
_cc("sharpen|params", kind="op", op="sharpen", args="sigma=1.5,m1=1,m2=2,x1=3", method="sharpen",
    kwargs=dict(sigma=1.5, m1=1.0, m2=2.0, x1=3.0), width=97, height=71, bands=3,
    dtype=np.dtype(np.uint8), seed=40, interp="srgb")
# BASELINE config 1 shape (vipsthumbnail 4096^2 -> 512^2: shrink 4 + reduce 2 per axis), scaled
_cc("thumbnail|c1", kind="op", op="thumbnail_image", args="width=64,height=64", method="thumbnail_image",
    kwargs=dict(width=64, height=64), width=512, height=512, bands=3, dtype=np.dtype(np.uint8), seed=42,
    interp="srgb")
_cc("thumbnail|fit", kind="op", op="thumbnail_image", args="width=100,height=40", method="thumbnail_image",
    kwargs=dict(width=100, height=40), width=517, height=389, bands=3, dtype=np.dtype(np.uint8), seed=43,
    interp="srgb")
_cc("thumbnail|force", kind="op", op="thumbnail_image", args="width=100,height=40,size=force",
    method="thumbnail_image", kwargs=dict(width=100, height=40, size="force"), width=517, height=389,
    bands=3, dtype=np.dtype(np.uint8), seed=44, interp="srgb")
_cc("thumbnail|linear", kind="op", op="thumbnail_image", args="width=90,linear=true",
    method="thumbnail_image", kwargs=dict(width=90, linear=True), width=517, height=389, bands=3,
    dtype=np.dtype(np.uint8), seed=45, interp="srgb")
# alpha: premultiply / unpremultiply and the RGBA thumbnail around them
for _dtype in (np.uint8, np.uint16, np.int16, np.float32):
    for _uchar in (False, True):
        _cc("premultiply|%s|%d" % (np.dtype(_dtype).name, _uchar), kind="op", op="premultiply",
            args="uchar=%s" % ("true" if _uchar else "false"), method="premultiply",
            kwargs=dict(uchar=_uchar), width=45, height=31, bands=4, dtype=np.dtype(_dtype), seed=46,
            interp="srgb")
        if _uchar and _dtype != np.uint8:
            # vips_unpremultiply_gen takes its uchar loop whenever the flag is set, whatever
            # the format (unpremultiply.c:222): undefined on non-uchar data, not a parity case
            continue
        _cc("unpremultiply|%s|%d" % (np.dtype(_dtype).name, _uchar), kind="op", op="unpremultiply",
            args="uchar=%s" % ("true" if _uchar else "false"), method="unpremultiply",
            kwargs=dict(uchar=_uchar), width=45, height=31, bands=4, dtype=np.dtype(_dtype), seed=47,
            interp="srgb")
_cc("premultiply|uint8|5band", kind="op", op="premultiply", args="uchar=true", method="premultiply",
    kwargs=dict(uchar=True), width=33, height=21, bands=5, dtype=np.dtype(np.uint8), seed=48,
    interp="multiband")
_cc("thumbnail|rgba", kind="op", op="thumbnail_image", args="width=100", method="thumbnail_image",
    kwargs=dict(width=100), width=517, height=389, bands=4, dtype=np.dtype(np.uint8), seed=49,
    interp="srgb")
_cc("thumbnail|rgba16-linear", kind="op", op="thumbnail_image", args="width=80,linear=true",
    method="thumbnail_image", kwargs=dict(width=80, linear=True), width=400, height=300, bands=4,
    dtype=np.dtype(np.uint8), seed=50, interp="srgb")
_FMT = {"uint8": "uchar", "int8": "char", "uint16": "ushort", "int16": "short", "uint32": "uint",
        "int32": "int", "float32": "float", "float64": "double"}
for _a in (np.uint8, np.int16, np.uint32, np.float32, np.float64):
    for _b in (np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32, np.float64):
        _cc("cast|%s|%s" % (np.dtype(_a).name, np.dtype(_b).name), kind="op", op="cast",
            args="format=%s" % _FMT[np.dtype(_b).name], method="cast",
            kwargs=dict(format=_FMT[np.dtype(_b).name]), width=40, height=30, bands=2, dtype=np.dtype(_a),
            seed=41, interp="multiband", spread=True)


def cc_input(case):
    """The synthetic input of a CC case (needs tests.helpers)."""
    from tests import helpers

    a = helpers.lcg_image(case["width"], case["height"], case["bands"], np.uint8, case["seed"])
    space = case.get("space_input")
    if space == "scrgb":
        return (a.astype(np.float32) / 200.0 - 0.1).astype(np.float32)
    if space == "xyz":
        return (a.astype(np.float32) / 2.3).astype(np.float32)
    if space == "lab":
        f = a.astype(np.float32)
        f[:, :, 0] = f[:, :, 0] / 2.55
        f[:, :, 1:3] -= 128
        return f
    if space == "labs":
        s = helpers.lcg_image(case["width"], case["height"], case["bands"], np.int16, case["seed"])
        s[:, :, 0] = np.abs(s[:, :, 0])
        return s
    src = helpers.lcg_image(case["width"], case["height"], case["bands"], case["dtype"], case["seed"])
    if case.get("spread") and src.dtype.kind == "f":
        src = (src - 100) * 300  # exercise the clipping of float -> int casts
    return src


def cc_reference(case):
    from tests import helpers

    src = cc_input(case)
    interp = INTERP[case["interp"]]
    if case["kind"] == "mask":
        mask, scale, offset = MASKS[case["mask"]]
        return helpers.Ref.run_mask(case["op"], src, mask, scale, offset, case["args"], interp)
    return helpers.Ref.run(case["op"], src, case["args"], interp)


# ----------------------------------------------------------------- precision=approximate
# vips_conva / vips_convasep.  Float inputs are k + j/256 pixels, whose
# box sums are exact in float, so the reference's rolling sums do not depend on tile geometry.
# double convasep is left out: its second pass sums inexact doubles in tile order.
_g3 = np.array([[1, 2, 4, 2, 1], [2, 6, 12, 6, 2], [4, 12, 20, 12, 4], [2, 6, 12, 6, 2], [1, 2, 4, 2, 1]],
               dtype=np.float64)
_g13 = np.rint(20 * np.exp(-(np.arange(-6, 7)[None, :] ** 2 + np.arange(-6, 7)[:, None] ** 2) / 18.0))
_log7 = np.array([[0, 0, -1, -1, -1, 0, 0], [0, -1, -3, -3, -3, -1, 0], [-1, -3, 0, 7, 0, -3, -1],
                  [-1, -3, 7, 24, 7, -3, -1], [-1, -3, 0, 7, 0, -3, -1], [0, -1, -3, -3, -3, -1, 0],
                  [0, 0, -1, -1, -1, 0, 0]], dtype=np.float64)
_row29 = np.array([[4, 5, 6, 8, 9, 11, 12, 14, 15, 16, 18, 19, 19, 20, 20, 20, 19, 19, 18, 16, 15, 14, 12, 11, 9,
                    8, 6, 5, 4]], dtype=np.float64)
CA_MASKS = {
    "g5": (_g3, float(_g3.sum()), 0.0),
    "g13": (_g13, float(_g13.sum()), 0.0),
    "box5": (np.ones((5, 5)), 25.0, 0.0),
    "log7": (_log7, 4.0, 100.0),
    "frac": (np.round(_rng.rand(4, 6) * 9 - 1.5, 2), 31.5, -2.4),
    "row29": (_row29, 372.0, 0.0),
    "row5": (np.array([[1.0, 2, 3, 4, 10]]), 20.0, 3.0),
    "rowneg": (np.array([[-1.0, -3, 12, -3, -1]]), 4.0, 10.0),
}
CA_CASES = []


def _ca(name, **kw):
    kw["name"] = name
    CA_CASES.append(kw)


_ALL = (np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32, np.float64)
for _dtype in _ALL:
    for _m in ("g5", "g13", "log7"):
        _ca("conva|%s|%s" % (np.dtype(_dtype).name, _m), op="conva", mask=_m, args="",
            method="conva", kwargs={}, width=53, height=41, bands=3, dtype=np.dtype(_dtype), seed=71)
for _m, _layers, _cluster in (("g13", 3, 2), ("g13", 12, 1), ("g13", 7, 4), ("box5", 5, 1), ("frac", 9, 1),
                              ("log7", 20, 3), ("g5", 1, 1)):
    for _dtype in (np.uint8, np.int16, np.float32):
        _ca("conva|%s|%s|l%d|c%d" % (np.dtype(_dtype).name, _m, _layers, _cluster), op="conva", mask=_m,
            args="layers=%d,cluster=%d" % (_layers, _cluster), method="conva",
            kwargs=dict(layers=_layers, cluster=_cluster), width=47, height=39, bands=2,
            dtype=np.dtype(_dtype), seed=72)
for _dtype in _ALL[:-1]:
    for _m in ("row29", "row5", "rowneg"):
        _ca("convasep|%s|%s" % (np.dtype(_dtype).name, _m), op="convasep", mask=_m, args="",
            method="convasep", kwargs={}, width=61, height=45, bands=3, dtype=np.dtype(_dtype), seed=73)
for _layers in (1, 3, 12):
    for _dtype in (np.uint8, np.uint16, np.float32):
        _ca("convasep|%s|row29|l%d" % (np.dtype(_dtype).name, _layers), op="convasep", mask="row29",
            args="layers=%d" % _layers, method="convasep", kwargs=dict(layers=_layers), width=64, height=40,
            bands=1, dtype=np.dtype(_dtype), seed=74)
# the front doors: vips_conv / vips_convsep / vips_gaussblur with precision=approximate
_ca("conv|approximate|uint8", op="conv", mask="g13", args="precision=approximate", method="conv",
    kwargs=dict(precision="approximate"), width=80, height=60, bands=3, dtype=np.dtype(np.uint8), seed=75)
_ca("conv|approximate|uint16|l8", op="conv", mask="g13", args="precision=approximate,layers=8,cluster=2",
    method="conv", kwargs=dict(precision="approximate", layers=8, cluster=2), width=80, height=60, bands=1,
    dtype=np.dtype(np.uint16), seed=76)
_ca("convsep|approximate|uint8", op="convsep", mask="row29", args="precision=approximate", method="convsep",
    kwargs=dict(precision="approximate"), width=80, height=60, bands=4, dtype=np.dtype(np.uint8), seed=77)
for _dtype in (np.uint8, np.uint16, np.int16, np.float32):
    _ca("gaussblur|approximate|%s" % np.dtype(_dtype).name, op="gaussblur", mask=None,
        args="sigma=8,precision=approximate", method="gaussblur", kwargs=dict(sigma=8.0, precision="approximate"),
        width=120, height=90, bands=3, dtype=np.dtype(_dtype), seed=78)


def ca_input(case):
    from tests import helpers

    return helpers.lcg_image(case["width"], case["height"], case["bands"], case["dtype"], case["seed"])


def ca_reference(case):
    from tests import helpers

    src = ca_input(case)
    if case["mask"] is None:
        return helpers.Ref.run(case["op"], src, case["args"])
    mask, scale, offset = CA_MASKS[case["mask"]]
    return helpers.Ref.run_mask(case["op"], src, mask, scale, offset, case["args"])


def extra_groups():
    def make():
        return {c["name"]: cc_reference(c) for c in CC_CASES}

    def make_ca():
        return {c["name"]: ca_reference(c) for c in CA_CASES}

    return [("conv_colour.npz", make), ("conva.npz", make_ca)]
