"""GPU parity: HIP conv / gaussblur / sharpen / colourspace / cast through the C ABI vs
the oracle (golden vectors from the compiled reference, the plain-C port, oracle/_ref).
Integer outputs bit-exact; float outputs within 1 ULP (they come out bitwise equal: the
device keeps the reference's operation order in double / float without FMA)."""
import ctypes
import os

import numpy as np
import pytest

import libvips_amd
from libvips_amd import Image, _ffi
from tests import helpers
from tests.golden import cases
from tests.helpers import Port, PortCC, Ref
from tests.test_resample_gpu import assert_same

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(helpers.GOLDEN, "conv_colour.npz"))


@pytest.fixture(scope="module", autouse=True)
def _init():
    libvips_amd.init(0)


def hip_call(case, src):
    im = Image.new_from_array(src, interpretation=case["interp"])
    kw = dict(case["kwargs"])
    if case["kind"] == "mask":
        mask, scale, offset = cases.MASKS[case["mask"]]
        return getattr(im, case["method"])(mask, scale=scale, offset=offset, **kw).numpy()
    return getattr(im, case["method"])(**kw).numpy()


@pytest.mark.parametrize("case", cases.CC_CASES, ids=[c["name"] for c in cases.CC_CASES])
def test_hip_matches_golden(case):
    src = cases.cc_input(case)
    assert_same(hip_call(case, src), GOLD[case["name"]], case["name"])


def test_gaussmat_matches_port():
    for sigma, min_ampl, sep, prec in ((8, 0.2, True, "integer"), (0.5, 0.1, True, "integer"),
                                       (5, 0.01, False, "float"), (1.2, 0.2, False, "integer"),
                                       (3, 0.2, True, "float")):
        m, s = libvips_amd.gaussmat(sigma, min_ampl, sep, prec)
        pm, ps = PortCC.gaussmat(sigma, min_ampl, sep, prec)
        assert np.array_equal(m, pm) and s == ps


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.float32])
def test_conv_edges_and_windows(dtype):
    """Region-level conv: an output rect in the middle of the image with an input window
    that only just covers it, plus rects touching every edge (clamped taps)."""
    lib = _ffi.lib
    src = helpers.lcg_image(90, 70, 2, dtype, 61)
    mask, scale, offset = cases.MASKS["rand5x7"]
    for prec in ("integer", "float"):
        want = PortCC.conv(src, mask, scale, offset, prec)
        c = lib.vips_hip_conv_new(mask.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), mask.shape[1],
                                  mask.shape[0], scale, offset, libvips_amd.PRECISIONS[prec])
        assert c
        try:
            for (left, top, w, h) in ((20, 15, 40, 30), (0, 0, 17, 9), (70, 55, 20, 15), (0, 60, 90, 10)):
                x0, y0 = max(left - 3, 0), max(top - 2, 0)
                x1, y1 = min(left + w + 3, 90), min(top + h + 2, 70)
                win = Image.new_from_array(np.ascontiguousarray(src[y0:y1, x0:x1]))
                rin = win.region()
                rin.left, rin.top, rin.im_width, rin.im_height = x0, y0, 90, 70
                out = Image.new_from_array(np.zeros((h, w, 2), want.dtype))
                rout = out.region()
                rout.left, rout.top, rout.im_width, rout.im_height = left, top, 90, 70
                _ffi.check(lib.vips_hip_conv_gen(c, ctypes.byref(rin), ctypes.byref(rout)))
                assert_same(out.numpy(), np.ascontiguousarray(want[top:top + h, left:left + w]),
                            str((prec, left, top)))
            # too-small window
            win = Image.new_from_array(np.ascontiguousarray(src[15:45, 20:60]))
            rin = win.region()
            rin.left, rin.top, rin.im_width, rin.im_height = 20, 15, 90, 70
            lib.vips_hip_error_clear()
            assert lib.vips_hip_conv_gen(c, ctypes.byref(rin), ctypes.byref(rout)) == -1
            assert "input region too small" in _ffi.error_buffer()
            lib.vips_hip_error_clear()
        finally:
            lib.vips_hip_conv_free(c)


def test_colour_single_steps_vs_port():
    """Every process_line on its own (the per-step region op)."""
    lib = _ffi.lib
    steps = {"sRGB2scRGB": 0, "scRGB2XYZ": 1, "XYZ2Lab": 2, "Lab2XYZ": 3, "XYZ2scRGB": 4,
             "scRGB2sRGB": 5, "Lab2LabS": 7, "LabS2Lab": 8}
    inputs = {"sRGB2scRGB": "srgb", "scRGB2XYZ": "scrgb", "XYZ2Lab": "xyz", "Lab2XYZ": "lab",
              "XYZ2scRGB": "xyz", "scRGB2sRGB": "scrgb", "Lab2LabS": "lab", "LabS2Lab": "labs"}
    for name, step in steps.items():
        case = dict(width=61, height=43, bands=3, seed=62, space_input=inputs[name],
                    dtype=np.dtype(np.uint8))
        src = cases.cc_input(case)
        want, _ = PortCC.colour_step(src, name, inputs[name])
        im = Image.new_from_array(src)
        out = Image.new_from_array(np.zeros(want.shape, want.dtype))
        ri, ro = im.region(), out.region()
        _ffi.check(lib.vips_hip_colour_gen(step, ctypes.byref(ri), ctypes.byref(ro)))
        assert_same(out.numpy(), want, name)


def test_colour_special_values():
    # NaN -> 0 in scRGB2sRGB (LabQ2sRGB.c:312-318), out-of-gamut clipping, huge XYZ
    src = np.array([[[np.nan, 0.5, 0.5], [2.0, -1.0, 0.5], [0.0, 1.0, 0.999999],
                     [1e30, 1e-30, -1e30]]], np.float32)
    got = Image.new_from_array(src, interpretation="scrgb").colourspace("srgb").numpy()
    want = PortCC.colourspace(src, "srgb", "scrgb")
    assert np.array_equal(got, want)
    got = Image.new_from_array(src[:, 1:], interpretation="xyz").colourspace("lab").numpy()
    want = PortCC.colourspace(src[:, 1:], "lab", "xyz")
    assert_same(got, want)


def test_lab_to_xyz_known_answer():
    xyz = Image.new_from_array(np.array([[[50, 0, 0]]], np.float32), interpretation="lab") \
        .colourspace("xyz").numpy()[0, 0]
    assert np.allclose(xyz, [17.5064, 18.4187, 20.0547], atol=1e-4)


def test_error_behaviour():
    im = Image.new_from_array(helpers.lcg_image(20, 20, 3, np.uint8, 63))
    with pytest.raises(libvips_amd.VipsHipError, match="positive"):
        im.conv([[-1.0]], precision="approximate")  # the reference fails here too
    with pytest.raises(libvips_amd.VipsHipError, match="no known route"):
        Image.new_from_array(helpers.lcg_image(20, 20, 1, np.uint8, 63), interpretation="b-w") \
            .colourspace("lab")
    with pytest.raises(libvips_amd.VipsHipError, match="no known route"):
        im.sharpen()  # multiband 3-band uchar is guessed sRGB on the way in, but there is
        # no route back to 'multiband' (same failure as the reference)


def test_c3_pipeline_reduced():
    """BASELINE config 3 (gaussblur sigma 8 -> sRGB->Lab on float), reduced size, vs the port
    and, when present, the compiled reference."""
    src = helpers.lcg_image(512, 384, 3, np.float32, 64)
    got = Image.new_from_array(src, interpretation="srgb").gaussblur(8.0).colourspace("lab").numpy()
    assert_same(got, PortCC.colourspace(PortCC.gaussblur(src, 8.0), "lab", "srgb"))
    if helpers.have_ref():
        want = Ref.run_chain("gaussblur:sigma=8;colourspace:space=lab", src, cases.INTERP["srgb"])
        assert_same(got, want)


def test_c4_pipeline_reduced():
    """BASELINE config 4 per image: resize(1/8) -> sharpen -> sRGB u8."""
    src = helpers.lcg_image(2048, 1536, 3, np.uint8, 65)
    got = Image.new_from_array(src, interpretation="srgb").resize(0.125).sharpen().numpy()
    if helpers.have_ref():
        want = Ref.run_chain("resize:scale=0.125;sharpen:", src, cases.INTERP["srgb"])
    else:
        want = PortCC.sharpen(Port.resize(src, 0.125), "srgb")
    assert np.array_equal(got, want)


def test_c5_conv31_reduced(float_mode):
    """BASELINE config 5 kernel: 31x31 float gaussian on ushort, one GPU, reduced size; both float
    modes (exact: bit for bit; default: 1 ULP)."""
    src = helpers.lcg_image(700, 500, 1, np.uint16, 66)
    mask, scale = libvips_amd.gaussmat(5, 0.01, False, "float")
    got = Image.new_from_array(src).conv(mask, scale=scale, precision="float").numpy()
    if helpers.have_ref():
        want = Ref.run_mask("conv", src, mask, scale, 0.0, "precision=float")
    else:
        want = PortCC.conv(src, mask, scale, 0.0, "float")
    if float_mode == "default":
        assert got.shape == want.shape and ulp_distance(got, want) <= 1
    else:
        assert_same(got, want)


@pytest.mark.parametrize("precision", ["integer", "float"])
@pytest.mark.parametrize("shape", [(700, 300), (2300, 140), (37, 411), (256, 64), (5, 3), (1030, 77)])
@pytest.mark.parametrize("sigma,space", [(8.0, "lab"), (2.0, "xyz"), (0.6, "scrgb"), (3.1, "lab")])
def test_gaussblur_colourspace_fused(shape, sigma, space, precision):
    """vips_hip_gaussblur_colourspace (convsep_stream.hip with the colour epilogue: BASELINE
    config 3 in one kernel) against the same two operations run one after the other on the older
    kernels, and against the port: several strips wide, several row segments, narrow / tiny
    images (all four edges clamped), 3..29 taps, both precisions; bit for bit."""
    w, h = shape
    src = helpers.lcg_image(w, h, 3, np.float32, 68)
    src[3 % h, 5 % w] = [300.0, -7.5, 128.25]  # out of the 0..255 range: clipped by the sRGB decode
    lib = _ffi.lib
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    try:
        im = Image.new_from_array(src, interpretation="srgb")
        got = im.gaussblur_colourspace(sigma, space, precision=precision).numpy()
        report = libvips_amd.gate_report()
    finally:
        lib.vips_hip_gate_enable(0)
        lib.vips_hip_gate_reset()
    assert any(k.startswith("convsep_stream") and k.endswith("_colour") for k in report) and len(report) == 1, report
    os.environ["VIPS_HIP_NO_STREAM_CONVSEP"] = "1"
    try:
        two_ops = Image.new_from_array(src, interpretation="srgb").gaussblur(sigma, precision=precision).colourspace(space).numpy()
    finally:
        del os.environ["VIPS_HIP_NO_STREAM_CONVSEP"]
    assert got.dtype == np.float32 and got.shape == two_ops.shape
    assert np.array_equal(got.view(np.int32), two_ops.view(np.int32))
    want = PortCC.colourspace(PortCC.gaussblur(src, sigma, precision=precision), space, "srgb")
    assert np.array_equal(got.view(np.int32), want.view(np.int32))
    # and the unfused call of the new kernel (blur only)
    blur = Image.new_from_array(src, interpretation="srgb").gaussblur(sigma, precision=precision).numpy()
    assert np.array_equal(blur.view(np.int32), PortCC.gaussblur(src, sigma, precision=precision).view(np.int32))


def test_gaussblur_colourspace_fallbacks():
    """Images the fused kernel does not take (uchar, 4 bands, a route that ends in a coding
    step) go through the two operations and give the same pixels as calling them."""
    u8 = helpers.lcg_image(300, 200, 3, np.uint8, 69)
    a = Image.new_from_array(u8, interpretation="srgb").gaussblur_colourspace(2.0, "lab").numpy()
    b = Image.new_from_array(u8, interpretation="srgb").gaussblur(2.0).colourspace("lab").numpy()
    assert np.array_equal(a.view(np.int32), b.view(np.int32))
    f4 = helpers.lcg_image(120, 90, 4, np.float32, 70)
    a = Image.new_from_array(f4, interpretation="srgb").gaussblur_colourspace(2.0, "lab").numpy()
    b = Image.new_from_array(f4, interpretation="srgb").gaussblur(2.0).colourspace("lab").numpy()
    assert np.array_equal(a.view(np.int32), b.view(np.int32))
    f3 = helpers.lcg_image(120, 90, 3, np.float32, 71)
    a = Image.new_from_array(f3, interpretation="srgb").gaussblur_colourspace(2.0, "labs").numpy()
    b = Image.new_from_array(f3, interpretation="srgb").gaussblur(2.0).colourspace("labs").numpy()
    assert a.dtype == np.int16 and np.array_equal(a, b)


@pytest.mark.parametrize("size", [(1024, 768), (67, 19), (64, 16), (130, 33), (5, 3), (1000, 1)])
@pytest.mark.parametrize("params", [dict(), dict(sigma=1.0), dict(sigma=0.3), dict(sigma=2.0),
                                    dict(sigma=0.8, x1=1.0, y2=20.0, y3=30.0, m1=0.5, m2=2.0),
                                    dict(x1=0.5, y2=80.0, y3=90.0, m2=1.0)])
@pytest.mark.parametrize("quad", [True, False, "skip", "skip-blocky", "adaptive", "adaptive-blocky"])
def test_sharpen_fused_uchar_srgb(size, params, quad, monkeypatch):
    """vips_sharpen on 3-band uchar sRGB in one kernel (colour.hip sharpen_fused_u8: the LabS round
    trip, the integer blur of L in LDS and the LUT step): tiles with partial edges, images smaller
    than a tile, 1..5-tap masks in the kernel and longer ones on the operation chain; against the
    port / the compiled reference and against the chain of six kernels."""
    # quad: every table in LDS (sharpen_quad_u8, cbrt_quad.h) -- the default when the LUT's bending part is short
    # enough (the last parameter set's is not); else the kernel that reads its tables through global memory
    # (by default only images of 4 Mpixels and more take it: forced here)
    # skip (round 6): the gather kernel with the pixels the LUT leaves alone handed through untouched
    # (sharpen_fused_u8_kernel<*, true>; the default for images under 4 Mpixels) -- on noise nearly every pixel
    # is on the list that goes the whole way, on the blocky image (flat 8 x 8 blocks of slowly varying colour, with
    # a few hard edges) nearly none; True / False: the two older kernels, every pixel the whole way
    # adaptive: what large images take by default -- the skip kernel first, the tiles whose list is long (here: longer
    # than 100 of 2 048 pixels, so that both kernels get tiles of one image) left on a device-side list for the
    # all-in-LDS kernel
    adaptive = isinstance(quad, str) and quad.startswith("adaptive")
    monkeypatch.setenv("VIPS_HIP_SHARPEN_QUAD", "1" if quad is True or adaptive else "0")
    monkeypatch.setenv("VIPS_HIP_SHARPEN_SKIP", "1" if isinstance(quad, str) else "0")
    survey = quad == "adaptive-blocky"
    if adaptive:
        monkeypatch.setenv("VIPS_HIP_SHARPEN_DEFER", "100")
        # who judges a tile first: sharpen_survey_kernel from the raw bytes of one row (the default) / the skip kernel
        # itself from L of three rows (round 6's first form)
        monkeypatch.setenv("VIPS_HIP_SHARPEN_DEFER_RAW", "12" if survey else "0")
    w, h = size
    src = helpers.lcg_image(w, h, 3, np.uint8, 72)
    if isinstance(quad, str) and quad.endswith("blocky"):
        small = helpers.lcg_image((w + 7) // 8, (h + 7) // 8, 3, np.uint8, 73).astype(np.int32)
        ramp = (np.arange(small.shape[1])[None, :, None] * 3 + np.arange(small.shape[0])[:, None, None] * 2) % 200
        base = np.where(small > 240, small, 20 + ramp + small % 4).astype(np.uint8)  # (a few outliers: hard edges)
        src = np.ascontiguousarray(np.kron(base, np.ones((8, 8, 1), np.uint8))[:h, :w])
    lib = _ffi.lib
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    try:
        got = Image.new_from_array(src, interpretation="srgb").sharpen(**params).numpy()
        report = libvips_amd.gate_report()
    finally:
        lib.vips_hip_gate_enable(0)
        lib.vips_hip_gate_reset()
    os.environ["VIPS_HIP_NO_FUSED_SHARPEN"] = "1"
    try:
        chain = Image.new_from_array(src, interpretation="srgb").sharpen(**params).numpy()
    finally:
        del os.environ["VIPS_HIP_NO_FUSED_SHARPEN"]
    assert got.dtype == np.uint8 and np.array_equal(got, chain)
    if helpers.have_ref():
        args = ",".join("%s=%s" % kv for kv in params.items())
        want = Ref.run("sharpen", src, args, cases.INTERP["srgb"])
    else:
        want = PortCC.sharpen(src, "srgb", **params)
    assert np.array_equal(got, want)
    if params.get("sigma", 0.5) <= 1.0:
        # (a LUT whose bending part is longer than 6144 entries -- the last two parameter sets -- does not fit LDS)
        wide_lut = "y3" in params
        if adaptive and "y3" not in params:
            # (an image of at most 16 rows: the all-in-LDS kernel alone)
            both = ["sharpen_quad_u8", "sharpen_skip_u8"] + (["sharpen_survey"] if survey else [])
            assert sorted(report) == (both if h > 16 else ["sharpen_quad_u8"]), report
        elif isinstance(quad, str):
            assert list(report) == ["sharpen_skip_u8"], report
        else:
            assert list(report) == ["sharpen_quad_u8" if quad and not wide_lut else "sharpen_fused_u8"], report


def ulp_distance(a, b):
    """Largest distance between two float32 arrays in units in the last place."""
    ai = np.ascontiguousarray(a).view(np.int32).astype(np.int64)
    bi = np.ascontiguousarray(b).view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return int(np.abs(ai - bi).max())


@pytest.mark.parametrize("bands", [1, 3])
def test_c5_conv31_default_mode(bands):
    """The library's DEFAULT float mode on BASELINE config 5's kernel: fused multiply-adds in the
    double sums (twice the FP64 rate).  north_star grants float paths 1 ULP: asserted here against
    the compiled reference / the port, with the exact mode (bit for bit) right next to it."""
    src = helpers.lcg_image(900, 400, bands, np.uint16, 66)
    mask, scale = libvips_amd.gaussmat(5, 0.01, False, "float")
    lib = _ffi.lib
    if helpers.have_ref():
        want = Ref.run_mask("conv", src, mask, scale, 0.0, "precision=float")
    else:
        want = PortCC.conv(src, mask, scale, 0.0, "float")
    assert lib.vips_hip_get_exact_float() == 1
    exact = Image.new_from_array(src).conv(mask, scale=scale, precision="float").numpy()
    assert_same(exact, want)
    lib.vips_hip_set_exact_float(0)
    try:
        fast = Image.new_from_array(src).conv(mask, scale=scale, precision="float").numpy()
    finally:
        lib.vips_hip_set_exact_float(1)
    assert fast.dtype == np.float32 and fast.shape == want.shape
    assert ulp_distance(fast, want) <= 1  # tolerance: 1 ULP (BASELINE.json north_star)


@pytest.mark.parametrize("precision", ["integer", "float"])
@pytest.mark.parametrize("shape", [(700, 300), (2300, 140), (37, 411), (1030, 77)])
@pytest.mark.parametrize("sigma,space", [(8.0, "lab"), (2.0, None), (3.1, "xyz"), (0.6, None)])
def test_gaussblur_default_mode(shape, sigma, space, precision):
    """The library's DEFAULT float mode on the streaming separable convolution (convsep_stream.hip
    MODE 3: coefficients mask / scale, one fused multiply-add per tap, no division): within 1 ULP of
    the exact mode -- which the tests above hold to the reference bit for bit -- with and without the
    colour epilogue.  Tolerance: 1 ULP (BASELINE.json north_star, float paths)."""
    w, h = shape
    src = helpers.lcg_image(w, h, 3, np.float32, 69)
    lib = _ffi.lib

    def run():
        im = Image.new_from_array(src, interpretation="srgb")
        if space is None:
            return im.gaussblur(sigma, precision=precision).numpy()
        return im.gaussblur_colourspace(sigma, space, precision=precision).numpy()

    assert lib.vips_hip_get_exact_float() == 1
    exact = run()
    lib.vips_hip_set_exact_float(0)
    try:
        fast = run()
    finally:
        lib.vips_hip_set_exact_float(1)
    assert fast.dtype == np.float32 and fast.shape == exact.shape
    assert ulp_distance(fast, exact) <= 1


@pytest.mark.parametrize("precision", ["integer", "float"])
@pytest.mark.parametrize("shape", [(700, 300, 3), (1500, 90, 1), (37, 411, 4), (2300, 140, 2), (5, 3, 3)])
@pytest.mark.parametrize("sigma", [0.6, 2.0, 8.0])
def test_fused_convsep_float(shape, sigma, precision, float_mode):
    """convsep_f32.hip / convsep_stream.hip (both passes of a float separable conv in one streaming
    kernel): several strips wide, several row segments, narrow / tiny images (all edges clamped),
    1..4 bands, masks of 3..29 taps, convi-on-float and convf arithmetic; in the exact mode bit-exact
    against the two-operation port and against the device's own two-pass path, in the default mode
    (fused multiply-adds) within 1 ULP of the port."""
    w, h, b = shape
    src = helpers.lcg_image(w, h, b, np.float32, 67)
    lib = _ffi.lib
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    try:
        got = Image.new_from_array(src).gaussblur(sigma, precision=precision).numpy()
        report = libvips_amd.gate_report()
    finally:
        lib.vips_hip_gate_enable(0)
        lib.vips_hip_gate_reset()
    assert any(k.startswith("convsep_stream") for k in report), report
    want = PortCC.gaussblur(src, sigma, precision=precision)
    if float_mode == "default":
        assert got.dtype == np.float32 and got.shape == want.shape
        assert ulp_distance(got, want) <= 1  # tolerance: 1 ULP (BASELINE.json north_star)
        return
    assert got.dtype == np.float32 and np.array_equal(got, want)
    os.environ["VIPS_HIP_NO_FUSED_CONVSEP"] = "1"
    try:
        two_pass = Image.new_from_array(src).gaussblur(sigma, precision=precision).numpy()
    finally:
        del os.environ["VIPS_HIP_NO_FUSED_CONVSEP"]
    assert np.array_equal(got, two_pass)


def test_fused_convsep_float_offsets_and_nonfinite():
    """An explicit convsep mask with scale and offset (the offset applies to the first pass
    only, convsep.c:91-106), negative taps, and inf / nan pixels propagating exactly as in
    the reference."""
    src = helpers.lcg_image(333, 200, 3, np.float32, 68)
    src[17, 40, 1] = np.inf
    src[90, 300, 0] = np.nan
    src[150, 5, 2] = -np.inf
    mask = np.array([[-1.0, 2.0, 5.0, 7.0, 5.0, 3.0, -2.0]])
    for precision, scale, offset in (("integer", 19.0, 3.0), ("float", 18.5, -0.75), ("integer", 1.0, 0.0)):
        got = Image.new_from_array(src).convsep(mask, scale=scale, offset=offset, precision=precision).numpy()
        want = PortCC.convsep(src, mask, scale, offset, precision)
        assert np.array_equal(got, want, equal_nan=True), precision


def test_fused_convsep_float_signed_zeros():
    """The sign of zero through the streaming kernel: blocks of +0.0, -0.0 and tiny negative
    pixels, masks with a positive and with a NEGATIVE scale, offsets 0.0, -0.0 and non-zero -- the
    bits of every output (v_div_fixup_f64 restores IEEE division's sign of zero; `+ offset` is only
    skipped where it cannot matter)."""
    src = helpers.lcg_image(300, 120, 3, np.float32, 71)
    src[:40] = 0.0
    src[40:80, :150] = -0.0
    src[40:80, 150:] = -1e-40
    src[100:, 200:] = -0.0
    mask = np.array([[1.0, 2.0, 5.0, 7.0, 5.0, 2.0, 1.0]])
    for m, scale in ((mask, 23.0), (mask, -23.0), (-mask, 1.0), (mask, 1.0)):
        for offset in (0.0, -0.0, 2.5):
            got = Image.new_from_array(src).convsep(m, scale=scale, offset=offset, precision="integer").numpy()
            want = PortCC.convsep(src, m, scale, offset, "integer")
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (scale, offset)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16])
@pytest.mark.parametrize("shape", [(700, 300, 3), (1500, 90, 1), (37, 411, 4), (2300, 140, 2), (5, 3, 3)])
@pytest.mark.parametrize("sigma", [0.6, 2.0, 8.0])
def test_fused_convsep_integer(shape, sigma, dtype):
    """The same streaming kernel on uchar / ushort / short images with an integer mask (the
    convi C path: 32-bit sums, C division by the scale, clip; the intermediate image keeps
    the format): bit-exact against the two-operation port and the device's two-pass path."""
    w, h, b = shape
    src = helpers.lcg_image(w, h, b, dtype, 69)
    lib = _ffi.lib
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    try:
        got = Image.new_from_array(src).gaussblur(sigma).numpy()
        report = libvips_amd.gate_report()
    finally:
        lib.vips_hip_gate_enable(0)
        lib.vips_hip_gate_reset()
    if sigma >= 2.0:  # masks shorter than 7 taps stay on the two register-tiled passes
        # (uchar images whose rows are whole dwords: the packed-byte kernel of conv_u8.hip)
        assert any(k.startswith("convsep_") or k in ("conv_u8_sep", "conv_u8_mfma_sep", "conv_u16_mfma_sep") for k in report), report
    want = PortCC.gaussblur(src, sigma)
    assert got.dtype == src.dtype and np.array_equal(got, want)
    os.environ["VIPS_HIP_NO_FUSED_CONVSEP"] = "1"
    try:
        two_pass = Image.new_from_array(src).gaussblur(sigma).numpy()
    finally:
        del os.environ["VIPS_HIP_NO_FUSED_CONVSEP"]
    assert np.array_equal(got, two_pass)


def test_fused_convsep_integer_signs_and_offsets():
    """Negative taps, a negative sum (the C division truncates toward zero), scale and offset,
    clipping at both ends, on all three integer formats."""
    mask = np.array([[-3.0, 2.0, 9.0, -14.0, 9.0, 2.0, -3.0]])
    for dtype in (np.uint8, np.uint16, np.int16):
        src = helpers.lcg_image(257, 190, 3, dtype, 70)
        for scale, offset in ((2.0, 0.0), (3.0, 37.0), (1.0, -5.0), (7.0, 200.0)):
            got = Image.new_from_array(src).convsep(mask, scale=scale, offset=offset, precision="integer").numpy()
            want = PortCC.convsep(src, mask, scale, offset, "integer")
            assert np.array_equal(got, want), (dtype, scale, offset)


# ---------------------------------------------------------------- precision=approximate

GOLD_CA = np.load(os.path.join(helpers.GOLDEN, "conva.npz"))


def hip_approx_call(case, src):
    im = Image.new_from_array(src)
    kw = dict(case["kwargs"])
    if case["mask"] is None:
        return getattr(im, case["method"])(**kw).numpy()
    mask, scale, offset = cases.CA_MASKS[case["mask"]]
    return getattr(im, case["method"])(mask, scale=scale, offset=offset, **kw).numpy()


@pytest.fixture(params=["default", "generic"])
def approx_path(request):
    """default: 8/16-bit images whose sums cannot wrap run on the convi / fused separable kernels
    with the approximated mask; generic: the box-sum kernels for everything."""
    if request.param == "generic":
        os.environ["VIPS_HIP_NO_APPROX_FAST"] = "1"
    yield request.param
    os.environ.pop("VIPS_HIP_NO_APPROX_FAST", None)


@pytest.mark.parametrize("case", cases.CA_CASES, ids=[c["name"] for c in cases.CA_CASES])
def test_hip_approximate_matches_golden(case, approx_path):
    src = cases.ca_input(case)
    got = hip_approx_call(case, src)
    want = GOLD_CA[case["name"]]
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), case["name"]


@pytest.mark.parametrize("dtype", [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32,
                                   np.float64])
def test_hip_approximate_random_masks_vs_port(dtype, approx_path):
    rng = np.random.RandomState(21)
    src = helpers.lcg_image(75, 58, 2, dtype, 83)
    im = Image.new_from_array(src)
    for _ in range(5):
        mw, mh = rng.randint(1, 12, size=2)
        mask = rng.randint(-4, 15, size=(mh, mw)).astype(np.float64)
        mask[rng.randint(mh), rng.randint(mw)] = 17
        scale, offset = float(rng.randint(1, 50)), float(rng.randint(-4, 5))
        layers, cluster = int(rng.randint(1, 15)), int(rng.randint(1, 6))
        want = PortCC.conva(src, mask, scale, offset, layers, cluster)
        got = im.conva(mask, scale, offset, layers, cluster).numpy()
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), ("conva", mask.shape, layers, cluster)
        row = mask.reshape(-1)[:25].copy()
        row[0] = 9
        want = PortCC.convasep(src, row, scale, offset, layers)
        got = im.convasep(row, scale, offset, layers).numpy()
        if np.dtype(dtype) == np.float64:
            # the port and the device agree bit for bit (same direct sums); the reference itself
            # is tile-order dependent here
            assert np.array_equal(got, want)
        else:
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), ("convasep", row.shape, layers)


def test_hip_approximate_regions(approx_path):
    """Region-level conva / convasep: output rects inside the image and on every edge, input
    windows that only just cover them."""
    lib = _ffi.lib
    W, H = 90, 70
    for dtype in (np.uint8, np.float32):
        src = helpers.lcg_image(W, H, 2, dtype, 84)
        mask, scale, offset = cases.CA_MASKS["g13"]
        pm = np.ascontiguousarray(mask).ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        want = PortCC.conva(src, mask, scale, offset, 5, 1)
        plan = lib.vips_hip_conva_new(pm, 13, 13, scale, offset, 5, 1)
        assert plan
        row, rscale, roffset = cases.CA_MASKS["row29"]
        prow = np.ascontiguousarray(row).ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        sep = lib.vips_hip_convasep_new(prow, 29, rscale, roffset, 5)
        assert sep
        # one pass each way, against the port's two-pass result through a full-size intermediate
        want_sep = PortCC.convasep(src, row, rscale, roffset, 5)
        try:
            for (left, top, w, h) in ((20, 15, 40, 30), (0, 0, 17, 9), (70, 55, 20, 15), (0, 60, 90, 10)):
                x0, y0 = max(left - 6, 0), max(top - 6, 0)
                x1, y1 = min(left + w + 6, W), min(top + h + 6, H)
                win = Image.new_from_array(np.ascontiguousarray(src[y0:y1, x0:x1]))
                rin = win.region()
                rin.left, rin.top, rin.im_width, rin.im_height = x0, y0, W, H
                out = Image.new_from_array(np.zeros((h, w, 2), want.dtype))
                rout = out.region()
                rout.left, rout.top, rout.im_width, rout.im_height = left, top, W, H
                _ffi.check(lib.vips_hip_conva_gen(plan, ctypes.byref(rin), ctypes.byref(rout)))
                assert np.array_equal(out.numpy(), want[top:top + h, left:left + w]), (dtype, left, top)
            # convasep: horizontal pass over a band of rows, vertical pass out of that band
            top, h = 20, 25
            y0, y1 = max(top - 14, 0), min(top + h + 14, H)
            band = Image.new_from_array(np.ascontiguousarray(src[y0:y1]))
            rin = band.region()
            rin.top, rin.im_height = y0, H
            mid = Image.new_from_array(np.zeros((y1 - y0, W, 2), src.dtype))
            rmid = mid.region()
            rmid.top, rmid.im_height = y0, H
            _ffi.check(lib.vips_hip_convasep_gen(sep, ctypes.byref(rin), ctypes.byref(rmid), 0))
            out = Image.new_from_array(np.zeros((h, W, 2), src.dtype))
            rout = out.region()
            rout.top, rout.im_height = top, H
            _ffi.check(lib.vips_hip_convasep_gen(sep, ctypes.byref(rmid), ctypes.byref(rout), 1))
            assert np.array_equal(out.numpy(), want_sep[top:top + h]), dtype
            # too-small window
            win = Image.new_from_array(np.ascontiguousarray(src[15:45, 20:60]))
            rin = win.region()
            rin.left, rin.top, rin.im_width, rin.im_height = 20, 15, W, H
            out = Image.new_from_array(np.zeros((30, 40, 2), src.dtype))
            rout = out.region()
            rout.left, rout.top, rout.im_width, rout.im_height = 20, 15, W, H
            lib.vips_hip_error_clear()
            assert lib.vips_hip_conva_gen(plan, ctypes.byref(rin), ctypes.byref(rout)) == -1
            assert "input region too small" in _ffi.error_buffer()
            lib.vips_hip_error_clear()
        finally:
            lib.vips_hip_conva_free(plan)
            lib.vips_hip_conva_free(sep)


CAST_FORMATS = [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32, np.float64]


@pytest.mark.parametrize("src_dtype", CAST_FORMATS, ids=lambda d: np.dtype(d).name)
def test_cast_quad_rows(src_dtype, monkeypatch):
    """Round 6: a plain vips_cast takes four elements per lane (cast_quad_kernel: one load, one 16-byte store for a
    float) when rows start on a group of four on both sides, and the element kernel takes the last ne % 4 of every
    row.  Every format pair on rows of 1 365 and 1 366 elements x 3 bands (the second leaves a tail of 2), against
    the port (conversion/cast.c:120-330) and against the element kernel alone."""
    for width in (455, 683):
        src = helpers.lcg_image(width, 37, 3, src_dtype, 90 + width)
        if np.issubdtype(src_dtype, np.floating):
            src = (src.astype(np.float64) * 700.0 - 40000.0).astype(src_dtype)  # (beyond every integer range, both signs)
        for dst in CAST_FORMATS:
            want = PortCC.cast(src, np.dtype(dst))
            got = Image.new_from_array(src).cast(helpers.DTYPE_FORMATS[np.dtype(dst)]).numpy()
            monkeypatch.setenv("VIPS_HIP_NO_CAST_QUAD", "1")
            old = Image.new_from_array(src).cast(helpers.DTYPE_FORMATS[np.dtype(dst)]).numpy()
            monkeypatch.delenv("VIPS_HIP_NO_CAST_QUAD")
            assert got.dtype == want.dtype and np.array_equal(got, want, equal_nan=True), (src_dtype, dst, width)
            assert np.array_equal(got, old, equal_nan=True)


@pytest.mark.parametrize("src_dtype", [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32],
                         ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("inverse", [False, True])
def test_premultiply_rgba_rows(src_dtype, inverse, monkeypatch):
    """Round 6: premultiply / unpremultiply of RGBA to float with RU rows in flight and, for 8-bit alpha, the factor
    from a 256-entry table the block makes with the per-pixel code itself (premul_rgba_kernel): against the port
    (conversion/premultiply.c:78-128, unpremultiply.c:85-186) and against the older kernel, every alpha value."""
    src = helpers.lcg_image(300, 41, 4, src_dtype, 93)
    src[0, :256, 3] = np.arange(256).astype(src_dtype)  # every 8-bit alpha (and small values of the wider formats)
    interp = "srgb"
    want = PortCC.premultiply(src, interp, uchar=False, inverse=inverse)
    im = Image.new_from_array(src, interpretation=interp)
    got = (im.unpremultiply() if inverse else im.premultiply()).numpy()
    monkeypatch.setenv("VIPS_HIP_NO_PREMUL_RGBA", "1")
    old = (im.unpremultiply() if inverse else im.premultiply()).numpy()
    assert got.dtype == np.float32 and np.array_equal(got.view(np.int32), want.view(np.int32))
    assert np.array_equal(got.view(np.int32), old.view(np.int32))


@pytest.mark.parametrize("raw", [None, 0])
@pytest.mark.parametrize("params", [dict(), dict(sigma=1.0, x1=1.0, m2=2.0)])
def test_sharpen_adaptive_large_mixed(params, raw, monkeypatch):
    """Round 6: an image of 4 Mpixels and more takes the adaptive pair by default -- the skip kernel first, the tiles
    whose sampled row is mostly outside the LUT's flat centre left on a device-side list for the all-in-LDS kernel.
    A 2 304 x 2 100 image that is smooth on the left, noise on the right and striped (a hard edge every 8 rows: tiles
    whose SAMPLED row says little about the rest) at the bottom: both kernels make tiles of one image, some 64 x 64
    tiles half by one and half by the other; against the compiled reference / the port, whole image.  The first
    judge of a tile: sharpen_survey_kernel on the raw bytes of its middle row (the default) / the skip kernel itself
    on L of three rows ($VIPS_HIP_SHARPEN_DEFER_RAW=0); the skip kernel's second rule -- a list longer than 640 pixels
    -- stands behind both (the striped tiles: horizontal neighbours equal, a third of the pixels on the list)."""
    if raw is not None:
        monkeypatch.setenv("VIPS_HIP_SHARPEN_DEFER_RAW", str(raw))
    w, h = 2304, 2100
    noise = helpers.lcg_image(w, h, 3, np.uint8, 97)
    small = helpers.lcg_image(w // 8 + 1, h // 8 + 1, 3, np.uint8, 98).astype(np.float32)
    smooth = np.kron(small, np.ones((8, 8, 1), np.float32))[:h, :w]
    # (a box blur of the blocks along x: neighbouring pixels near each other, as a resized photograph's are)
    smooth = (smooth + np.roll(smooth, 1, 1) + np.roll(smooth, 2, 1) + np.roll(smooth, 3, 1)) / 4.0
    src = smooth.astype(np.uint8)
    src[:, w // 2:] = noise[:, w // 2:]
    stripes = (np.arange(h) // 8 % 2 * 200 + 20).astype(np.uint8)
    src[3 * h // 4:, : w // 2] = stripes[3 * h // 4:, None, None]
    lib = _ffi.lib
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    try:
        got = Image.new_from_array(src, interpretation="srgb").sharpen(**params).numpy()
        report = libvips_amd.gate_report()
    finally:
        lib.vips_hip_gate_enable(0)
        lib.vips_hip_gate_reset()
    assert sorted(report) == ["sharpen_quad_u8", "sharpen_skip_u8"] + (["sharpen_survey"] if raw is None else []), report
    if helpers.have_ref():
        args = ",".join("%s=%s" % kv for kv in params.items())
        want = Ref.run("sharpen", src, args, cases.INTERP["srgb"])
    else:
        want = PortCC.sharpen(src, "srgb", **params)
    bad = np.argwhere((got != want).any(axis=-1))
    assert len(bad) == 0, (len(bad), bad[:5].tolist())
