"""GPU parity: HIP reduce/shrink through the C ABI vs the oracle.

Checker order: committed golden vectors (from the compiled reference), the plain-C
port on fresh seeded inputs, the compiled reference itself (oracle/_ref travels to
the GPU box) at BASELINE sizes.  Integer formats bit-exact; float bitwise too (the
device keeps the reference's double mul/add order), asserted as <= 1 ULP, the
tolerance north_star states.
"""
import ctypes
import math
import os

import numpy as np
import pytest

import libvips_amd
from libvips_amd import Image, _ffi
from tests import helpers
from tests.golden import cases
from tests.helpers import Port, Ref

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(helpers.GOLDEN, "resample.npz"))


@pytest.fixture(scope="module", autouse=True)
def _init():
    libvips_amd.init(0)


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b).max()


def assert_same(got, want, what=""):
    assert got.shape == want.shape, what
    assert got.dtype == want.dtype, what
    if got.dtype.kind == "f" and got.dtype.itemsize == 4:
        assert ulp_diff(got, want) <= 1, what  # tolerance: 1 ULP (north_star)
    elif got.dtype.kind == "f":
        np.testing.assert_allclose(got, want, rtol=2e-16, atol=0, err_msg=what)
    else:
        assert np.array_equal(got, want), what


def hip_call(case, src):
    fn, kw = case["call"]
    return getattr(Image.new_from_array(src), fn)(**kw).numpy()


@pytest.mark.parametrize("case", cases.RESAMPLE_CASES, ids=[c["name"] for c in cases.RESAMPLE_CASES])
def test_hip_matches_golden(case):
    src = helpers.lcg_image(case["width"], case["height"], case["bands"], case["dtype"], case["seed"])
    assert_same(hip_call(case, src), GOLD[case["name"]], case["name"])


@pytest.mark.parametrize("dtype", [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32])
@pytest.mark.parametrize("bands", [1, 2, 3, 4, 5])
def test_reduce_formats_vs_port(dtype, bands):
    src = helpers.lcg_image(157, 121, bands, dtype, 41)
    im = Image.new_from_array(src)
    for kernel, h, v in (("lanczos3", 2.3, 3.1), ("cubic", 1.1, 1.999), ("linear", 4.0, 2.0),
                         ("mks2021", 1.5, 1.5), ("nearest", 2.0, 3.0)):
        assert_same(im.reduce(h, v, kernel=kernel).numpy(), Port.reduce(src, h, v, kernel),
                    "%s %s %s" % (kernel, h, v))


@pytest.mark.parametrize("dtype", [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32,
                                   np.float32, np.float64])
def test_shrink_formats_vs_port(dtype):
    for bands in (1, 3, 4):
        src = helpers.lcg_image(203, 157, bands, dtype, 42)
        im = Image.new_from_array(src)
        for h, v, ceil in ((2, 2, False), (3, 5, True), (4, 4, False), (7, 9, True), (8, 1, False)):
            assert_same(im.shrink(h, v, ceil=ceil).numpy(), Port.shrink(src, h, v, ceil),
                        "%s %s %s" % (h, v, ceil))


def test_reduce_ragged_and_tiny():
    # edge cases: 1-pixel outputs, widths not a multiple of anything, heavy clamping
    for (w, h, hs, vs) in ((1, 1, 1.0, 1.0), (3, 2, 2.0, 2.0), (17, 5, 16.9, 4.9), (1000, 3, 37.3, 1.2),
                           (5, 700, 1.0, 49.5)):
        src = helpers.lcg_image(w, h, 3, np.uint8, 43)
        assert_same(Image.new_from_array(src).reduce(hs, vs).numpy(), Port.reduce(src, hs, vs),
                    str((w, h, hs, vs)))


def test_tile_seeding_matches_reference_tiles():
    src = helpers.lcg_image(33, 1500, 1, np.uint8, 32)
    got = Image.new_from_array(src).reducev(2.7182818).numpy()
    assert_same(got, Port.reducev(src, 2.7182818, "lanczos3", tile=16))
    if helpers.have_ref():
        assert_same(got, Ref.run("reducev", src, "vshrink=2.7182818,kernel=lanczos3"))


def test_region_generate_contract():
    """Drive the region-level gens the way a libvips generate would: arbitrary output
    rects, input windows that only just cover the needed rows/columns."""
    lib = _ffi.lib
    src = helpers.lcg_image(300, 260, 4, np.uint8, 44)
    h, w, b = src.shape
    vshrink = 3.3
    oh = int(h / vshrink + 0.5)
    want = Port.reducev(src, vshrink, "lanczos3", tile=0)  # seeded once at row 0
    r = lib.vips_hip_reduce_new(5, vshrink, h, oh, math.nan)
    assert r
    full = Image.new_from_array(src)
    try:
        # (1) whole image, one seed
        out = Image.new_from_array(np.zeros((oh, w, b), np.uint8))
        ri, ro = full.region(), out.region()
        _ffi.check(lib.vips_hip_reducev_gen(r, ctypes.byref(ri), ctypes.byref(ro)))
        assert np.array_equal(out.numpy(), want)
        # (2) a sub-rect, seeded at its own top: equals the port seeded the same way
        top, height, left, width = 17, 23, 40, 100
        t0, tn = ctypes.c_int(), ctypes.c_int()
        lib.vips_hip_reducev_need(r, top, height, ctypes.byref(t0), ctypes.byref(tn))
        win = np.ascontiguousarray(src[t0.value:t0.value + tn.value, left:left + width])
        dwin = Image.new_from_array(win)
        rin = dwin.region()
        rin.left, rin.top, rin.im_width, rin.im_height = left, t0.value, w, h
        dout = Image.new_from_array(np.zeros((height, width, b), np.uint8))
        rout = dout.region()
        rout.left, rout.top, rout.im_width, rout.im_height = left, top, w, oh
        _ffi.check(lib.vips_hip_reducev_gen(r, ctypes.byref(rin), ctypes.byref(rout)))
        # reference semantics: Y is seeded at r->top (reducev.cpp:548).  A port run whose
        # strips are `top` rows high seeds its second strip at exactly that row.
        exp = Port.reducev(src, vshrink, "lanczos3", tile=top)[top:top + height, left:left + width]
        # rows top..2*top-1 come from the strip seeded at `top`
        n = min(height, top)
        assert np.array_equal(dout.numpy()[:n], exp[:n])
        # (3) a window that is too small is rejected loudly
        rin.height -= 1
        lib.vips_hip_error_clear()
        assert lib.vips_hip_reducev_gen(r, ctypes.byref(rin), ctypes.byref(rout)) == -1
        assert "input region too small" in _ffi.error_buffer()
        lib.vips_hip_error_clear()
    finally:
        lib.vips_hip_reduce_free(r)


def test_error_behaviour():
    im = Image.new_from_array(helpers.lcg_image(20, 20, 3, np.uint8, 45))
    with pytest.raises(libvips_amd.VipsHipError, match="reduce factor should be >= 1.0"):
        im.reduceh(0.5)
    with pytest.raises(libvips_amd.VipsHipError, match="shrink factors should be >= 1"):
        im.shrinkh(0)
    with pytest.raises(libvips_amd.VipsHipError, match="reduce gap should be >= 1.0"):
        im.reducev(2.0, gap=0.5)
    with pytest.raises(libvips_amd.VipsHipError, match="scale must be > 0"):
        im.resize(-0.5)


@pytest.mark.skipif(not helpers.have_ref(), reason="needs oracle/_ref: the x87 long double sums are the reference's")
@pytest.mark.parametrize("op,args", [
    ("reduceh", "hshrink=3.1"), ("reducev", "vshrink=2.5,kernel=cubic"), ("reduce", "hshrink=8,vshrink=8"),
    ("reduce", "hshrink=1.7,vshrink=4.3,kernel=lanczos2"), ("reduce", "hshrink=2,vshrink=2,kernel=linear"),
    ("reduce", "hshrink=3.3,vshrink=1.2,kernel=mks2021"), ("reducev", "vshrink=5,kernel=nearest"),
    ("resize", "scale=0.37"), ("resize", "scale=0.125,kernel=mitchell"),
    # upsizing: the no-table bicubic (bicubic.cpp:419-480: Catmull-Rom coefficients of the exact
    # offsets, all in double), bilinear, nearest
    ("resize", "scale=2.2"), ("resize", "scale=2.5,kernel=cubic"), ("resize", "scale=1.7,kernel=linear"),
    ("resize", "scale=3,kernel=nearest"),
])
@pytest.mark.parametrize("bands", [1, 3])
def test_reduce_double_vs_reference(op, args, bands):
    """Double images: the reference's "ultra-high-quality" path (reduceh.cpp:196-213,
    reducev.cpp:497-515): no coefficient table, a LONG DOUBLE mask per output position and a long
    double sum.  The device does that arithmetic -- x87 extended: 64 bits of mantissa, one rounding
    per operation -- with integers (x80.h); the masks are made on the host in long double.  Bit for
    bit against the compiled reference, values over a wide range of magnitudes and signs."""
    rng = np.random.default_rng(5)
    w, h = 403, 297
    src = (rng.standard_normal((h, w, bands)) * np.exp2(rng.integers(-20, 20, size=(h, w, bands)))).astype(np.float64)
    src[3, 5] = 0.0
    src[7, 11] = 5e-324          # a denormal
    want = Ref.run(op, src, args)
    im = Image.new_from_array(src)
    kw = dict(kv.split("=") for kv in args.split(","))
    kernel = kw.pop("kernel", "lanczos3")
    kw = {k: float(v) for k, v in kw.items()}
    if op == "reduceh":
        got = im.reduceh(kw["hshrink"], kernel=kernel)
    elif op == "reducev":
        got = im.reducev(kw["vshrink"], kernel=kernel)
    elif op == "reduce":
        got = im.reduce(kw["hshrink"], kw["vshrink"], kernel=kernel)
    else:
        got = im.resize(kw["scale"], kernel=kernel)
    got = got.numpy()
    assert got.dtype == np.float64 and got.shape == want.shape
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_c2_quarter_size_vs_reference_checksum():
    # SURVEY.md 8(c) golden: 4096^2 x4 LCG -> 512^2, checksum 16793779256
    src = helpers.lcg_image(4096, 4096, 4, np.uint8, 12345)
    got = Image.new_from_array(src).reduce(8, 8, kernel="lanczos3").numpy()
    assert got.shape == (512, 512, 4)
    assert helpers.checksum(got) == 16793779256


def test_c2_full_size():
    """BASELINE config 2: 16384^2 RGBA -> 2048^2.  Bit-exact against the compiled
    reference when it travelled with the snapshot, plus size-independent properties."""
    import torch

    n = 16384
    src = helpers.lcg_image(n, n, 4, np.uint8, 12345)
    t = torch.from_numpy(src).cuda()
    got = Image.new_from_tensor(t).reduce(8, 8, kernel="lanczos3").numpy()
    assert got.shape == (2048, 2048, 4)
    # property: the top-left 512^2 block depends only on the top-left 4096+ pixels; the
    # image rows of the LCG stream differ from the 4096^2 case, so check against a port
    # run on a crop instead (interior rows only: away from the crop's clamped edges)
    crop = np.ascontiguousarray(src[:1024, :1024])
    want = Port.reduce(crop, 8, 8, "lanczos3")
    assert np.array_equal(got[:120, :120], want[:120, :120])
    # property: average preserved (test_resample.py:83-92)
    assert abs(got.mean() - src[::64, ::64].mean()) < 2
    if helpers.have_ref():
        want_full = Ref.run("reduce", src, "hshrink=8,vshrink=8,kernel=lanczos3")
        assert np.array_equal(got, want_full)
    # constant image stays constant (test_resample.py:94-103), full size
    t.fill_(201)
    const = Image.new_from_tensor(t).reduce(8, 8, kernel="lanczos3").numpy()
    assert const.min() == 201 and const.max() == 201


@pytest.mark.parametrize("shrink", [2, 4, 8])
@pytest.mark.parametrize("size", [(1203, 917), (2048, 1024), (640, 8), (96, 2000)])
def test_fused_reduce_rgba(shrink, size):
    """The fused reducev+reduceh kernel (reduce_u8.hip): even integer shrink, RGBA uchar;
    sizes that are / are not multiples of the shrink (phase 0 and constant non-zero
    phase), several tiles wide and tall, all four edges clamped."""
    from libvips_amd import lib

    w, h = size
    src = helpers.lcg_image(w, h, 4, np.uint8, 46)
    im = Image.new_from_array(src)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    try:
        got = im.reduce(shrink, shrink, kernel="lanczos3").numpy()
        report = libvips_amd.gate_report()
    finally:
        lib.vips_hip_gate_enable(0)
        lib.vips_hip_gate_reset()
    assert any(k.startswith("reduce_fused_u8") for k in report), report
    assert_same(got, Port.reduce(src, shrink, shrink, "lanczos3"), str((shrink, size)))


@pytest.mark.parametrize("two_kernels", [False, True])
@pytest.mark.parametrize("size", [(512, 264), (1024, 776), (2048, 1100), (1536, 2056), (4096, 8), (512, 4100)])
def test_fused_reduce_exchange(size, two_kernels, monkeypatch):
    """Round 6: the matrix-core reduce WITHOUT the tiles' horizontal halo (reduce_fused_u8x4_mfma_x: a tile is 512
    aligned columns and makes all 64 of its outputs, the six outputs that straddle a tile boundary as partial sums by
    both tiles, added and rounded by whichever of the two arrives at the boundary LAST -- or, two_kernels, by
    reduce_fused_edges behind it) -- and that TWICE in a row, the arrival counters live on) -- images one tile wide (both sides the image's edge:
    vips_embed COPY through the replicated edge column), several tiles either way, heights the tile rows do not
    divide (a last row of tiles walked bottom-up); against the port and against the kernel with halos.  ($VIPS_HIP_FUSED_EXCH=1: by default only launches of 384 tiles and more take it.)"""
    w, h = size
    src = helpers.lcg_image(w, h, 4, np.uint8, 48)
    im = Image.new_from_array(src)
    monkeypatch.setenv("VIPS_HIP_FUSED_EXCH", "0")
    old = im.reduce(8, 8, kernel="lanczos3").numpy()
    monkeypatch.setenv("VIPS_HIP_FUSED_EXCH", "1")
    if two_kernels:
        monkeypatch.setenv("VIPS_HIP_FUSED_DEBUG", "32")
    lib = libvips_amd.lib
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    try:
        got = im.reduce(8, 8, kernel="lanczos3").numpy()
        again = im.reduce(8, 8, kernel="lanczos3").numpy()
        report = libvips_amd.gate_report()
    finally:
        lib.vips_hip_gate_enable(0)
        lib.vips_hip_gate_reset()
    assert sorted(report) == (["reduce_fused_edges"] if two_kernels else []) + ["reduce_fused_u8_mfma_x"], report
    assert_same(got, Port.reduce(src, 8, 8, "lanczos3"), str(size))
    assert np.array_equal(got, old) and np.array_equal(again, old)


@pytest.mark.parametrize("oht,deep", [(0, None), (8, 0), (128, 1), (0, 0)])
@pytest.mark.parametrize(
    "size", [(640, 264), (680, 72), (1280, 1100), (2048, 1029), (3000, 520), (96, 2056), (24, 16), (5124, 301), (1282, 400)]
)
def test_fused_reduce_rgb(size, oht, deep, monkeypatch):
    """Round 6: vips_reduce by 8 on THREE interleaved bands in one kernel (reduce_fused_u8x3_mfma: the vertical pass
    on the row's bytes, the T rows in LDS as they lie in memory, the horizontal walk picking its band's bytes out
    of 24-byte groups) -- images one tile wide (both of the image's edges in the same T row), several tiles either
    way, a last tile of fewer than 80 outputs, heights the tile rows do not divide (a last row of tiles walked
    bottom-up), sizes that are not multiples of 8 (a constant non-zero phase, seven groups of taps), tiles of 8 and
    of 128 rows, both depths of the row ring; against the port and against reducev + reduceh.  A width that is not a multiple of 8 is not the
    kernel's case (the first tap moves off a dword boundary, or the last dword straddles the image's edge)."""
    w, h = size
    src = helpers.lcg_image(w, h, 3, np.uint8, 49)
    im = Image.new_from_array(src)
    monkeypatch.setenv("VIPS_HIP_NO_FUSED3", "1")
    two = im.reduce(8, 8, kernel="lanczos3").numpy()
    monkeypatch.delenv("VIPS_HIP_NO_FUSED3")
    if oht:
        monkeypatch.setenv("VIPS_HIP_FUSED3_OHT", str(oht))
    if deep is not None:  # (two row groups in flight a lane, what launches of under 768 tiles take / one)
        monkeypatch.setenv("VIPS_HIP_FUSED3_DEEP", str(deep))
    lib = libvips_amd.lib
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    try:
        got = im.reduce(8, 8, kernel="lanczos3").numpy()
        report = libvips_amd.gate_report()
    finally:
        lib.vips_hip_gate_enable(0)
        lib.vips_hip_gate_reset()
    if w % 8 == 0:
        assert sorted(report) == ["reduce_fused_u8x3_mfma"], report
    else:
        assert "reduce_fused_u8x3_mfma" not in report, report
    assert_same(got, Port.reduce(src, 8, 8, "lanczos3"), str(size))
    assert np.array_equal(got, two)


@pytest.mark.parametrize("kernel", ["lanczos3"])
@pytest.mark.parametrize("size", [(4099, 3001), (2048, 1024), (1000, 8), (96, 2600), (9000, 700), (8192, 8197)])
@pytest.mark.parametrize("align", [0, 1])
def test_fused_reduce_mfma_variants(align, size, kernel, monkeypatch):
    """Both tile layouts of the matrix-core kernel with halos on the same inputs: line-aligned tiles (lanes
    start on the 128-byte line holding the first tap; the default when base and stride allow)
    and tiles that start at the first tap (VIPS_HIP_FUSED_ALIGN=0: what windows with an odd
    base get), shrink 8, sizes with partial tiles on every side, several tiles in both
    directions, 6 and 7 tap groups (phase 0 and a constant non-zero phase), all edges clamped.
    (Rounds 2-5 also ran the kernel's retired A/B forms here -- 512 threads, the edge fix-up at the loads, plain
    loads; round 6 deleted them.)"""
    from libvips_amd import lib

    w, h = size
    src = helpers.lcg_image(w, h, 4, np.uint8, 47)
    monkeypatch.setenv("VIPS_HIP_FUSED_ALIGN", str(align))
    monkeypatch.setenv("VIPS_HIP_FUSED_EXCH", "0")  # (the kernel WITH halos; the other: test_fused_reduce_exchange)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    try:
        got = Image.new_from_array(src).reduce(8, 8, kernel=kernel).numpy()
        report = libvips_amd.gate_report()
    finally:
        lib.vips_hip_gate_enable(0)
        lib.vips_hip_gate_reset()
    assert list(report) == ["reduce_fused_u8_mfma"], report
    assert_same(got, Port.reduce(src, 8, 8, kernel), str((align, size, kernel)))


@pytest.mark.parametrize("align", [0, 1])
def test_fused_reduce_region_windows(align):
    """vips_hip_reduce_gen the way a strip owner (one GPU of several, libvips_amd/sharding.py)
    calls it: an output sub-rect and an input window that only just covers the rows and
    columns vips_hip_reduce{v,h}_need report, at image edges and in the middle; must equal
    the same rect of the whole-image result."""
    lib = _ffi.lib
    w, h = 2400, 1608
    src = helpers.lcg_image(w, h, 4, np.uint8, 48)
    full = Image.new_from_array(src).reduce(8, 8, kernel="lanczos3").numpy()
    oh, ow = full.shape[:2]
    rv = _ffi.check_handle(lib.vips_hip_reduce_new(5, 8.0, h, oh, math.nan))
    rh = _ffi.check_handle(lib.vips_hip_reduce_new(5, 8.0, w, ow, math.nan))
    os.environ["VIPS_HIP_FUSED_ALIGN"] = str(align)
    try:
        for (left, top, width, height) in ((0, 0, ow, 37), (0, 37, ow, oh - 37), (10, 50, 200, 100),
                                           (ow - 61, oh - 40, 61, 40), (0, 100, 59, 1)):
            t0, tn, l0, ln = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            lib.vips_hip_reducev_need(rv, top, height, ctypes.byref(t0), ctypes.byref(tn))
            lib.vips_hip_reduceh_need(rh, left, width, ctypes.byref(l0), ctypes.byref(ln))
            win = np.ascontiguousarray(src[t0.value:t0.value + tn.value, l0.value:l0.value + ln.value])
            dwin = Image.new_from_array(win)
            rin = dwin.region()
            rin.left, rin.top, rin.im_width, rin.im_height = l0.value, t0.value, w, h
            dout = Image.new_from_array(np.zeros((height, width, 4), np.uint8))
            rout = dout.region()
            rout.left, rout.top, rout.im_width, rout.im_height = left, top, ow, oh
            r = lib.vips_hip_reduce_gen(rv, rh, ctypes.byref(rin), ctypes.byref(rout))
            assert r == 0, (r, _ffi.error_buffer())
            assert np.array_equal(dout.numpy(), full[top:top + height, left:left + width]), (left, top, width, height)
    finally:
        del os.environ["VIPS_HIP_FUSED_ALIGN"]
        lib.vips_hip_reduce_free(rv)
        lib.vips_hip_reduce_free(rh)


def test_fused_reduce_rgb_region_windows():
    """The three-band kernel on vips_hip_reduce_gen's sub-rects (a strip owner's call, the module's strips): an
    input window that only just covers what the rect needs -- its first column is the rect's first tap, so the
    window's byte 0 is a tile's -- at the image's left, right, top and bottom edges and in the middle; must equal
    the same rect of the whole-image result.  A return of 1 (not this kernel's case: the caller runs reducev, then
    reduceh) is allowed, but not for every rect."""
    lib = _ffi.lib
    w, h = 2400, 1608
    src = helpers.lcg_image(w, h, 3, np.uint8, 50)
    full = Image.new_from_array(src).reduce(8, 8, kernel="lanczos3").numpy()
    assert np.array_equal(full, Port.reduce(src, 8, 8, "lanczos3"))
    oh, ow = full.shape[:2]
    rv = _ffi.check_handle(lib.vips_hip_reduce_new(5, 8.0, h, oh, math.nan))
    rh = _ffi.check_handle(lib.vips_hip_reduce_new(5, 8.0, w, ow, math.nan))
    taken = 0
    try:
        for (left, top, width, height, widen) in ((0, 0, ow, 37, 0), (0, 37, ow, oh - 37, 0), (10, 50, 200, 100, 1),
                                                  (10, 50, 200, 100, 0), (ow - 61, oh - 40, 61, 40, 0), (0, 100, 59, 1, 1),
                                                  (85, 3, 163, 70, 1)):
            t0, tn, l0, ln = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            lib.vips_hip_reducev_need(rv, top, height, ctypes.byref(t0), ctypes.byref(tn))
            lib.vips_hip_reduceh_need(rh, left, width, ctypes.byref(l0), ctypes.byref(ln))
            if widen:  # (a window of whole dwords: the kernel's case)
                ln.value = min((ln.value + 3) & ~3, w - l0.value)
            win = np.ascontiguousarray(src[t0.value:t0.value + tn.value, l0.value:l0.value + ln.value])
            dwin = Image.new_from_array(win)
            rin = dwin.region()
            rin.left, rin.top, rin.im_width, rin.im_height = l0.value, t0.value, w, h
            dout = Image.new_from_array(np.full((height, width, 3), 77, np.uint8))
            rout = dout.region()
            rout.left, rout.top, rout.im_width, rout.im_height = left, top, ow, oh
            lib.vips_hip_gate_reset()
            lib.vips_hip_gate_enable(1)
            try:
                r = lib.vips_hip_reduce_gen(rv, rh, ctypes.byref(rin), ctypes.byref(rout))
                report = libvips_amd.gate_report()
            finally:
                lib.vips_hip_gate_enable(0)
                lib.vips_hip_gate_reset()
            assert r in (0, 1), (r, _ffi.error_buffer())
            if r == 0:
                assert sorted(report) == ["reduce_fused_u8x3_mfma"], report
                taken += 1
                assert np.array_equal(dout.numpy(), full[top:top + height, left:left + width]), (left, top, width, height)
    finally:
        lib.vips_hip_reduce_free(rv)
        lib.vips_hip_reduce_free(rh)
    assert taken >= 5, taken


@pytest.mark.parametrize("dtype", [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32])
@pytest.mark.parametrize("kernel", ["nearest", "linear", "cubic", "lanczos3"])
def test_resize_upsizing_vs_port(dtype, kernel):
    """The upsizing half of vips_resize (vips_affine + nearest / bilinear / bicubic, vips_zoom):
    every format, up / up, up / down mixes, integral and fractional scales, rows wider than one
    block; bit-exact against the port (itself pinned on the compiled reference)."""
    for (w, h, b, hs, vs) in ((37, 29, 3, 2.3, 2.3), (300, 20, 1, 3.0, 2.0), (51, 40, 4, 1.5, 0.7),
                              (40, 33, 2, 0.6, 1.9), (130, 16, 3, 4.0, 4.0), (200, 9, 3, 1.01, 7.3)):
        if kernel == "nearest" and (hs < 1 or vs < 1):
            continue
        src = helpers.lcg_image(w, h, b, dtype, 78)
        got = Image.new_from_array(src).resize(hs, vscale=vs, kernel=kernel).numpy()
        want = Port.resize(src, hs, vs, kernel=kernel)
        assert got.shape == want.shape, (kernel, w, h, hs, vs)
        if dtype == np.float32:
            assert np.array_equal(got.view(np.int32), want.view(np.int32)), (kernel, w, h, hs, vs)
        else:
            assert np.array_equal(got, want), (kernel, w, h, hs, vs)


def test_upsize_region_rects_and_reference():
    """vips_hip_upsize_gen on output sub-rects with an input window that just covers them
    (a strip owner's call), and a 2048-wide enlargement against the compiled reference."""
    lib = _ffi.lib
    src = helpers.lcg_image(310, 207, 3, np.uint8, 79)
    hs, vs = 2.7, 1.9
    idx, idy = 0.5 * (1 - 1 / hs), 0.5 * (1 - 1 / vs)
    full = Port.affine_scale(src, hs, vs, idx, idy, "bicubic")
    oh, ow = full.shape[:2]
    h, w = src.shape[:2]
    for (left, top, width, height) in ((0, 0, ow, 50), (0, 50, ow, oh - 50), (100, 70, 333, 129), (ow - 40, oh - 9, 40, 9)):
        x0 = max(int(left / hs) - 4, 0)
        x1 = min(int((left + width) / hs) + 5, w)
        y0 = max(int(top / vs) - 4, 0)
        y1 = min(int((top + height) / vs) + 5, h)
        win = np.ascontiguousarray(src[y0:y1, x0:x1])
        dwin = Image.new_from_array(win)
        rin = dwin.region()
        rin.left, rin.top, rin.im_width, rin.im_height = x0, y0, w, h
        dout = Image.new_from_array(np.zeros((height, width, 3), np.uint8))
        rout = dout.region()
        rout.left, rout.top, rout.im_width, rout.im_height = left, top, ow, oh
        r = lib.vips_hip_upsize_gen(ctypes.byref(rin), ctypes.byref(rout), hs, vs, idx, idy, 2, 0)
        assert r == 0, _ffi.error_buffer()
        assert np.array_equal(dout.numpy(), full[top:top + height, left:left + width]), (left, top)
    # a window that is too small is refused loudly
    rin.width -= 8
    lib.vips_hip_error_clear()
    assert lib.vips_hip_upsize_gen(ctypes.byref(rin), ctypes.byref(rout), hs, vs, idx, idy, 2, 0) == -1
    assert "input region too small" in _ffi.error_buffer()
    lib.vips_hip_error_clear()
    if helpers.have_ref():
        big = helpers.lcg_image(1024, 700, 4, np.uint8, 80)
        got = Image.new_from_array(big).resize(2.0, kernel="lanczos3").numpy()
        assert np.array_equal(got, Ref.run("resize", big, "scale=2,kernel=lanczos3"))


@pytest.mark.parametrize("ring", [None, 1, 4])
@pytest.mark.parametrize("bands", [1, 2, 3, 4])
def test_reducev_mfma_any_bands(bands, ring, monkeypatch):
    """reducev_u8_mfma: the vertical pass on the matrix cores for uchar images of any band
    count (integer-8 shrink, one phase, rows of whole 8-byte columns), alone and as the first
    half of vips_reduce on RGB; several tiles wide and tall, ragged heights, clamped edges; one and four
    row groups in flight a lane (round 6: launches of at most a tile a CU take four)."""
    from libvips_amd import lib

    if ring:
        monkeypatch.setenv("VIPS_HIP_REDUCEV8_NB", str(ring))
    monkeypatch.setenv("VIPS_HIP_NO_FUSED3", "1")  # (the one-kernel form of RGB: test_fused_reduce_rgb)

    for (w, h) in ((2048, 1603), (8 * 40 // bands * bands if bands != 3 else 640, 4099), (4096, 200)):
        if (w * bands) % 8:
            continue
        src = helpers.lcg_image(w, h, bands, np.uint8, 49)
        im = Image.new_from_array(src)
        lib.vips_hip_gate_reset()
        lib.vips_hip_gate_enable(1)
        try:
            got_v = im.reducev(8.0, kernel="lanczos3").numpy()
            got = im.reduce(8.0, 8.0, kernel="lanczos3").numpy()
            report = libvips_amd.gate_report()
        finally:
            lib.vips_hip_gate_enable(0)
            lib.vips_hip_gate_reset()
        assert "reducev_u8_mfma" in report or bands == 4, report
        assert_same(got_v, Port.reducev(src, 8.0, "lanczos3"), str((bands, w, h)))
        assert_same(got, Port.reduce(src, 8.0, 8.0, "lanczos3"), str((bands, w, h)))


@pytest.mark.parametrize("bands", [1, 3, 4])
@pytest.mark.parametrize("size", [(2048, 1536), (1000, 1203), (4099, 2051), (256, 64), (8192, 520)])
@pytest.mark.parametrize("scale,kernel", [(0.125, "lanczos3"), (0.1, "lanczos3"), (0.125, "cubic"), (1.0 / 6.0, "lanczos3")])
def test_resize_uchar_gap_shrinks(bands, size, scale, kernel):
    """vips_resize on uchar with its default gap (an integer box shrink in front of the residual
    reduce on both axes): heights that are / are not multiples of the shrink, rows of 16-byte
    and of odd length, 9 and 13 taps; bit-exact against the port."""
    w, h = size
    src = helpers.lcg_image(w, h, bands, np.uint8, 51)
    got = Image.new_from_array(src).resize(scale, kernel=kernel).numpy()
    assert_same(got, Port.resize(src, scale, kernel=kernel), str((bands, size, scale, kernel)))


@pytest.mark.parametrize("bands", [1, 2, 3, 4])
@pytest.mark.parametrize("size,scale,vscale", [
    ((2048, 1536), 0.125, None), ((1531, 1203), 0.3, None), ((4099, 1051), 0.07, 0.19),
    ((640, 480), 0.45, 0.26), ((3000, 700), 0.021, 0.3), ((1024, 1024), 1.0 / 3.0, 0.125),
    ((517, 2049), 0.26, 0.031), ((200, 100), 0.2, None), ((8192, 300), 0.0125, 0.4)])
def test_resize_tail_fused(bands, size, scale, vscale, monkeypatch):
    """The fused tail of vips_resize on uchar (reducev -> shrinkh -> reduceh in one kernel,
    resize_tail.hip): ran, and bit-exact against the port and against the separate kernels;
    scales with and without box pre-shrinks on either axis, irregular tap positions, all
    band counts, narrow outputs."""
    w, h = size
    src = helpers.lcg_image(w, h, bands, np.uint8, 77)
    kw = {} if vscale is None else {"vscale": vscale}
    im = Image.new_from_array(src)
    monkeypatch.setenv("VIPS_HIP_NO_RESIZE_STREAM", "1")  # the whole-chain kernel takes 1 / (2 k) scales first
    monkeypatch.setenv("VIPS_HIP_NO_RESIZE_BAND", "1")  # ... and the matrix-core chain the images of 8 MB and more
    libvips_amd.lib.vips_hip_gate_reset()
    libvips_amd.lib.vips_hip_gate_enable(1)
    try:
        got = im.resize(scale, **kw).numpy()
        report = libvips_amd.gate_report()
    finally:
        libvips_amd.lib.vips_hip_gate_enable(0)
        libvips_amd.lib.vips_hip_gate_reset()
    assert "resize_tail_u8" in report, report
    assert not any(k.startswith("reduceh") or k.startswith("shrinkh") for k in report), report
    assert_same(got, Port.resize(src, scale, **kw), str((bands, size, scale, vscale)))
    monkeypatch.setenv("VIPS_HIP_NO_RESIZE_TAIL", "1")
    assert np.array_equal(got, im.resize(scale, **kw).numpy())


@pytest.mark.parametrize("bands", [1, 2, 3, 4])
@pytest.mark.parametrize("size,scale,vscale", [
    ((2048, 1536), 0.125, None), ((4104, 1203), 0.125, None), ((2048, 1537), 0.25, None),
    ((2052, 1202), 1.0 / 6.0, None), ((2560, 1004), 0.1, None), ((3072, 1205), 1.0 / 12.0, None),
    ((4096, 1607), 1.0 / 16.0, None), ((2048, 1001), 0.5, None), ((4096, 900), 0.125, 0.25),
    ((8192, 2563), 0.125, None), ((2048, 40), 0.25, None), ((640, 481), 0.25, None), ((256, 64), 0.125, None),
    ((100, 1000), 0.5, 0.1)])
def test_resize_stream_fused(bands, size, scale, vscale, monkeypatch):
    """vips_resize by 1 / (2 k) on uchar in ONE kernel (resize_stream.hip: shrinkv -> reducev ->
    shrinkh -> reduceh streaming down column strips): ran alone, and bit-exact against the port
    and against the separate kernels; box shrinks 1..8 on either axis, heights the box does not
    divide (ceil mode), several strips and segments, rows shorter than a strip's 2 KB span, every
    band count."""
    w, h = size
    src = helpers.lcg_image(w, h, bands, np.uint8, 78)
    kw = {} if vscale is None else {"vscale": vscale}
    im = Image.new_from_array(src)
    monkeypatch.setenv("VIPS_HIP_STREAM_BLOCKS", "4096")  # short segments: several per image
    libvips_amd.lib.vips_hip_gate_reset()
    libvips_amd.lib.vips_hip_gate_enable(1)
    try:
        got = im.resize(scale, **kw).numpy()
        report = libvips_amd.gate_report()
    finally:
        libvips_amd.lib.vips_hip_gate_enable(0)
        libvips_amd.lib.vips_hip_gate_reset()
    assert list(report) == ["resize_stream_u8"], report
    assert_same(got, Port.resize(src, scale, **kw), str((bands, size, scale, vscale)))
    monkeypatch.delenv("VIPS_HIP_STREAM_BLOCKS")
    assert np.array_equal(got, im.resize(scale, **kw).numpy())  # one tall segment per strip
    monkeypatch.setenv("VIPS_HIP_NO_RESIZE_STREAM", "1")
    assert np.array_equal(got, im.resize(scale, **kw).numpy())


@pytest.mark.parametrize("burst,window,hf", [(0, 13, 1), (1, 13, 1), (2, 4, 1), (14, 40, 1), (5, 9, 1), (14, 13, 0)])
@pytest.mark.parametrize("bands,size,scale,vscale", [
    (3, (8192, 2563), 0.125, None), (3, (4104, 1203), 0.125, None), (4, (4096, 1607), 1.0 / 16.0, None),
    (4, (2052, 1202), 0.25, 0.125), (1, (4096, 900), 0.125, 0.25), (2, (2048, 1537), 0.25, None),
    (4, (2050, 801), 0.125, None)])
def test_resize_stream_output_stage(bands, size, scale, vscale, burst, window, hf, monkeypatch):
    """The output stage of resize_stream's dword horizontal pass: rows wait in LDS (`burst` slabs,
    0 = written as they are made) and leave when the chip-wide clock crosses a multiple of
    2^window ticks -- every stage size and window length, also windows shorter than a slab (a
    write after every slab) and longer than a launch (only the full stage and the segment's end
    write), gives the byte-by-byte pass's pixels and the port's: when a row leaves cannot change
    what is in it.  The last case (2050 wide: the box does not divide the row) takes the byte form."""
    w, h = size
    src = helpers.lcg_image(w, h, bands, np.uint8, 83)
    kw = {} if vscale is None else {"vscale": vscale}
    im = Image.new_from_array(src)
    want = Port.resize(src, scale, **kw)
    monkeypatch.setenv("VIPS_HIP_STREAM_BURST", str(burst))
    monkeypatch.setenv("VIPS_HIP_STREAM_WINDOW", str(window))
    monkeypatch.setenv("VIPS_HIP_STREAM_HF", str(hf))
    for blocks in ("4096", None):  # short segments, then one tall segment per strip
        if blocks:
            monkeypatch.setenv("VIPS_HIP_STREAM_BLOCKS", blocks)
        else:
            monkeypatch.delenv("VIPS_HIP_STREAM_BLOCKS")
        libvips_amd.lib.vips_hip_gate_reset()
        libvips_amd.lib.vips_hip_gate_enable(1)
        try:
            got = im.resize(scale, **kw).numpy()
            report = libvips_amd.gate_report()
        finally:
            libvips_amd.lib.vips_hip_gate_enable(0)
            libvips_amd.lib.vips_hip_gate_reset()
        assert list(report) == ["resize_stream_u8"], report
        assert_same(got, want, str((bands, size, scale, vscale, burst, window, hf, blocks)))


@pytest.mark.parametrize("kernel", ["linear", "cubic", "mitchell", "lanczos2", "mks2013"])
@pytest.mark.parametrize("bands,size,scale", [(3, (2048, 1203), 0.125), (1, (4104, 777), 0.25), (4, (2560, 1004), 0.1)])
def test_resize_stream_other_kernels(kernel, bands, size, scale, monkeypatch):
    """The one-kernel chain with 5 (linear), 9 (cubic, mitchell, lanczos2) and 13 (mks2013)
    vertical taps: 3, 5 and 7 output rows in flight per column."""
    w, h = size
    src = helpers.lcg_image(w, h, bands, np.uint8, 79)
    im = Image.new_from_array(src)
    monkeypatch.setenv("VIPS_HIP_STREAM_BLOCKS", "4096")
    libvips_amd.lib.vips_hip_gate_reset()
    libvips_amd.lib.vips_hip_gate_enable(1)
    try:
        got = im.resize(scale, kernel=kernel).numpy()
        report = libvips_amd.gate_report()
    finally:
        libvips_amd.lib.vips_hip_gate_enable(0)
        libvips_amd.lib.vips_hip_gate_reset()
    assert list(report) == ["resize_stream_u8"], report
    assert_same(got, Port.resize(src, scale, kernel=kernel), str((kernel, bands, size, scale)))
    monkeypatch.setenv("VIPS_HIP_NO_RESIZE_STREAM", "1")
    assert np.array_equal(got, im.resize(scale, kernel=kernel).numpy())


@pytest.mark.parametrize("overlap", [False, True, None])
def test_resize_stream_batch_chunks(overlap, monkeypatch):
    """More images than one launch of the streaming resize / the one-kernel sharpen holds (64):
    every image of the batch equals the pipeline run on it alone; also with the sharpen of a
    chunk on a second stream next to the next chunk's resize ($VIPS_HIP_BATCH_OVERLAP).
    overlap None: resize AND sharpen in one kernel (resize_sharpen.hip, $VIPS_HIP_RESIZE_SHARPEN=1)."""
    if overlap:
        monkeypatch.setenv("VIPS_HIP_BATCH_OVERLAP", "1")
    if overlap is None:
        monkeypatch.setenv("VIPS_HIP_RESIZE_SHARPEN", "1")
    srcs = [helpers.lcg_image(688, 96, 3, np.uint8, 900 + k) for k in range(70)]
    ims = [Image.new_from_array(s, interpretation="srgb") for s in srcs]
    libvips_amd.lib.vips_hip_gate_reset()
    libvips_amd.lib.vips_hip_gate_enable(1)
    try:
        outs = libvips_amd.resize_sharpen_batch(ims, 0.125, threads=4)
        report = libvips_amd.gate_report()
    finally:
        libvips_amd.lib.vips_hip_gate_enable(0)
        libvips_amd.lib.vips_hip_gate_reset()
    # (round 6: the one-kernel sharpen is the skip form -- colour.hip sharpen_fused_u8_kernel<*, true> -- by default)
    assert sorted(report) == (["resize_sharpen_u8"] if overlap is None else ["resize_stream_u8", "sharpen_skip_u8"]), report
    for k in (0, 1, 63, 64, 69):
        assert np.array_equal(outs[k].numpy(), ims[k].resize(0.125).sharpen().numpy()), k
        assert np.array_equal(outs[k].numpy(), helpers.PortCC.sharpen(Port.resize(srcs[k], 0.125)))


@pytest.mark.parametrize("bands", [1, 2, 3, 4])
@pytest.mark.parametrize("size,scale,vscale", [
    ((2048, 1536), 0.11, None), ((1531, 1203), 0.3, None), ((4100, 1051), 0.07, 0.19),
    ((640, 480), 0.45, 0.26), ((3000, 700), 0.021, 0.3), ((1024, 1024), 1.0 / 3.0, 0.125),
    ((516, 2049), 0.26, 0.031), ((200, 100), 0.2, None), ((8192, 300), 0.025, 0.4), ((8192, 2563), 1.0 / 9.0, None)])
def test_resize_stream_general(bands, size, scale, vscale, monkeypatch):
    """vips_resize on uchar at ANY scale in one kernel (resize_streamg.hip: the vertical pass from a
    host-made schedule of rows, slots and coefficient phases): ran alone when the rows are dword
    multiples, bit-exact against the port and against the separate kernels; box shrinks 1..40,
    residuals between 1 and 4 on either axis, all band counts, several strips and segments."""
    w, h = size
    src = helpers.lcg_image(w, h, bands, np.uint8, 81)
    kw = {} if vscale is None else {"vscale": vscale}
    im = Image.new_from_array(src)
    monkeypatch.setenv("VIPS_HIP_STREAM_BLOCKS", "4096")
    monkeypatch.setenv("VIPS_HIP_STREAMG_ALWAYS", "1")  # also where the dispatcher prefers the fused tail
    libvips_amd.lib.vips_hip_gate_reset()
    libvips_amd.lib.vips_hip_gate_enable(1)
    try:
        got = im.resize(scale, **kw).numpy()
        report = libvips_amd.gate_report()
    finally:
        libvips_amd.lib.vips_hip_gate_enable(0)
        libvips_amd.lib.vips_hip_gate_reset()
    if (w * bands) % 4 == 0:
        assert list(report) == ["resize_streamg_u8"], report
    assert_same(got, Port.resize(src, scale, **kw), str((bands, size, scale, vscale)))
    monkeypatch.delenv("VIPS_HIP_STREAM_BLOCKS")
    assert np.array_equal(got, im.resize(scale, **kw).numpy())
    monkeypatch.setenv("VIPS_HIP_NO_RESIZE_STREAM", "1")
    assert np.array_equal(got, im.resize(scale, **kw).numpy())


@pytest.mark.parametrize("scale,env,want", [
    (0.07, "", ["resize_streamg_u8", "sharpen_skip_u8"]),
    (0.07, "VIPS_HIP_NO_RESIZE_STREAMG", ["resize_tail_u8", "sharpen_skip_u8", "shrinkv_u8"]),
    (0.3, "VIPS_HIP_NO_RESIZE_STREAMG", ["resize_tail_u8", "sharpen_skip_u8"])])
def test_resize_batch_any_scale(scale, env, want, monkeypatch):
    """A uniform batch whose scale is not 1 / (2 k): the scheduled one-kernel chain
    (resize_streamg.hip), or without it the vertical box shrink and the fused tail (reducev ->
    shrinkh -> reduceh), each one launch per 64 images, then the batched sharpen; every image
    equals the pipeline run on it alone and the port."""
    if env:
        monkeypatch.setenv(env, "1")
    srcs = [helpers.lcg_image(2052, 777, 3, np.uint8, 400 + k) for k in range(70)]
    ims = [Image.new_from_array(s, interpretation="srgb") for s in srcs]
    libvips_amd.lib.vips_hip_gate_reset()
    libvips_amd.lib.vips_hip_gate_enable(1)
    try:
        outs = libvips_amd.resize_sharpen_batch(ims, scale, threads=4)
        report = libvips_amd.gate_report()
    finally:
        libvips_amd.lib.vips_hip_gate_enable(0)
        libvips_amd.lib.vips_hip_gate_reset()
    assert sorted(report) == want, report
    assert all(v[0] == 2 for v in report.values()), report  # two chunks: 64 + 6 images
    for k in (0, 1, 63, 64, 69):
        assert np.array_equal(outs[k].numpy(), ims[k].resize(scale).sharpen().numpy()), k
    for k in (0, 69):
        assert np.array_equal(outs[k].numpy(), helpers.PortCC.sharpen(Port.resize(srcs[k], scale)))


RSH_CASES = [
    # (width, height, images, scale, sigma, sharpen arguments, environment): the cases of
    # tests/test_emul_resize_sharpen.py (where the same kernel body runs on host fibers)
    (704, 512, 2, 0.125, 0.5, {}, {}),
    (1408, 776, 1, 0.125, 0.5, {}, {}),
    (2048, 1000, 2, 0.125, 1.0, {}, {}),
    (2112, 640, 1, 0.0625, 0.5, {}, {}),
    (4096, 256, 1, 0.125, 0.5, {}, {}),
    (4000, 184, 1, 0.125, 0.7, {}, {}),
    (1408, 512, 1, 0.125, 0.5, {"flat": True}, {}),
    (1408, 512, 1, 0.125, 0.5, {"flat": True, "m1": 1.0, "m2": 2.0, "x1": 1.0, "y2": 4.0, "y3": 6.0}, {}),
    (704, 320, 1, 0.125, 1.0, {"m2": 5.0, "y2": 30.0, "y3": 40.0}, {}),
    (1408, 776, 1, 0.125, 0.5, {}, {"VIPS_HIP_STREAM_SEG": "7", "VIPS_HIP_RSH_TW": "9", "VIPS_HIP_STREAM_BURST": "2"}),
    (1408, 776, 1, 0.125, 1.0, {}, {"VIPS_HIP_STREAM_SEG": "7", "VIPS_HIP_RSH_TW": "9", "VIPS_HIP_STREAM_BURST": "2"}),
    (1408, 776, 2, 0.125, 0.5, {}, {"VIPS_HIP_STREAM_SEG": "49", "VIPS_HIP_RSH_TW": "61"}),
    # the BASELINE config 4 and config 1 image sizes
    (8192, 8192, 1, 0.125, 0.5, {}, {}),
    (4096, 4096, 3, 0.125, 0.5, {}, {}),
]


@pytest.mark.parametrize("case", range(len(RSH_CASES)))
def test_resize_sharpen_one_kernel(case, monkeypatch):
    """vips_resize(1 / (2k)) -> vips_sharpen of 3-band sRGB images in ONE kernel
    (resize_sharpen.hip, on request: $VIPS_HIP_RESIZE_SHARPEN=1): ran alone, bit-exact against the compiled reference (or the port) --
    strips, segments, halo rows and columns, 3- and 5-tap blurs, LUT arguments, flat and dark
    areas, the BASELINE config 1 and 4 image sizes."""
    w, h, n, scale, sigma, kw, env = RSH_CASES[case]
    monkeypatch.setenv("VIPS_HIP_RESIZE_SHARPEN", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    srcs = [helpers.lcg_image(w, h, 3, np.uint8, 100 + i) for i in range(n)]
    kw = dict(kw)
    if kw.pop("flat", False):
        for s in srcs:
            s[: h // 2, : w // 2] = (s[: h // 2, : w // 2] // 32).astype(np.uint8)
            s[h // 2:, w // 2:] = 250
    ims = [Image.new_from_array(s, interpretation="srgb") for s in srcs]
    libvips_amd.lib.vips_hip_gate_reset()
    libvips_amd.lib.vips_hip_gate_enable(1)
    try:
        outs = libvips_amd.resize_sharpen_batch(ims, scale, sigma=sigma, **kw)
        report = libvips_amd.gate_report()
    finally:
        libvips_amd.lib.vips_hip_gate_enable(0)
        libvips_amd.lib.vips_hip_gate_reset()
    assert list(report) == ["resize_sharpen_u8"], report
    chain = "resize:scale=%r;sharpen:sigma=%r" % (scale, sigma) + "".join(",%s=%r" % kv for kv in sorted(kw.items()))
    for s, o, im in zip(srcs, outs, ims):
        got = o.numpy()
        if helpers.have_ref():
            want = helpers.Ref.run_chain(chain, s, helpers.INTERP["srgb"])
        else:
            want = helpers.PortCC.sharpen(Port.resize(s, scale), sigma=sigma, **kw)
        assert got.shape == want.shape
        bad = np.argwhere(got != want)
        assert len(bad) == 0, (len(bad), bad[:4])
    # ... and equal to the two operations run one after the other on the device
    monkeypatch.delenv("VIPS_HIP_RESIZE_SHARPEN")
    assert np.array_equal(outs[0].numpy(), ims[0].resize(scale).sharpen(sigma=sigma, **kw).numpy())


def test_resize_sharpen_batch():
    """vips_hip_resize_sharpen_batch (BASELINE config 4's batch entry point): every image of the
    batch equals the same pipeline run on it alone; also without the sharpen, and with a failing
    image in the batch."""
    srcs = [helpers.lcg_image(1024, 768, 3, np.uint8, 60 + k) for k in range(7)]
    ims = [Image.new_from_array(s, interpretation="srgb") for s in srcs]
    outs = libvips_amd.resize_sharpen_batch(ims, 0.125, threads=4)
    for im, out in zip(ims, outs):
        assert np.array_equal(out.numpy(), im.resize(0.125).sharpen().numpy())
    outs = libvips_amd.resize_sharpen_batch(ims, 0.25, sharpen=False, threads=3)
    for im, out in zip(ims, outs):
        assert np.array_equal(out.numpy(), im.resize(0.25).numpy())
    bad = ims[:2] + [Image.new_from_array(helpers.lcg_image(64, 48, 2, np.uint8, 70))] + ims[2:4]
    with pytest.raises(libvips_amd.VipsHipError):
        libvips_amd.resize_sharpen_batch(bad, 0.125, threads=2)  # a 2-band image has no route to LabS
    # as a C caller drives it (bench.py's C4 step): handle arrays, results released in one call
    import ctypes

    from libvips_amd._ffi import lib

    n = len(ims)
    hin = (ctypes.c_void_p * n)(*[im._h.value for im in ims])
    hout = (ctypes.c_void_p * n)()
    for _ in range(2):
        lib.vips_hip_image_unref_many(hout, n)
        assert lib.vips_hip_resize_sharpen_batch(hin, n, hout, 0.125, 5, 2.0, 0.5, 2.0, 10.0, 20.0, 0.0, 3.0, 4) == 0
    got = [Image(h) for h in hout]
    for im, out in zip(ims, got):
        assert np.array_equal(out.numpy(), im.resize(0.125).sharpen().numpy())


def test_resize_sharpen_batch_queued():
    """vips_hip_resize_sharpen_batch_queue: a uniform batch of more than one launch's worth returns
    when it is queued; three batches back to back, each releasing the one before while the device
    may still be working on it (the pool hands those blocks to work queued behind it), then one
    synchronize: the last batch's thumbnails are the synchronous form's, pixel for pixel."""
    import ctypes

    from libvips_amd._ffi import lib

    srcs = [helpers.lcg_image(1024 + 8 * (k % 3 == 0) * 0, 768, 3, np.uint8, 90 + k) for k in range(70)]
    ims = [Image.new_from_array(s, interpretation="srgb") for s in srcs]
    want = libvips_amd.resize_sharpen_batch(ims, 0.125, threads=4)
    n = len(ims)
    hin = (ctypes.c_void_p * n)(*[im._h.value for im in ims])
    hout = (ctypes.c_void_p * n)()
    for _ in range(3):
        lib.vips_hip_image_unref_many(hout, n)
        assert lib.vips_hip_resize_sharpen_batch_queue(hin, n, hout, 0.125, 5, 2.0, 0.5, 2.0, 10.0, 20.0, 0.0, 3.0, 4) == 0
    libvips_amd.synchronize()
    got = [Image(h) for h in hout]
    for k in range(n):
        assert np.array_equal(got[k].numpy(), want[k].numpy()), k
    # the Python mirror's spelling, and a batch the library runs image by image (it completes before the return)
    outs = libvips_amd.resize_sharpen_batch(ims[:5] + [Image.new_from_array(helpers.lcg_image(640, 480, 3, np.uint8, 7),
                                                                            interpretation="srgb")], 0.125, wait=False)
    for k in range(5):
        assert np.array_equal(outs[k].numpy(), want[k].numpy()), k
    assert outs[5].width == 80


@pytest.mark.parametrize("shrink,kernel", [(8, "lanczos3"), (4, "lanczos3"), (2, "lanczos3"), (3, "lanczos3"), (5, "lanczos3"),
                                           (6, "lanczos3"), (8, "cubic"), (4, "lanczos2"), (2, "mitchell")])
def test_reducev_shrinkv_f32_stream(shrink, kernel, monkeypatch):
    """Round 6: vips_reducev / vips_shrinkv of float images as streams (resample_f32.hip): integer shrinks with one
    coefficient phase, the taps added in the reference's order in double (templates.h:183-194), every row read once a
    segment; bit for bit the general kernels' and the port's results (float: 0 ULP, both are the same IEEE operations
    in the same order) on an image with several lanes' worth of columns, heights the factor does and does not divide,
    a region that starts below the top, values over many binades."""
    rng = np.random.RandomState(7 + shrink)
    for (w, h, bands) in ((512, 64 * shrink, 3), (344 * shrink, 37 * shrink + 5, 3), (300 * shrink, 41, 4), (1100, 19, 1)):
        src = (rng.standard_normal((h, w, bands)) * np.exp2(rng.randint(-8, 8, (h, w, bands)))).astype(np.float32)
        im = Image.new_from_array(src)
        lib = libvips_amd.lib
        for op, ref in ((lambda i: i.reducev(shrink, kernel=kernel), lambda a: Port.reducev(a, shrink, kernel)),
                        (lambda i: i.reduceh(shrink, kernel=kernel), lambda a: Port.reduceh(a, shrink, kernel)),
                        (lambda i: i.shrinkv(shrink), lambda a: Port.shrinkv(a, shrink))):
            lib.vips_hip_gate_reset()
            lib.vips_hip_gate_enable(1)
            try:
                got = op(im).numpy()
                report = libvips_amd.gate_report()
            finally:
                lib.vips_hip_gate_enable(0)
                lib.vips_hip_gate_reset()
            if bands != 1:  # (1 100 x 19 x 1: too few elements a row for the streams -- the general kernels' case)
                assert any(k.endswith("_f32_stream") or k == "reduceh_f32_lds" for k in report), report
            monkeypatch.setenv("VIPS_HIP_NO_F32_STREAM", "1")
            old = op(im).numpy()
            monkeypatch.delenv("VIPS_HIP_NO_F32_STREAM")
            want = ref(src)
            assert got.dtype == np.float32 and got.shape == want.shape
            assert np.array_equal(got.view(np.int32), old.view(np.int32)), (shrink, kernel, w, h)
            assert np.array_equal(got.view(np.int32), want.view(np.int32)), (shrink, kernel, w, h)


@pytest.mark.parametrize("bands", [1, 2, 3, 4])
def test_upsize_bicubic_walk(bands, monkeypatch):
    """Round 6: bicubic enlargement of uchar images by a thread walking its output column down a segment of rows
    with the four rounded horizontal sums in registers (upsize_bicubic_u8_walk): several segments, enlarging and
    (vertically) reducing scales -- the window then moves by more than one input row a step --, against the kernel
    that makes every pixel from scratch and against the port (resample/bicubic.cpp:482-600)."""
    lib = libvips_amd.lib
    for (w, h, hs, vs) in ((200, 150, 2.5, 2.5), (97, 260, 1.3, 3.7), (120, 400, 2.0, 0.45), (64, 90, 5.0, 1.0)):
        src = helpers.lcg_image(w, h, bands, np.uint8, 91)
        im = Image.new_from_array(src)
        lib.vips_hip_gate_reset()
        lib.vips_hip_gate_enable(1)
        try:
            got = im.resize(hs, vscale=vs, kernel="cubic").numpy()
            report = libvips_amd.gate_report()
        finally:
            lib.vips_hip_gate_enable(0)
            lib.vips_hip_gate_reset()
        assert "upsize_bicubic_u8_walk" in report, report
        monkeypatch.setenv("VIPS_HIP_NO_UPSIZE_WALK", "1")
        old = im.resize(hs, vscale=vs, kernel="cubic").numpy()
        monkeypatch.delenv("VIPS_HIP_NO_UPSIZE_WALK")
        assert np.array_equal(got, old), (w, h, hs, vs)
        assert np.array_equal(got, Port.resize(src, hs, vs, kernel="cubic")), (w, h, hs, vs)


@pytest.mark.parametrize("size,width", [((1024, 700), 100), ((2047, 1233), 250), ((640, 480), 64), ((3000, 501), 333)])
def test_thumbnail_rgba_premultiply_on_load(size, width, monkeypatch):
    """Round 6: vips_thumbnail_image of an RGBA uchar image (thumbnail.c:848-904: premultiply -> resize -> unpremultiply)
    with vips_premultiply's uchar fast path applied to the pixels the resize's FIRST kernel loads
    (reduce_band_body.h rb_premul: no premultiplied image in between) -- every alpha value, alpha 0 and 255 runs, sizes
    whose box shrink is 2 .. 9; against the port (pinned on the compiled reference: goldens thumbnail|rgba) / the
    reference, and against the separate premultiply kernel."""
    w, h = size
    src = helpers.lcg_image(w, h, 4, np.uint8, 31 + width)
    src[: h // 4, : w // 3, 3] = 255
    src[h // 4: h // 2, : w // 3, 3] = 0
    src[0, :256, 3] = np.arange(256)
    im = Image.new_from_array(src, interpretation="srgb")
    lib = libvips_amd.lib
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    try:
        got = im.thumbnail_image(width).numpy()
        report = libvips_amd.gate_report()
    finally:
        lib.vips_hip_gate_enable(0)
        lib.vips_hip_gate_reset()
    # (a residual of exactly 2 -- 640 -> 64 -- is the one-kernel chain's: the premultiply stays its own kernel there)
    if width != 64:
        assert "premultiply" not in report and "shrinkv_reducev_u8_band" in report and "unpremultiply" in report, report
    else:
        assert "premultiply" in report, report
    monkeypatch.setenv("VIPS_HIP_NO_BAND_PREMUL", "1")
    old = im.thumbnail_image(width).numpy()
    assert np.array_equal(got, old)
    if helpers.have_ref():
        want = helpers.Ref.run("thumbnail_image", src, "width=%d" % width, helpers.INTERP["srgb"])
    else:
        want = helpers.PortCC.thumbnail_image(src, "srgb", width)
    assert got.shape == want.shape and np.array_equal(got, want)
