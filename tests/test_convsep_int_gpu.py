"""GPU parity of convsep_stream's integer horizontal pass (convsep_int_body.h): gaussblur and
gaussblur + colourspace on float images that hold the integers 0 .. 255 (BASELINE config 3's input),
images that hold them almost everywhere, and float images proper; the result must not depend on the
path: $VIPS_HIP_STREAM_INT=1 against =0 against the oracle (the plain-C port), bit for bit, and =2
(a refused window is poisoned instead of recomputed in double) proves the integer pass made every
pixel of an all-integer image."""
import os

import numpy as np
import pytest

import libvips_amd
from libvips_amd import Image
from tests import helpers
from tests.helpers import PortCC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    libvips_amd.init(0)


def _with_int(mode, fn):
    old = os.environ.get("VIPS_HIP_STREAM_INT")
    os.environ["VIPS_HIP_STREAM_INT"] = str(mode)
    try:
        return fn()
    finally:
        if old is None:
            del os.environ["VIPS_HIP_STREAM_INT"]
        else:
            os.environ["VIPS_HIP_STREAM_INT"] = old


def _bits(a):
    return a.view(np.uint32)


def _integer_image(w, h, bands, seed):
    return helpers.lcg_image(w, h, bands, np.uint8, seed).astype(np.float32)


def _default(fn):
    """fn() with $VIPS_HIP_STREAM_INT unset: the product's own choice of path."""
    old = os.environ.pop("VIPS_HIP_STREAM_INT", None)
    try:
        return fn()
    finally:
        if old is not None:
            os.environ["VIPS_HIP_STREAM_INT"] = old


def _want_blur(src, sigma):
    """The compiled reference itself when it is there (the port is pinned against it, but this file is the
    integer pass's only direct comparison), else the port."""
    if helpers.have_ref():
        return helpers.Ref.run_chain("gaussblur:sigma=%r,precision=integer" % sigma, src)
    return PortCC.gaussblur(src, sigma, precision="integer")


@pytest.mark.parametrize("shape", [(700, 300, 3), (2300, 140, 3), (37, 411, 3), (5, 3, 3), (1030, 77, 3), (1500, 90, 1), (260, 200, 4)])
@pytest.mark.parametrize("sigma", [8.0, 2.0, 0.6])
def test_integer_image_blur(shape, sigma):
    w, h, b = shape
    src = _integer_image(w, h, b, 91)
    want = _want_blur(src, sigma)
    got = _default(lambda: Image.new_from_array(src).gaussblur(sigma, precision="integer").numpy())
    assert np.array_equal(_bits(got), _bits(want)), ("default", shape, sigma)
    for mode in (0, 1, 2):
        got = _with_int(mode, lambda: Image.new_from_array(src).gaussblur(sigma, precision="integer").numpy())
        assert got.dtype == np.float32 and got.shape == want.shape
        assert np.array_equal(_bits(got), _bits(want)), (mode, shape, sigma)


@pytest.mark.parametrize("shape", [(700, 300), (2300, 140), (37, 411), (1030, 77)])
@pytest.mark.parametrize("sigma,space", [(8.0, "lab"), (2.0, "xyz"), (3.1, "lab")])
def test_integer_image_blur_colourspace(shape, sigma, space):
    """BASELINE config 3's shape: the colour epilogue behind the integer pass."""
    w, h = shape
    src = _integer_image(w, h, 3, 92)
    want = PortCC.colourspace(PortCC.gaussblur(src, sigma, precision="integer"), space, "srgb")
    for mode in (0, 1, 2):
        got = _with_int(mode, lambda: Image.new_from_array(src, interpretation="srgb").gaussblur_colourspace(sigma, space).numpy())
        assert np.array_equal(_bits(got), _bits(want)), (mode, shape, sigma, space)


def test_epilogue_forms():
    """BASELINE config 3's instantiation has two forms of the colour epilogue (convsep_stream.hip, EPIF):
    the same bits from both, on an integer image and on a float image proper."""
    for src in (_integer_image(1600, 200, 3, 96), helpers.lcg_image(1600, 200, 3, np.float32, 97)):
        want = PortCC.colourspace(PortCC.gaussblur(src, 8.0, precision="integer"), "lab", "srgb")
        for form in ("0", "1"):
            os.environ["VIPS_HIP_STREAM_EPI"] = form
            try:
                got = Image.new_from_array(src, interpretation="srgb").gaussblur_colourspace(8.0, "lab").numpy()
            finally:
                del os.environ["VIPS_HIP_STREAM_EPI"]
            assert np.array_equal(_bits(got), _bits(want)), form


@pytest.mark.parametrize("sigma", [8.0, 2.0])
def test_almost_integer_image(sigma):
    """Integers with everything else sprinkled in: waves whose windows hold a fraction, a negative or
    too large integer, -0, inf, NaN or a denormal take the double pass, their neighbours the integer
    one, and the seams do not show."""
    w, h = 1800, 260
    src = _integer_image(w, h, 3, 93)
    src[100:121, 50:90] += 0.25
    src[7, 1000, 1] = 300.0
    src[8, 1200, 0] = -7.0
    src[30, 400, 2] = -0.0
    src[31, 401, 2] = 1e-40
    src[200, 1700, 0] = 255.00002
    src[150, 5, 1] = 0.5
    src[h - 1, w - 1, 2] = 77.5
    for nonfinite in (False, True):
        if nonfinite:
            src[60, 900, 0] = np.inf
            src[61, 901, 1] = np.nan
        want = _want_blur(src, sigma)
        a = _with_int(1, lambda: Image.new_from_array(src).gaussblur(sigma, precision="integer").numpy())
        b = _with_int(0, lambda: Image.new_from_array(src).gaussblur(sigma, precision="integer").numpy())
        d = _default(lambda: Image.new_from_array(src).gaussblur(sigma, precision="integer").numpy())
        assert np.array_equal(_bits(a), _bits(b)), (sigma, nonfinite)
        assert np.array_equal(_bits(a), _bits(d)), (sigma, nonfinite)  # (mixed windows under the product's default)
        assert np.array_equal(a, want, equal_nan=True), (sigma, nonfinite)
        if not nonfinite:
            assert np.array_equal(_bits(a), _bits(want)), sigma
    # the poisoning mode must show here (the test of the test)
    c = _with_int(2, lambda: Image.new_from_array(src).gaussblur(sigma, precision="integer").numpy())
    assert np.isnan(c).sum() > np.isnan(want).sum()


def test_float_image_and_masks_outside_the_pass():
    """A float image proper (every wave refuses, once per work item), and masks the pass does not
    take (negative taps, an offset, taps above 255): the same bits as without it."""
    src = helpers.lcg_image(900, 200, 3, np.float32, 94)
    for sigma in (8.0, 2.0):
        a = _with_int(1, lambda: Image.new_from_array(src).gaussblur(sigma, precision="integer").numpy())
        b = _with_int(0, lambda: Image.new_from_array(src).gaussblur(sigma, precision="integer").numpy())
        assert np.array_equal(_bits(a), _bits(b)), sigma
    isrc = _integer_image(900, 200, 3, 95)
    for mask, scale, offset in (([[-1.0, 2.0, 5.0, 7.0, 5.0, 3.0, -2.0]], 19.0, 0.0), ([[1.0, 2.0, 5.0, 2.0, 1.0]], 11.0, 3.0),
                                ([[1.0, 300.0, 1.0]], 302.0, 0.0), ([[1.0, 2.0, 5.0, 2.0, 1.0]], 1.0, 0.0),
                                ([[3.0, 200.0, 255.0, 200.0, 3.0]], 661.0, 0.0)):
        m = np.array(mask)
        want = PortCC.convsep(isrc, m, scale, offset, "integer")
        for mode in (0, 1):
            got = _with_int(mode, lambda: Image.new_from_array(isrc).convsep(m, scale=scale, offset=offset, precision="integer").numpy())
            assert np.array_equal(_bits(got), _bits(want)), (mask, scale, offset, mode)
