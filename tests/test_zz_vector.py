"""convi as a Highway-built libvips computes it on uchar images (second
half): 8-bit mantissas with one shared exponent, int32 sum, arithmetic shift.

PARITY UNPINNED: libhwy is not in this image, so the reference's vector path cannot be run; the
oracle restates convolution/convi.c:925-1120 (the intize) and the scalar tail of
convi_hwy.cpp:264-273, which the vector body has to equal lane for lane.  What can be checked:
the product's intize == the oracle's, the result stays within the tolerance the reference's own
test-suite allows between its vector and C paths, and (GPU) the device == the oracle.  The switch
is off by default and these tests run last."""
import ctypes

import numpy as np
import pytest

from tests import helpers
from tests.golden import cases
from tests.helpers import PortCC


def _masks():
    rng = np.random.RandomState(3)
    masks = [cases.MASKS[k] for k in ("blur3", "rand5x7", "sobel", "zeros", "row5")]
    masks += [cases.CA_MASKS[k] for k in sorted(cases.CA_MASKS)]
    for sig in (0.5, 1, 2, 3, 8):
        for sep in (True, False):
            m, s = PortCC.gaussmat(sig, 0.2, sep, "integer")
            masks.append((m, s, 0.0))
    for _ in range(120):
        mw, mh = rng.randint(1, 12, size=2)
        masks.append((np.round(rng.randn(mh, mw) * rng.choice([1, 5, 40]), 2),
                      float(rng.choice([1, 3.5, 16, 100])), float(rng.randint(-3, 4))))
    return masks


def test_product_intize_equals_the_oracle():
    from libvips_amd._ffi import lib

    accepted = refused = 0
    for m, scale, offset in _masks():
        m = np.ascontiguousarray(m, dtype=np.float64)
        want = PortCC.convi_vector_intize(m, scale)
        plan = lib.vips_hip_conv_new(m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), m.shape[1], m.shape[0],
                                     scale, offset, 0)
        assert plan
        exp = ctypes.c_int()
        mant = (ctypes.c_int * m.size)()
        pos = (ctypes.c_int * m.size)()
        k = lib.vips_hip_conv_get_vector(plan, ctypes.byref(exp), mant, pos, m.size)
        lib.vips_hip_conv_free(plan)
        got = None if k == 0 else (exp.value, [(mant[i], pos[i]) for i in range(k)])
        assert got == want, (m.shape, scale)
        accepted += got is not None
        refused += got is None
    assert accepted > 50 and refused > 5


def test_vector_arithmetic_stays_close_to_the_c_path():
    # the reference's test-suite compares its convolutions with a true value within a few grey
    # levels (test_convolution.py:68-86, 167-196); the intize itself accepts at most 2 on a flat image
    src = helpers.lcg_image(64, 48, 3, np.uint8, 101)
    for sigma in (1.0, 2.5, 8.0):
        m, s = PortCC.gaussmat(sigma, 0.2, False, "integer")
        v = PortCC.convi_vector(src, m, s, 0.0).astype(int)
        c = PortCC.conv(src, m, s, 0.0, "integer").astype(int)
        assert np.abs(v - c).max() <= 3
    # arithmetic shift is a floor: a negative total must not round toward zero
    lap = np.array([[0, -1, 0], [-1, 4, -1], [0, -1, 0.0]])
    assert PortCC.convi_vector_intize(lap, 1.0) is not None
    v = PortCC.convi_vector(src, lap, 1.0, 128.0)
    assert v.dtype == np.uint8 and v.shape == src.shape


def test_switch_defaults_to_off():
    import libvips_amd

    assert libvips_amd.vector_isenabled() is False
    libvips_amd.vector_set_enabled(True)
    assert libvips_amd.vector_isenabled() is True
    libvips_amd.vector_set_enabled(False)
    assert libvips_amd.vector_isenabled() is False


@pytest.mark.gpu
def test_hip_vector_convi_matches_the_oracle():
    import libvips_amd
    from libvips_amd import Image

    libvips_amd.init(0)
    src = helpers.lcg_image(157, 93, 3, np.uint8, 102)
    im = Image.new_from_array(src)
    other = helpers.lcg_image(80, 60, 2, np.uint16, 103)
    libvips_amd.vector_set_enabled(True)
    try:
        n_vector = 0
        for m, scale, offset in _masks()[:70]:
            want = PortCC.convi_vector(src, m, scale, offset)
            got = im.conv(m, scale=scale, offset=offset, precision="integer").numpy()
            assert np.array_equal(got, want), (np.asarray(m).shape, scale)
            n_vector += PortCC.convi_vector_intize(m, scale) is not None
        assert n_vector > 20
        # gaussblur = two convi passes, each with the vector arithmetic
        m, s = PortCC.gaussmat(3.0, 0.2, True, "integer")
        t = PortCC.convi_vector(src, m, s, 0.0)
        want = PortCC.convi_vector(t, m.reshape(-1, 1), s, 0.0)
        assert np.array_equal(im.gaussblur(3.0, precision="integer").numpy(), want)
        # other formats, float precision and precision=approximate do not change
        assert np.array_equal(Image.new_from_array(other).gaussblur(3.0, precision="integer").numpy(),
                              PortCC.gaussblur(other, 3.0, precision="integer"))
        assert np.array_equal(im.gaussblur(3.0, precision="approximate").numpy(),
                              PortCC.gaussblur(src, 3.0, precision="approximate"))
    finally:
        libvips_amd.vector_set_enabled(False)
    assert np.array_equal(im.gaussblur(3.0, precision="integer").numpy(),
                          PortCC.gaussblur(src, 3.0, precision="integer"))
