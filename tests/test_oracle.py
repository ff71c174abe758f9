"""CPU: pin the oracle (oracle/port) against the reference.

(a) golden vectors produced by the compiled reference (tests/golden/*.npz),
(b) the SURVEY.md 8(c) known checksum, (c) directly against oracle/_ref when the
    compiled reference is present, (d) the statistical properties the reference's own
    tests check (test/test-suite/test_resample.py:77-169).
"""
import os

import numpy as np
import pytest

from tests import helpers
from tests.golden import cases
from tests.helpers import Port, Ref

GOLD = np.load(os.path.join(helpers.GOLDEN, "resample.npz"))

needs_ref = pytest.mark.skipif(not helpers.have_ref(), reason="oracle/_ref not built")


def port_call(case, src):
    fn, kw = case["call"]
    return getattr(Port, fn)(src, **kw)


@pytest.mark.parametrize("case", cases.RESAMPLE_CASES, ids=[c["name"] for c in cases.RESAMPLE_CASES])
def test_port_matches_golden(case):
    src = helpers.lcg_image(case["width"], case["height"], case["bands"], case["dtype"], case["seed"])
    want = GOLD[case["name"]]
    got = port_call(case, src)
    assert got.shape == want.shape
    assert got.dtype == want.dtype
    # bit-exact, floats included (same double arithmetic in the same order)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


def test_port_known_checksum():
    # vips_reduce(8,8,lanczos3) on the 4096^2 x4 LCG image
    src = helpers.lcg_image(4096, 4096, 4, np.uint8, 12345)
    got = Port.reduce(src, 8, 8, "lanczos3")
    assert got.shape == (512, 512, 4)
    assert helpers.checksum(got) == 16793779256


@needs_ref
def test_ref_known_checksum():
    src = helpers.lcg_image(4096, 4096, 4, np.uint8, 12345)
    got = Ref.run("reduce", src, "hshrink=8,vshrink=8,kernel=lanczos3")
    assert helpers.checksum(got) == 16793779256


@needs_ref
@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.float32])
@pytest.mark.parametrize("fac", [1.0, 1.1, 1.5, 1.999])
def test_port_vs_ref_reduce_factors(dtype, fac):
    # the factor walk of test_resample.py:77-92
    src = helpers.lcg_image(90, 70, 3, dtype, 31)
    for kernel in ("nearest", "linear", "cubic", "mitchell", "lanczos2", "lanczos3", "mks2013", "mks2021"):
        want = Ref.run("reduce", src, "hshrink=%g,vshrink=%g,kernel=%s" % (fac, fac, kernel))
        got = Port.reduce(src, fac, fac, kernel)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), kernel


@needs_ref
def test_port_vs_ref_tile_seeding():
    # non-dyadic factor, tall image: Y is re-seeded every 16 output rows
    # (reducev.cpp:548 + thread.c:301-325); the port's `tile` must reproduce it.
    src = helpers.lcg_image(33, 1500, 1, np.uint8, 32)
    want = Ref.run("reducev", src, "vshrink=2.7182818,kernel=lanczos3")
    got = Port.reducev(src, 2.7182818, "lanczos3", tile=16)
    assert np.array_equal(got, want)


def test_reduce_constant_stays_constant():
    # test_resample.py:94-103
    for const in (0, 1, 127, 255):
        src = np.full((40, 50, 3), const, dtype=np.uint8)
        for kernel in ("linear", "cubic", "lanczos3", "mks2021"):
            out = Port.reduce(src, 1.5, 1.999, kernel)
            assert out.min() == const and out.max() == const, (const, kernel)


def test_reduce_average_preserved():
    # test_resample.py:83-92: abs(r.avg() - im.avg()) < 2
    src = helpers.lcg_image(200, 160, 3, np.uint8, 33)
    for fac in (1.1, 1.5, 1.999):
        out = Port.reduce(src, fac, fac, "lanczos3")
        assert abs(out.mean() - src.mean()) < 2


def test_shrink_average_preserved():
    # test_resample.py:148-169
    src = helpers.lcg_image(200, 160, 3, np.uint8, 34)
    out = Port.shrink(src, 4, 4)
    assert out.shape == (40, 50, 3)
    assert abs(out.mean() - src.mean()) < 1
