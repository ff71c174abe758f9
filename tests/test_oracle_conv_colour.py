"""CPU: pin the conv / colour / sharpen / cast half of the oracle (oracle/port) against
golden vectors from the compiled reference, the reference's known answers, and oracle/_ref
directly where it is present."""
import os

import numpy as np
import pytest

from tests import helpers
from tests.golden import cases
from tests.helpers import PortCC, Ref

GOLD = np.load(os.path.join(helpers.GOLDEN, "conv_colour.npz"))
needs_ref = pytest.mark.skipif(not helpers.have_ref(), reason="oracle/_ref not built")


def port_call(case, src):
    kw = dict(case["kwargs"])
    if case["kind"] == "mask":
        mask, scale, offset = cases.MASKS[case["mask"]]
        return getattr(PortCC, case["method"])(src, mask, scale, offset, **kw)
    if case["method"] == "colourspace":
        return PortCC.colourspace(src, kw["space"], case["interp"])
    if case["method"] == "sharpen":
        return PortCC.sharpen(src, case["interp"], **kw)
    if case["method"] in ("premultiply", "unpremultiply"):
        return PortCC.premultiply(src, case["interp"], uchar=kw["uchar"],
                                  inverse=case["method"] == "unpremultiply")
    if case["method"] == "thumbnail_image":
        return PortCC.thumbnail_image(src, case["interp"], **kw)
    if case["method"] == "cast":
        inv = {v: k for k, v in cases._FMT.items()}
        return PortCC.cast(src, np.dtype(inv[kw["format"]]))
    return getattr(PortCC, case["method"])(src, **kw)


@pytest.mark.parametrize("case", cases.CC_CASES, ids=[c["name"] for c in cases.CC_CASES])
def test_port_matches_golden(case):
    src = cases.cc_input(case)
    want = GOLD[case["name"]]
    got = port_call(case, src)
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


def test_lab_to_xyz_known_answer():
    # test/test-suite/test_colour.py:53-57: Lab(50,0,0) -> XYZ (17.5064, 18.4187, 20.0547)
    xyz = PortCC.colourspace(np.array([[[50, 0, 0]]], np.float32), "xyz", "lab")[0, 0]
    assert np.allclose(xyz, [17.5064, 18.4187, 20.0547], atol=1e-4)


def test_gaussmat_known_answers():
    # SURVEY.md appendix (probed from the reference CLI)
    m, scale = PortCC.gaussmat(8, 0.2, True, "integer")
    assert m.shape == (1, 29) and scale == 372
    assert list(m[0].astype(int)) == [4, 5, 6, 8, 9, 11, 12, 14, 15, 16, 18, 19, 19, 20, 20, 20, 19, 19,
                                      18, 16, 15, 14, 12, 11, 9, 8, 6, 5, 4]
    m, scale = PortCC.gaussmat(0.5, 0.1, True, "integer")
    assert list(m[0].astype(int)) == [3, 20, 3] and scale == 26
    m, scale = PortCC.gaussmat(5, 0.01, False, "float")
    assert m.shape == (31, 31)


def test_colour_round_trips():
    # test_colour.py:41-51: round trips through every colourspace within 0.1
    lab = np.array([[[50.0, 10.0, 20.0]]], np.float32)
    for space in ("xyz", "scrgb", "srgb", "labs"):
        there = PortCC.colourspace(lab, space, "lab")
        back = PortCC.colourspace(there, "lab", space)
        tol = 1.0 if space == "srgb" else 0.1  # 8-bit sRGB quantises
        assert np.abs(back - lab).max() < tol, space


def test_sharpen_identity():
    # test_convolution.py:198-219: m1 = m2 = 0 leaves the image alone (max diff 0)
    src = helpers.lcg_image(40, 30, 3, np.uint8, 51)
    out = PortCC.sharpen(src, "srgb", m1=0.0, m2=0.0)
    # identity up to the LabS round trip, which the reference's test tolerates with
    # "max diff 0" on ITS sample; check against the round trip itself here
    trip = PortCC.colourspace(PortCC.colourspace(src, "labs", "srgb"), "srgb", "labs")
    assert np.array_equal(out, trip)


def test_conv_spot_check():
    # test_convolution.py:13-23,68-80: conv against a direct python sum at one pixel
    src = helpers.lcg_image(30, 20, 1, np.uint8, 52)
    mask, scale, offset = cases.MASKS["blur3"]
    out = PortCC.conv(src, mask, scale, offset, "float")
    x, y = 10, 7
    s = sum(mask[j, i] * float(src[y + j - 1, x + i - 1, 0]) for j in range(3) for i in range(3))
    assert abs(out[y, x, 0] - (s / scale + offset)) < 1e-4


@needs_ref
def test_port_vs_ref_c3_pipeline():
    # BASELINE config 3 at reduced size: gaussblur(sigma 8) -> colourspace(LAB) on float sRGB
    src = helpers.lcg_image(160, 120, 3, np.float32, 53)
    want = Ref.run_chain("gaussblur:sigma=8;colourspace:space=lab", src, cases.INTERP["srgb"])
    got = PortCC.colourspace(PortCC.gaussblur(src, 8.0), "lab", "srgb")
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


@needs_ref
def test_port_vs_ref_c4_pipeline():
    # BASELINE config 4 per image at reduced size: resize(1/8) -> sharpen, sRGB u8
    src = helpers.lcg_image(512, 384, 3, np.uint8, 54)
    want = Ref.run_chain("resize:scale=0.125;sharpen:", src, cases.INTERP["srgb"])
    got = PortCC.sharpen(helpers.Port.resize(src, 0.125), "srgb")
    assert np.array_equal(got, want)


@needs_ref
def test_port_vs_ref_c5_conv31():
    # BASELINE config 5 at reduced size: 31x31 float gaussian mask on ushort
    src = helpers.lcg_image(96, 80, 1, np.uint16, 55)
    mask, scale = PortCC.gaussmat(5, 0.01, False, "float")
    want = Ref.run_mask("conv", src, mask, scale, 0.0, "precision=float")
    got = PortCC.conv(src, mask, scale, 0.0, "float")
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


@needs_ref
def test_port_vs_ref_convsep_signed_zeros():
    """The sign of zero through convsep on float (convi.c:721-741: the sum starts at +0.0, the
    division by a negative scale makes -0.0, + offset): the port against the compiled reference,
    bit for bit -- the input the GPU suite's test_fused_convsep_float_signed_zeros uses."""
    src = helpers.lcg_image(300, 120, 3, np.float32, 71)
    src[:40] = 0.0
    src[40:80, :150] = -0.0
    src[40:80, 150:] = -1e-40
    src[100:, 200:] = -0.0
    mask = np.array([[1.0, 2.0, 5.0, 7.0, 5.0, 2.0, 1.0]])
    for m, scale in ((mask, 23.0), (mask, -23.0), (-mask, 1.0), (mask, 1.0)):
        for offset in (0.0, -0.0, 2.5):
            want = Ref.run_mask("convsep", src, m, scale, offset, "precision=integer")
            got = PortCC.convsep(src, m, scale, offset, "integer")
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (scale, offset)


@needs_ref
def test_srgb_labs_srgb_is_the_identity_on_every_colour():
    """What colour.hip's sharpen_skip_u8 rests on (round 6): with the compiled reference, sRGB -> LabS -> sRGB gives
    back every one of the 2^24 uchar colours -- so a pixel vips_sharpen's LUT leaves alone (lut[L - blur] == 0: the
    whole centre section for m1 = 0, sharpen.c:230-257) leaves the operation as it came in.  (The library repeats the
    check on the device with its own tables and functions before it selects that kernel: sharpen_identity_kernel.)"""
    v = np.arange(1 << 24, dtype=np.uint32)
    img = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], axis=-1).astype(np.uint8).reshape(4096, 4096, 3)
    back = Ref.run_chain("colourspace:space=labs;colourspace:space=srgb", img, cases.INTERP["srgb"])
    assert back.dtype == np.uint8 and back.shape == img.shape
    assert int((back != img).any(axis=-1).sum()) == 0
    # and the LUT's zero run with the default parameters: |difference| <= 654 or so LabS units either side of 0
    i = np.arange(65536)
    d = (i - 32767) / 327.67
    y = np.where(d < -2.0, (d + 2.0) * 3.0, np.where(d < 2.0, d * 0.0, (d - 2.0) * 3.0))
    lut = np.rint(np.clip(y, -20.0, 10.0) * 327.67).astype(np.int64)
    zeros = np.nonzero(lut == 0)[0] - 32768
    assert zeros.min() <= -650 and zeros.max() >= 650 and np.all(np.diff(zeros) == 1)
