"""GPU: vips_reducev on uchar with a coefficient row per output row as a banded matrix product on the matrix
cores (reduce_band.hip) against the compiled reference, whole image, bit for bit -- the cases of
tests/test_emul_reduce_band.py on the device, plus large images and the two-axis fractional reduce."""
import numpy as np
import pytest

import libvips_amd
from libvips_amd import Image
from tests import helpers
from tests.test_emul_reduce_band import CASES, CASES16, HCASES

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not helpers.have_ref(), reason="oracle/_ref missing")]

BIG = [(8192, 2100, 3, 7.3, "lanczos3", "reducev_u8_band"), (8192, 8192, 3, 7.3, "lanczos3", "reducev_u8_band"),
       (5000, 3000, 4, 2.9, "lanczos3", "reducev_u8_band"), (1000, 9000, 1, 16.5, "lanczos3", "reducev_u8_band")]


BIG16 = [(4096, 3001, 3, 7.3, "lanczos3", "reducev_u16_band"), (16384, 2048, 4, 8.0, "lanczos3", "reducev_u16_band")]


@pytest.mark.parametrize("w,h,bands,shrink,kernel,gate", CASES + BIG + [c for c in CASES16 if c[5].startswith("reducev")] + BIG16)
def test_reducev_band_vs_reference(w, h, bands, shrink, kernel, gate):
    lib = libvips_amd.lib
    dt = np.uint16 if "u16" in gate else np.uint8
    src = helpers.lcg_image(w, h, bands, dt, 11 + w)
    src[: h // 3, : w // 2] = 65535 if dt == np.uint16 else 255
    src[h // 3: h // 2, w // 2:] = 0
    im = Image.new_from_array(src)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    got = im.reducev(shrink, kernel=kernel).numpy()
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    want = helpers.Ref.run_chain("reducev:vshrink=%r,kernel=%s" % (shrink, kernel), src)
    assert list(report) == [gate], report
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.array_equal(got, want)


HBIG = [(8192, 1122, 3, 7.3, "lanczos3", "reduceh_u8_band"), (5000, 700, 4, 2.9, "lanczos3", "reduceh_u8_band"),
        (9000, 300, 1, 16.5, "lanczos3", "reduceh_u8_band")]


HBIG16 = [(8192, 1122, 3, 7.3, "lanczos3", "reduceh_u16_band"), (16384, 512, 4, 8.0, "lanczos3", "reduceh_u16_band")]


@pytest.mark.parametrize("w,h,bands,shrink,kernel,gate", HCASES + HBIG + [c for c in CASES16 if c[5].startswith("reduceh")] + HBIG16)
def test_reduceh_band_vs_reference(w, h, bands, shrink, kernel, gate):
    lib = libvips_amd.lib
    dt = np.uint16 if "u16" in gate else np.uint8
    src = helpers.lcg_image(w, h, bands, dt, 11 + w)
    src[: h // 3, : w // 2] = 65535 if dt == np.uint16 else 255
    src[h // 3: h // 2, w // 2:] = 0
    im = Image.new_from_array(src)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    got = im.reduceh(shrink, kernel=kernel).numpy()
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    want = helpers.Ref.run_chain("reduceh:hshrink=%r,kernel=%s" % (shrink, kernel), src)
    assert list(report) == [gate], report
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.array_equal(got, want)


def test_reduce_fractional_both_axes():
    src = helpers.lcg_image(4096, 2048, 3, np.uint8, 3)
    lib = libvips_amd.lib
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    got = Image.new_from_array(src).reduce(7.3, 7.3).numpy()
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    assert sorted(report) == ["reduceh_u8_band", "reducev_u8_band"], report
    want = helpers.Ref.run_chain("reduce:hshrink=7.3,vshrink=7.3", src)
    assert np.array_equal(got, want)


from tests.test_emul_reduce_band import RESIZE_CASES  # noqa: E402

BV, BH, SH = "shrinkv_reducev_u8_band", "reduceh_u8_band", "shrinkh_u8_stream"
RESIZE_BIG = [
    (8192, 8192, 3, 1000 / 8192.0, None, [BV, SH, BH]), (8192, 8192, 3, 500 / 8192.0, None, [BV, SH, BH]),
    (6000, 4000, 3, 0.19, None, [BV, SH, BH]), (5001, 3337, 4, 0.3, 0.07, [BV, BH]),
    (4096, 4099, 1, 0.16, None, [BV, SH, BH]), (3000, 3000, 2, 1 / 11.1, 1 / 13.9, [BV, SH, BH]),
    (8192, 4096, 3, 0.04, None, [BV, SH, BH]), (2048, 8000, 4, 0.3, 1 / 33.3, [BV, BH]),
    # a box shrink of 17 or more: the one-kernel chain
    (8192, 4096, 3, 0.0199, None, ["resize_streamg_u8"]),
]


@pytest.mark.parametrize("w,h,bands,scale,vscale,gates", RESIZE_CASES + RESIZE_BIG)
def test_resize_band_chain_vs_reference(w, h, bands, scale, vscale, gates, monkeypatch):
    """vips_resize at a scale that leaves a fractional reduce on both axes: shrinkv + reducev as one banded
    matrix-core kernel, shrinkh, reduceh (ops_resample.cpp resize_down_u8_stream) -- against the compiled
    reference, against the one-kernel chain and against the separate operations."""
    if (w * bands) % 4:
        pytest.skip("rows of whole dwords only")
    monkeypatch.setenv("VIPS_HIP_RESIZE_BAND_MIN", "0")
    lib = libvips_amd.lib
    src = helpers.lcg_image(w, h, bands, np.uint8, 5 + w)
    src[: h // 3, : w // 2] = 255
    src[h // 3: h // 2, w // 2:] = 0
    im = Image.new_from_array(src)
    kw = {} if vscale is None else {"vscale": vscale}
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    got = im.resize(scale, **kw).numpy()
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    chain = "resize:scale=%r" % scale + ("" if vscale is None else ",vscale=%r" % vscale)
    want = helpers.Ref.run_chain(chain, src)
    assert sorted(report) == sorted(gates), report
    assert got.shape == want.shape and np.array_equal(got, want)
    monkeypatch.setenv("VIPS_HIP_NO_RESIZE_BAND", "1")
    assert np.array_equal(got, im.resize(scale, **kw).numpy())


def test_resize_band_chain_where_it_pays():
    """Without VIPS_HIP_RESIZE_BAND_MIN: a single image of any size takes the three launches (the one-kernel chain
    walks a small image with a handful of blocks), a batch of small images stays with the one-kernel chain (64
    images a launch), a batch of large ones takes the three launches per image."""
    lib = libvips_amd.lib

    def gates(call):
        lib.vips_hip_gate_reset()
        lib.vips_hip_gate_enable(1)
        try:
            call()
            return sorted(libvips_amd.gate_report())
        finally:
            lib.vips_hip_gate_enable(0)
            lib.vips_hip_gate_reset()

    band = sorted([BH, BV, SH])
    for (w, h) in ((2048, 1000), (4096, 3000), (400, 300)):
        im = Image.new_from_array(helpers.lcg_image(w, h, 3, np.uint8, 9))
        assert gates(lambda: im.resize(0.123).numpy()) == band, (w, h)
    small = [Image.new_from_array(helpers.lcg_image(1024, 768, 3, np.uint8, 20 + k), interpretation="srgb") for k in range(6)]
    assert gates(lambda: libvips_amd.resize_sharpen_batch(small, 0.123, sharpen=False)) == ["resize_streamg_u8"]
    large = [Image.new_from_array(helpers.lcg_image(4096, 1024, 3, np.uint8, 30 + k), interpretation="srgb") for k in range(5)]
    outs = None

    def run_large():
        nonlocal outs
        outs = libvips_amd.resize_sharpen_batch(large, 0.123, sharpen=False)
    assert gates(run_large) == band
    for k in (0, 4):
        assert np.array_equal(outs[k].numpy(), large[k].resize(0.123).numpy())
