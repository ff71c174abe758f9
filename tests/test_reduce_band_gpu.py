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
