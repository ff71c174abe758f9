"""GPU: the packed-byte vips_reduceh on uchar (libvips_amd/csrc/reduceh_u8.hip) against the compiled
reference, whole image, bit for bit -- the cases of tests/test_emul_reduceh_u8.py on the device, plus
larger images and the two-axis vips_reduce on 3 bands that now ends in it."""
import numpy as np
import pytest

import libvips_amd
from libvips_amd import Image
from tests import helpers
from tests.test_emul_reduceh_u8 import CASES

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not helpers.have_ref(), reason="oracle/_ref missing")]


@pytest.mark.parametrize("w,h,bands,shrink,kernel,gate", CASES + [(8192, 300, 3, 8.0, "lanczos3", "reduceh_u8_packed"),
                                                                  (16384, 70, 4, 4.0, "lanczos3", "reduceh_u8_packed")])
def test_reduceh_u8_vs_reference(w, h, bands, shrink, kernel, gate, monkeypatch):
    monkeypatch.setenv("VIPS_HIP_REDUCE_BAND", "0")  # (this kernel's fall-backs; the banded one: test_reduce_band_gpu.py)
    monkeypatch.setenv("VIPS_HIP_NO_REDUCEH3", "1")  # (... and three bands by 8 on the matrix cores: below)
    lib = libvips_amd.lib
    src = helpers.lcg_image(w, h, bands, np.uint8, 11 + w)
    src[: h // 3, : w // 2] = 255
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


@pytest.mark.parametrize("one_kernel", [True, False])
def test_reduce_rgb_both_axes(one_kernel, monkeypatch):
    """vips_reduce(8, 8) on RGB against the compiled reference: the one-kernel form (round 6) and reducev, then the
    packed reduceh ($VIPS_HIP_NO_FUSED3: what a width that is not a multiple of 8 still takes)."""
    if not one_kernel:
        monkeypatch.setenv("VIPS_HIP_NO_FUSED3", "1")
        monkeypatch.setenv("VIPS_HIP_NO_REDUCEH3", "1")
    src = helpers.lcg_image(4096, 2048, 3, np.uint8, 3)
    lib = libvips_amd.lib
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    got = Image.new_from_array(src).reduce(8.0, 8.0).numpy()
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    want = helpers.Ref.run_chain("reduce:hshrink=8.0,vshrink=8.0", src)
    assert ("reduce_fused_u8x3_mfma" if one_kernel else "reduceh_u8_packed") in report, report
    assert np.array_equal(got, want)


@pytest.mark.parametrize("oht", [0, 8, 128])
@pytest.mark.parametrize("w,h", [(4096, 21), (640, 100), (680, 9), (8192, 300), (1280, 77), (16384, 1), (3000, 130), (24, 5)])
def test_reduceh_rgb_by_8_on_the_matrix_cores(w, h, oht, monkeypatch):
    """Round 6: vips_reduceh(8) on three bands as the one-kernel reduce's horizontal walk (reduceh_u8x3_mfma: eight
    rows a batch copied into LDS with whole-line loads, lane (row, segment, band) on 24-byte groups) -- one tile
    wide with both image edges in a row, several tiles, a last tile of fewer than 80 outputs, heights the batches and
    the tiles do not divide, tiles of 8 and of 128 rows -- against the compiled reference and the packed kernel."""
    lib = libvips_amd.lib
    src = helpers.lcg_image(w, h, 3, np.uint8, 13 + w)
    src[: h // 3, : w // 2] = 255
    src[h // 3: h // 2, w // 2:] = 0
    im = Image.new_from_array(src)
    if oht:
        monkeypatch.setenv("VIPS_HIP_REDUCEH3_OHT", str(oht))
    monkeypatch.setenv("VIPS_HIP_REDUCEH3_MIN", "0")  # (by default only images of ~100 MB and more take it)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    got = im.reduceh(8.0, kernel="lanczos3").numpy()
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    assert list(report) == ["reduceh_u8x3_mfma"], report
    want = helpers.Ref.run_chain("reduceh:hshrink=8.0,kernel=lanczos3", src)
    assert got.shape == want.shape and np.array_equal(got, want)
    monkeypatch.setenv("VIPS_HIP_NO_REDUCEH3", "1")
    assert np.array_equal(got, im.reduceh(8.0, kernel="lanczos3").numpy())
