"""GPU: the packed-byte vips_shrinkh on uchar (libvips_amd/csrc/shrinkh_u8.hip) against the compiled
reference, whole image, bit for bit -- the cases of tests/test_emul_shrinkh_u8.py on the device, plus
larger images and the two-axis vips_shrink that now ends in it."""
import numpy as np
import pytest

import libvips_amd
from libvips_amd import Image
from tests import helpers
from tests.test_emul_shrinkh_u8 import CASES

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not helpers.have_ref(), reason="oracle/_ref missing")]


@pytest.mark.parametrize("w,h,bands,hs,ceil,gate", CASES + [(8192, 300, 3, 4, 0, "shrinkh_u8_stream"),
                                                            (16384, 64, 4, 8, 0, "shrinkh_u8_stream")])
def test_shrinkh_u8_vs_reference(w, h, bands, hs, ceil, gate):
    lib = libvips_amd.lib
    src = helpers.lcg_image(w, h, bands, np.uint8, 11 + w)
    src[: h // 3, : w // 2] = 255
    src[h // 3: h // 2, w // 2:] = 0
    im = Image.new_from_array(src)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    got = im.shrinkh(hs, ceil=bool(ceil)).numpy()
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    want = helpers.Ref.run_chain("shrinkh:hshrink=%d%s" % (hs, ",ceil=true" if ceil else ""), src)
    assert list(report) == [gate], report
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.array_equal(got, want)


def test_shrink_both_axes():
    src = helpers.lcg_image(4096, 1000, 3, np.uint8, 3)
    got = Image.new_from_array(src).shrink(4, 4).numpy()
    want = helpers.Ref.run_chain("shrink:hshrink=4,vshrink=4", src)
    assert np.array_equal(got, want)
