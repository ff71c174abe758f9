"""GPU: the streaming integer convolution on ushort (libvips_amd/csrc/conv_u16.hip) against the compiled
reference, whole image, bit for bit -- the cases of tests/test_emul_conv_u16.py on the device, plus
larger images."""
import numpy as np
import pytest

import libvips_amd
from libvips_amd import Image
from tests import helpers
from tests.test_emul_conv_u16 import CASES, K3, K5

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not helpers.have_ref(), reason="oracle/_ref missing")]


@pytest.mark.parametrize("w,h,bands,mask,scale,gate,flat", CASES + [(8192, 400, 3, K3, 8, "conv_u16_2d", 0),
                                                                    (4096, 300, 4, K5, 256, "conv_u16_2d", 1)])
def test_conv_u16_vs_reference(w, h, bands, mask, scale, gate, flat):
    lib = libvips_amd.lib
    src = helpers.lcg_image(w, h, bands, np.uint16, 7 + w)
    if flat:
        src[: h // 2] = 65535
        src[h // 2:, : w // 3] = 0
    m = np.asarray(mask, dtype=np.float64)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    got = Image.new_from_array(src).conv(m, scale=scale, precision="integer").numpy()
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    want = helpers.Ref.run_mask("conv", src, m, scale, 0.0, "precision=integer")
    assert list(report) == [gate], report
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.array_equal(got, want)
