"""GPU: the integer convolutions on uchar -- the packed-byte kernels (conv_u8.hip) and the matrix-core
separable one (conv_u8_mfma.hip), which takes the separable masks first -- against the compiled reference
(or the port), whole images, bit for bit: the cases of tests/test_emul_conv_u8.py / test_emul_conv_u8_mfma.py,
where the same kernel bodies run on host fibers, plus BASELINE-sized images."""
import numpy as np
import pytest

import libvips_amd
from libvips_amd import Image
from tests import helpers
from tests.test_emul_conv_u8 import K3, K5, K37

pytestmark = pytest.mark.gpu

CASES = [
    ("blur", 1100, 70, 3, 2.0), ("blur", 332, 41, 1, 1.0), ("blur", 2071, 37, 4, 2.0),
    ("blur", 1028, 150, 3, 4.0), ("blur", 600, 130, 3, 6.0), ("blur", 532, 140, 3, 8.0),
    ("blur", 96, 33, 3, 2.0), ("blur", 640, 64, 3, 2.0, "flat"),
    ("sep", 700, 50, 3, ([1, -3, 9, -3, 1], 5)), ("sep", 260, 40, 1, ([5, 1, 5], 11)),
    ("conv", 1100, 70, 3, (K3, 8)), ("conv", 332, 41, 1, (K5, 256)), ("conv", 2071, 37, 4, (K3, 8)),
    ("conv", 1028, 50, 3, (K5, 256)), ("conv", 600, 45, 3, (K37, 30)), ("conv", 96, 33, 3, (K3, 8)),
    ("conv", 640, 64, 3, (K3, 8), "flat"),
    ("blur", 4096, 4096, 3, 2.0), ("blur", 4096, 3001, 3, 8.0), ("conv", 4096, 4096, 3, (K3, 8)),
    ("conv", 8192, 1027, 4, (K5, 256)), ("blur", 8192, 2048, 1, 3.0),
    ("blur", 8192, 8192, 3, 8.0), ("blur", 3000, 1000, 4, 8.0), ("blur", 300, 140, 3, 8.0), ("blur", 160, 230, 3, 8.0),
    ("sep", 172, 70, 3, (list(range(1, 18)) + list(range(16, 0, -1)), 289)),
    ("sep", 1200, 300, 3, ([-20, 40, 90, 140, 90, 40, -20], 360)),
]


def _reference(kind, src, arg):
    if kind == "blur":
        if helpers.have_ref():
            return helpers.Ref.run_chain("gaussblur:sigma=%r" % arg, src)
        return helpers.PortCC.gaussblur(src, arg)
    mask, scale = arg
    m = np.asarray(mask, dtype=np.float64)
    if kind == "sep":
        if helpers.have_ref():
            return helpers.Ref.run_mask("convsep", src, m[None, :], scale, 0.0, "precision=integer")
        return helpers.PortCC.convsep(src, m, scale=scale, precision="integer")
    if helpers.have_ref():
        return helpers.Ref.run_mask("conv", src, m, scale, 0.0, "precision=integer")
    return helpers.PortCC.conv(src, m, scale=scale, precision="integer")


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("env", [{}, {"VIPS_HIP_CONV_U8_SEG": "3", "VIPS_HIP_CONV_MFMA_SEG": "2"},
                                 {"VIPS_HIP_CONV_U8_MFMA": "0"}, {"VIPS_HIP_CONV_U8_MFMA": "0", "VIPS_HIP_CONV_U8_SEG": "3"}])
def test_conv_u8(case, env, monkeypatch):
    case = CASES[case]
    kind, w, h, bands, arg = case[:5]
    if len(env) > 1 and w * h > 2000 * 2000:
        pytest.skip("short segments on the small cases only")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sep_gate = "conv_u8_sep" if env.get("VIPS_HIP_CONV_U8_MFMA") == "0" else "conv_u8_mfma_sep"
    if kind == "sep" and sep_gate == "conv_u8_sep" and max(abs(v) for v in arg[0]) > 127:
        sep_gate = "convsep_u8_convi"  # (the packed-byte kernel's coefficients are signed bytes)
    src = helpers.lcg_image(w, h, bands, np.uint8, 7 + w)
    if len(case) > 5:
        src[: h // 2] = 255
        src[h // 2:, : w // 3] = 0
    im = Image.new_from_array(src)
    libvips_amd.lib.vips_hip_gate_reset()
    libvips_amd.lib.vips_hip_gate_enable(1)
    try:
        if kind == "blur":
            got = im.gaussblur(arg).numpy()
        elif kind == "sep":
            got = im.convsep(arg[0], scale=arg[1], precision="integer").numpy()
        else:
            got = im.conv(np.asarray(arg[0], dtype=np.float64), scale=arg[1], precision="integer").numpy()
        report = libvips_amd.gate_report()
    finally:
        libvips_amd.lib.vips_hip_gate_enable(0)
        libvips_amd.lib.vips_hip_gate_reset()
    conv_gate = "conv_u8_2d" if env.get("VIPS_HIP_CONV_U8_MFMA") == "0" else "conv_u8_mfma_2d"
    assert list(report) == [conv_gate if kind == "conv" else sep_gate], report
    want = _reference(kind, src, arg)
    assert got.shape == want.shape and got.dtype == want.dtype
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (len(bad), bad[:5])
    # ... and the older kernels agree
    monkeypatch.setenv("VIPS_HIP_NO_CONV_U8", "1")
    if kind == "blur":
        assert np.array_equal(got, im.gaussblur(arg).numpy())


CASES16 = [
    ("blur", 1100, 70, 3, 2.0), ("blur", 332, 41, 1, 1.0), ("blur", 2071, 37, 4, 2.0), ("blur", 1028, 150, 3, 4.0),
    ("blur", 600, 130, 2, 6.0), ("blur", 532, 140, 3, 8.0), ("blur", 96, 33, 3, 2.0), ("blur", 640, 64, 3, 2.0, "flat"),
    ("blur", 1024, 90, 4, 8.0, "flat"),
    ("sep", 700, 50, 3, ([1, -3, 9, -3, 1], 5)), ("sep", 260, 40, 1, ([5, 1, 5], 11)),
    ("sep", 172, 70, 3, (list(range(1, 18)) + list(range(16, 0, -1)), 289)),
    ("blur", 4096, 4096, 3, 2.0), ("blur", 4096, 3001, 4, 8.0), ("blur", 8192, 2048, 1, 3.0), ("blur", 8192, 4099, 3, 8.0),
]


@pytest.mark.parametrize("case", range(len(CASES16)))
@pytest.mark.parametrize("env", [{}, {"VIPS_HIP_CONV_MFMA_SEG": "2"}, {"VIPS_HIP_CONV_MFMA_NARROW": "1"}, {"VIPS_HIP_U16_FIN64": "1"}, {"VIPS_HIP_U16_FIN64": "0"}])
def test_conv_u16_separable_on_the_matrix_cores(case, env, monkeypatch):
    """vips_gaussblur / vips_convsep (precision integer) on ushort images: conv_u8_mfma_body.h with the image's bytes
    as 2 x bands planes, two exact products per sample and pass -- against the compiled reference and against the
    vector-ALU kernel it replaced."""
    case = CASES16[case]
    kind, w, h, bands, arg = case[:5]
    if env and w * h > 2000 * 2000:
        pytest.skip("the variants on the small cases only")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    src = helpers.lcg_image(w, h, bands, np.uint16, 7 + w)
    if len(case) > 5:
        src[: h // 2] = 65535
        src[h // 2:, : w // 3] = 0
    im = Image.new_from_array(src)

    def run():
        if kind == "blur":
            return im.gaussblur(arg).numpy()
        return im.convsep(arg[0], scale=arg[1], precision="integer").numpy()

    libvips_amd.lib.vips_hip_gate_reset()
    libvips_amd.lib.vips_hip_gate_enable(1)
    try:
        got = run()
        report = libvips_amd.gate_report()
    finally:
        libvips_amd.lib.vips_hip_gate_enable(0)
        libvips_amd.lib.vips_hip_gate_reset()
    assert list(report) == ["conv_u16_mfma_sep"], report
    want = _reference(kind, src, arg)
    assert got.shape == want.shape and got.dtype == want.dtype
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (len(bad), bad[:5])
    monkeypatch.setenv("VIPS_HIP_NO_CONV_U16_MFMA", "1")
    assert np.array_equal(got, run())
