"""GPU: the ushort streaming resample kernels (resample16.hip) against the compiled reference (or
the port), whole images, bit for bit: the cases of tests/test_emul_resample16.py plus BASELINE-sized
images."""
import numpy as np
import pytest

import libvips_amd
from libvips_amd import Image
from tests import helpers
from tests.helpers import Port
from tests.test_emul_resample16 import CASES

pytestmark = pytest.mark.gpu

BIG = [
    ("reduce", 4096, 4096, 4, (8.0, 8.0), ["reducev_u16_stream", "reduceh_u16_lds"]),
    ("reduce", 4096, 3001, 3, (7.3, 7.3), ["reducev_u16_stream", "reduceh_u16_lds"]),
    ("shrink", 8192, 2051, 4, (4.0, 4.0), ["shrinkv_u16_stream", "shrinkh_u16_stream"]),
    ("reducev", 16384, 2048, 4, (8.0,), ["reducev_u16_stream"]),
]


@pytest.mark.parametrize("case", range(len(CASES) + len(BIG)))
@pytest.mark.parametrize("seg", [None, "5"])
def test_ushort_streaming(case, seg, monkeypatch):
    monkeypatch.setenv("VIPS_HIP_REDUCE_BAND", "0")  # (the vector-ALU kernels; the matrix-core ones: test_reduce_band_gpu.py)
    monkeypatch.setenv("VIPS_HIP_NO_SHRINKBOX16", "1")  # (vips_shrink as its two kernels; in one: test_shrinkbox16)
    op, w, h, bands, args, gates = (CASES + BIG)[case]
    if seg:
        if case >= len(CASES) or op not in ("reducev", "shrinkv", "reduce"):
            pytest.skip("short segments: the vertical kernels, small cases")
        monkeypatch.setenv("VIPS_HIP_R16_SEG", seg)
    src = helpers.lcg_image(w, h, bands, np.uint16, 11 + w)
    src[: h // 3, : w // 2] = 65535
    src[h // 3: h // 2, w // 2:] = 0
    im = Image.new_from_array(src)
    libvips_amd.lib.vips_hip_gate_reset()
    libvips_amd.lib.vips_hip_gate_enable(1)
    try:
        got = getattr(im, op)(*args).numpy()
        report = libvips_amd.gate_report()
    finally:
        libvips_amd.lib.vips_hip_gate_enable(0)
        libvips_amd.lib.vips_hip_gate_reset()
    assert sorted(report) == sorted(gates), report
    if helpers.have_ref():
        names = {"reducev": "vshrink", "reduceh": "hshrink", "shrinkv": "vshrink", "shrinkh": "hshrink"}
        if op in ("reduce", "shrink"):
            chain = "%s:hshrink=%r,vshrink=%r" % (op, args[0], args[1])
        else:
            chain = "%s:%s=%r" % (op, names[op], args[0])
        if op.startswith("reduce") and len(args) > (2 if op == "reduce" else 1):
            chain += ",kernel=" + args[-1]
        want = helpers.Ref.run_chain(chain, src)
    else:
        want = getattr(Port, op)(src, *args)
    assert got.shape == want.shape and got.dtype == want.dtype
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (len(bad), bad[:5])
    monkeypatch.setenv("VIPS_HIP_NO_STREAM16", "1")
    assert np.array_equal(got, getattr(im, op)(*args).numpy())


@pytest.mark.parametrize("bands", [1, 2, 3, 4])
def test_shrinkbox16(bands, monkeypatch):
    """Round 6: vips_shrink on ushort in ONE kernel (shrinkbox16: a thread per output pixel, the vertical sums of its
    columns rounded as shrinkv rounds, their sum as shrinkh does -- shrink.c:77-119, shrinkv.c:233-244,
    shrinkh.c:98-112): factors whose boxes are and are not whole 16-byte groups, sizes the factors do not divide
    (floor and ceil), against the port and against the pair of kernels."""
    import libvips_amd
    from libvips_amd import Image

    lib = libvips_amd.lib
    for (w, h, hs, vs, ceil) in ((512, 256, 4, 4, False), (515, 259, 4, 4, True), (300, 200, 2, 3, False),
                                 (301, 203, 3, 2, True), (640, 96, 8, 5, False), (129, 67, 6, 7, True)):
        src = helpers.lcg_image(w, h, bands, np.uint16, 120 + hs)
        im = Image.new_from_array(src)
        lib.vips_hip_gate_reset()
        lib.vips_hip_gate_enable(1)
        try:
            got = im.shrink(hs, vs, ceil=ceil).numpy()
            report = libvips_amd.gate_report()
        finally:
            lib.vips_hip_gate_enable(0)
            lib.vips_hip_gate_reset()
        assert list(report) == ["shrinkbox_u16"], report
        monkeypatch.setenv("VIPS_HIP_NO_SHRINKBOX16", "1")
        old = im.shrink(hs, vs, ceil=ceil).numpy()
        monkeypatch.delenv("VIPS_HIP_NO_SHRINKBOX16")
        want = helpers.Port.shrink(src, hs, vs, ceil=ceil)
        assert got.shape == want.shape and np.array_equal(got, want), (w, h, hs, vs, ceil)
        assert np.array_equal(got, old)
