"""BASELINE-size parity for configs 3, 4 and 5, WHOLE outputs, against the compiled reference
(oracle/_ref): what bench.py only samples.  C2's full-size test lives in
test_resample_gpu.py.  Skipped without the GPU or without oracle/_ref; the reference runs on the
box's host cores, band by band where the whole image would not fit comfortably in host memory (a
band is handed to the reference with the rows its taps reach above and below, so every output row
compared is the row the reference computes for the whole image).
"""
import os

import numpy as np
import pytest

from tests import helpers

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not helpers.have_ref(), reason="needs oracle/_ref")]


def _ulp(a, b):
    ai = np.ascontiguousarray(a).view(np.int32).astype(np.int64)
    bi = np.ascontiguousarray(b).view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return int(np.abs(ai - bi).max())


def _ctx():
    import torch

    import libvips_amd
    from bench import lcg_image_device

    libvips_amd.init(0)
    return torch, libvips_amd, lcg_image_device


def test_c3_full_size():
    """BASELINE configs[2]: vips_gaussblur(sigma 8) + vips_colourspace(sRGB -> Lab) on
    32768 x 32768 x 3 float, every row, bit for bit -- in the exact float mode and in the
    library's default mode alike: the fused blur + colour kernel always uses the reference's own
    arithmetic (a 1 ULP difference in the blur becomes a whole table step behind the sRGB decode,
    convsep_stream.hip MODE 3), so "within 1 ULP" is asserted and 0 is what must come out."""
    torch, vh, lcg = _ctx()
    from libvips_amd import Image, lib

    n = int(os.environ.get("FULL_C3_SIZE", "32768"))
    dev = torch.device("cuda", 0)
    src = lcg(torch, n, n, 3, 12345, dev).float()
    torch.cuda.synchronize()
    im = Image.new_from_tensor(src, interpretation="srgb")
    assert lib.vips_hip_get_exact_float() == 1  # conftest pins the suite to the exact mode
    exact = im.gaussblur_colourspace(8.0, "lab")
    lib.vips_hip_set_exact_float(0)
    try:
        fast = im.gaussblur_colourspace(8.0, "lab")
    finally:
        lib.vips_hip_set_exact_float(1)
    chain = "gaussblur:sigma=8;colourspace:space=lab"
    interp = helpers.INTERP["srgb"]
    halo, band = 14, 4096  # 29 taps
    worst_fast = 0
    for r0 in range(0, n, band):
        rows = min(band, n - r0)
        i0, i1 = max(r0 - halo, 0), min(r0 + rows + halo, n)
        want = helpers.Ref.run_chain(chain, src[i0:i1].cpu().numpy(), interp)[r0 - i0:r0 - i0 + rows]
        got = exact.extract_area(0, r0, n, rows).numpy()
        assert got.shape == want.shape and got.dtype == np.float32
        assert np.array_equal(got.view(np.int32), want.view(np.int32)), "exact mode, rows %d..%d" % (r0, r0 + rows)
        got = fast.extract_area(0, r0, n, rows).numpy()
        worst_fast = max(worst_fast, _ulp(got, want))
        del got, want
    assert worst_fast <= 1, "default mode: %d ULP" % worst_fast  # tolerance: 1 ULP (BASELINE.json north_star)
    assert worst_fast == 0


def test_c4_batch_full():
    """BASELINE configs[3], one GPU's share of the batch at full image size: 256 x (8192 x 8192 x 3
    uchar) through vips_resize(1/8) + vips_sharpen in 64-image launches; 18 thumbnails compared
    whole with the reference, on both sides of every chunk boundary (63 | 64, 127 | 128, 191 | 192)
    and at the ends."""
    torch, vh, lcg = _ctx()
    from libvips_amd import Image

    n = int(os.environ.get("FULL_C4_SIZE", "8192"))
    images = int(os.environ.get("FULL_C4_IMAGES", "256"))
    dev = torch.device("cuda", 0)
    store = torch.empty((images, n, n, 3), dtype=torch.uint8, device=dev)
    for k in range(images):
        lcg(torch, n, n, 3, 12345 + k, dev, out=store[k])
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    ims = [Image.new_from_tensor(store[k], interpretation="srgb") for k in range(images)]
    outs = vh.resize_sharpen_batch(ims, 0.125, threads=8)
    assert len(outs) == images
    picks = sorted(set(k for k in (0, 1, 31, 62, 63, 64, 65, 100, 126, 127, 128, 129, 190, 191, 192, 193, 254, 255)
                       if k < images) | {images - 1})
    chain = "resize:scale=0.125;sharpen:"
    interp = helpers.INTERP["srgb"]
    for k in picks:
        want = helpers.Ref.run_chain(chain, store[k].cpu().numpy(), interp)
        got = outs[k].numpy()
        assert got.shape == want.shape == (n // 8, n // 8, 3)
        assert np.array_equal(got, want), "thumbnail %d" % k


@pytest.mark.parametrize("mode", ["exact", "default"])
def test_c5_slab_full(mode):
    """BASELINE configs[4], one GPU's slab: the 31 x 31 float mask on rows 32768..40960 of the
    65536 x 65536 ushort image, its window carrying the 15-row halos a neighbour sends; ALL 8192
    rows against the reference: bit for bit in the exact mode, within 1 ULP in the default mode
    (fused multiply-adds)."""
    torch, vh, lcg = _ctx()
    from bench import c5_mask, c5_rows_device
    from libvips_amd import lib, sharding

    width = int(os.environ.get("FULL_C5_WIDTH", "65536"))
    im_height = width
    rows = max(width // 8, 64)
    dev = torch.device("cuda", 0)
    mask, scale = c5_mask(vh)
    plan = sharding.StripPlan(im_height, im_height, im_height // rows, sharding.conv_need(31, im_height))
    slab = (im_height // rows) // 2
    w0, w1 = plan.windows[slab]
    o0, o1 = plan.out_bounds[slab]
    window = c5_rows_device(torch, width, w0, w1 - w0, dev)
    torch.cuda.synchronize()
    lib.vips_hip_set_exact_float(1 if mode == "exact" else 0)
    try:
        out = sharding.conv_strip(window, w0, plan, slab, mask, scale=scale, precision="float")
        torch.cuda.synchronize()
    finally:
        lib.vips_hip_set_exact_float(1)
    assert tuple(out.shape[:2]) == (o1 - o0, width)
    band = 2048
    worst = 0
    for r0 in range(o0, o1, band):
        nrows = min(band, o1 - r0)
        host = window[r0 - 15 - w0:r0 + nrows + 15 - w0].cpu().numpy()
        want = helpers.Ref.run_mask("conv", host, mask, scale, 0.0, "precision=float")[15:15 + nrows]
        got = out[r0 - o0:r0 - o0 + nrows].cpu().numpy().reshape(want.shape)
        if mode == "exact":
            assert np.array_equal(got.view(np.int32), want.view(np.int32)), "rows %d..%d" % (r0, r0 + nrows)
        worst = max(worst, _ulp(got, want))
    assert worst <= (0 if mode == "exact" else 1), "%s mode: %d ULP" % (mode, worst)  # tolerance: 1 ULP
