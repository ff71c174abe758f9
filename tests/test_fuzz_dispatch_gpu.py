"""GPU: a seeded fuzz over the DISPATCH GUARDS of the streaming and matrix-core kernels of rounds 4 and 5
(reduceh_u8_packed, shrinkh_u8_stream, conv_u8 / conv_u8_mfma, conv_u16, reducev_u8_stream, reduce_band,
resample16): sizes round the dword / quad / 16-byte / strip / segment boundaries, 1 .. 4 bands, widths whose
rows are NOT whole dwords (the fall-back must be taken and still match), fractional and integer factors, masks
round the coefficient limits.  Whatever kernel the library picks, the result must equal the reference's (the
compiled reference when present, else the port) bit for bit; the report must name a kernel.  The host-fiber fuzz
of round 4 (tools/fuzz_emul.py) cannot see what only the device does: waitcnt, compiler traps, the LDS-DMA."""
import random

import numpy as np
import pytest

import libvips_amd
from libvips_amd import Image
from tests import helpers

pytestmark = pytest.mark.gpu

EDGES = [4, 8, 12, 16, 28, 32, 36, 60, 64, 68, 124, 128, 132, 252, 256, 260, 508, 512, 516, 1020, 1024, 1028]


def _size(rng, lo=8, hi=1400):
    # a boundary, a boundary +- a little, or anything
    r = rng.random()
    if r < 0.4:
        return max(lo, rng.choice(EDGES) + rng.choice([-3, -2, -1, 0, 0, 1, 2, 3]))
    if r < 0.6:
        return max(lo, rng.choice(EDGES) * rng.choice([1, 2, 3]))
    return rng.randrange(lo, hi)


def _reference(chain, src, port_call):
    if helpers.have_ref():
        return helpers.Ref.run_chain(chain, src)
    return port_call()


def _case(rng, kind):
    seed = rng.randrange(1 << 30)
    bands = rng.choice([1, 2, 3, 3, 4])
    if kind == "reduce":
        w, h = _size(rng, 16), _size(rng, 16)
        dt = rng.choice([np.uint8, np.uint8, np.uint16])
        src = helpers.lcg_image(w, h, bands, dt, seed)
        shrink = rng.choice([2.0, 4.0, 8.0, 1.3 + 9 * rng.random(), 2.0 + 6 * rng.random()])
        shrink = round(shrink, 3)
        kernel = rng.choice(["lanczos3", "lanczos3", "lanczos2", "cubic", "mitchell", "linear"])
        axis = rng.choice(["reduceh", "reducev", "reduce"])
        im = Image.new_from_array(src)
        if axis == "reduce":
            call = lambda: im.reduce(shrink, shrink, kernel=kernel)
            chain = "reduce:hshrink=%r,vshrink=%r,kernel=%s" % (shrink, shrink, kernel)
            port = lambda: helpers.Port.reduce(src, shrink, shrink, kernel)
        elif axis == "reduceh":
            call = lambda: im.reduceh(shrink, kernel=kernel)
            chain = "reduceh:hshrink=%r,kernel=%s" % (shrink, kernel)
            port = lambda: helpers.Port.reduce(src, shrink, 1.0, kernel)
        else:
            call = lambda: im.reducev(shrink, kernel=kernel)
            chain = "reducev:vshrink=%r,kernel=%s" % (shrink, kernel)
            port = lambda: helpers.Port.reduce(src, 1.0, shrink, kernel)
        return (kind, axis, w, h, bands, dt.__name__, shrink, kernel), call, lambda: _reference(chain, src, port)
    if kind == "shrink":
        w, h = _size(rng, 16), _size(rng, 16)
        dt = rng.choice([np.uint8, np.uint8, np.uint16])
        src = helpers.lcg_image(w, h, bands, dt, seed)
        hs, vs = rng.choice([1, 2, 3, 4, 5, 8, 12]), rng.choice([1, 2, 3, 4, 8])
        im = Image.new_from_array(src)
        chain = "shrink:hshrink=%d,vshrink=%d" % (hs, vs)
        return (kind, w, h, bands, dt.__name__, hs, vs), lambda: im.shrink(hs, vs), \
            lambda: _reference(chain, src, lambda: helpers.Port.shrink(src, hs, vs))
    if kind == "blur":
        w, h = _size(rng, 8), _size(rng, 8)
        src = helpers.lcg_image(w, h, bands, np.uint8, seed)
        sigma = rng.choice([0.8, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, 8.0])
        im = Image.new_from_array(src)
        return (kind, w, h, bands, sigma), lambda: im.gaussblur(sigma), \
            lambda: _reference("gaussblur:sigma=%r" % sigma, src, lambda: helpers.PortCC.gaussblur(src, sigma))
    if kind == "convsep":
        w, h = _size(rng, 8), _size(rng, 8)
        src = helpers.lcg_image(w, h, bands, np.uint8, seed)
        n = rng.choice([3, 5, 7, 9, 13, 17, 25, 33])
        top = rng.choice([9, 127, 128, 300, 2047])
        mask = [rng.randrange(-top // 4, top + 1) for _ in range(n)]
        mask[n // 2] = top
        scale = max(1, sum(mask))
        im = Image.new_from_array(src)

        def ref():
            m = np.asarray(mask, dtype=np.float64)
            if helpers.have_ref():
                return helpers.Ref.run_mask("convsep", src, m[None, :], scale, 0.0, "precision=integer")
            return helpers.PortCC.convsep(src, m, scale=scale, precision="integer")
        return (kind, w, h, bands, mask, scale), lambda: im.convsep(mask, scale=scale, precision="integer"), ref
    if kind == "resize":
        # round 6: vips_resize across the SIZE threshold that picks the chain of band kernels
        # (images of 8 MB and more, tools/band_threshold.py) or the one-kernel chain: images of 7 .. 9 MB
        bands = rng.choice([3, 3, 4, 1])
        target = rng.choice([7.0, 7.9, 8.0, 8.1, 9.0]) * (1 << 20)
        w = rng.randrange(900, 2600)
        w -= (w * bands) % 4 if rng.random() < 0.7 else 0  # (mostly rows of whole dwords: the streaming kernels' case)
        h = max(64, int(target / (w * bands)) + rng.choice([-1, 0, 1]))
        src = helpers.lcg_image(w, h, bands, np.uint8, seed)
        scale = round(rng.choice([0.125, 0.25, 0.05 + 0.4 * rng.random(), 0.05 + 0.4 * rng.random()]), 4)
        im = Image.new_from_array(src)
        return (kind, w, h, bands, scale), lambda: im.resize(scale), \
            lambda: _reference("resize:scale=%r" % scale, src, lambda: helpers.Port.resize(src, scale))
    if kind == "float":
        # the float streams of round 6 (resample_f32.hip) and what they decline: rows of odd element counts, images
        # too narrow, fractional factors, other kernels
        w, h = _size(rng, 16, 2200), _size(rng, 16, 600)
        bands = rng.choice([1, 3, 3, 4])
        src = (helpers.lcg_image(w, h, bands, np.uint8, seed).astype(np.float32) - 90.0) * np.float32(0.37)
        op = rng.choice(["reducev", "reduceh", "reduce", "shrinkv", "shrink"])
        im = Image.new_from_array(src)
        if op.startswith("shrink"):
            s1, s2 = rng.choice([2, 3, 4, 7, 8]), rng.choice([2, 3, 4, 8])
            if op == "shrinkv":
                return (kind, op, w, h, bands, s2), lambda: im.shrinkv(s2), \
                    lambda: _reference("shrinkv:vshrink=%d" % s2, src, lambda: helpers.Port.shrinkv(src, s2))
            return (kind, op, w, h, bands, s1, s2), lambda: im.shrink(s1, s2), \
                lambda: _reference("shrink:hshrink=%d,vshrink=%d" % (s1, s2), src, lambda: helpers.Port.shrink(src, s1, s2))
        shrink = round(rng.choice([2.0, 3.0, 4.0, 8.0, 8.0, 1.5 + 7 * rng.random()]), 3)
        kernel = rng.choice(["lanczos3", "lanczos3", "lanczos2", "cubic", "linear"])
        if op == "reducev":
            return (kind, op, w, h, bands, shrink, kernel), lambda: im.reducev(shrink, kernel=kernel), \
                lambda: _reference("reducev:vshrink=%r,kernel=%s" % (shrink, kernel), src, lambda: helpers.Port.reduce(src, 1.0, shrink, kernel))
        if op == "reduceh":
            return (kind, op, w, h, bands, shrink, kernel), lambda: im.reduceh(shrink, kernel=kernel), \
                lambda: _reference("reduceh:hshrink=%r,kernel=%s" % (shrink, kernel), src, lambda: helpers.Port.reduce(src, shrink, 1.0, kernel))
        return (kind, op, w, h, bands, shrink, kernel), lambda: im.reduce(shrink, shrink, kernel=kernel), \
            lambda: _reference("reduce:hshrink=%r,vshrink=%r,kernel=%s" % (shrink, shrink, kernel), src,
                               lambda: helpers.Port.reduce(src, shrink, shrink, kernel))
    if kind == "upsize":
        # bicubic enlargement of uchar by the column walk (upsize.hip), any bands, scales that move the window by
        # 0 / 1 input rows a step and (below 1 on one axis) by more
        w, h = _size(rng, 8, 400), _size(rng, 8, 300)
        src = helpers.lcg_image(w, h, bands, np.uint8, seed)
        hs = round(rng.choice([1.0 + 3 * rng.random(), 2.0, 2.5, 4.0]), 3)
        vs = round(rng.choice([hs, 1.0 + 3 * rng.random(), 1.0]), 3)
        im = Image.new_from_array(src)
        chain = "resize:scale=%r,vscale=%r,kernel=cubic" % (hs, vs)
        return (kind, w, h, bands, hs, vs), lambda: im.resize(hs, vscale=vs, kernel="cubic"), \
            lambda: _reference(chain, src, lambda: helpers.Port.resize(src, hs, vs, kernel="cubic"))
    if kind == "blur16":
        # ushort gaussblur across the matrix-core kernel's limits (masks of up to 33 taps: sigma 9 and more fall back)
        w, h = _size(rng, 8), _size(rng, 8)
        src = helpers.lcg_image(w, h, bands, np.uint16, seed)
        sigma = rng.choice([0.8, 1.5, 2.0, 4.0, 8.0, 8.9, 9.5, 11.0])
        im = Image.new_from_array(src)
        return (kind, w, h, bands, sigma), lambda: im.gaussblur(sigma), \
            lambda: _reference("gaussblur:sigma=%r" % sigma, src, lambda: helpers.PortCC.gaussblur(src, sigma))
    # a small 2-D mask on uchar / ushort
    w, h = _size(rng, 8), _size(rng, 8)
    dt = rng.choice([np.uint8, np.uint16])
    src = helpers.lcg_image(w, h, bands, dt, seed)
    mw, mh = rng.choice([1, 3, 5, 7, 9]), rng.choice([3, 5, 7]) if dt == np.uint8 else rng.choice([3, 5])
    if dt == np.uint16:
        mw = min(mw, 5)
    m = np.array([[rng.randrange(-3, 12) for _ in range(mw)] for _ in range(mh)], dtype=np.float64)
    scale = max(1, int(m.sum()))
    im = Image.new_from_array(src)

    def ref2():
        if helpers.have_ref():
            return helpers.Ref.run_mask("conv", src, m, scale, 0.0, "precision=integer")
        return helpers.PortCC.conv(src, m, scale=scale, precision="integer")
    return (kind, w, h, bands, dt.__name__, m.tolist(), scale), lambda: im.conv(m, scale=scale, precision="integer"), ref2


@pytest.mark.parametrize("kind,count,seed", [("reduce", 60, 501), ("shrink", 30, 502), ("blur", 40, 503),
                                             ("convsep", 30, 504), ("conv", 30, 505), ("resize", 14, 506),
                                             ("float", 40, 507), ("upsize", 24, 508), ("blur16", 24, 509)])
def test_fuzz_dispatch(kind, count, seed):
    rng = random.Random(seed)
    lib = libvips_amd.lib
    kernels = {}
    for _ in range(count):
        desc, call, ref = _case(rng, kind)
        lib.vips_hip_gate_reset()
        lib.vips_hip_gate_enable(1)
        try:
            got = call().numpy()
            report = libvips_amd.gate_report()
        finally:
            lib.vips_hip_gate_enable(0)
            lib.vips_hip_gate_reset()
        assert report, desc
        for k in report:
            kernels[k] = kernels.get(k, 0) + 1
        want = ref()
        assert got.shape == want.shape and got.dtype == want.dtype, (desc, got.shape, want.shape)
        if got.dtype == np.float32:  # (bit for bit, NaN-safe: the float paths are the reference's operations in its order)
            got, want = got.view(np.int32), want.view(np.int32)
        bad = np.argwhere(got != want)
        assert len(bad) == 0, (desc, dict(report), len(bad), bad[:4].tolist())
    # the sweep must have reached the fast kernels AND their fall-backs
    assert len(kernels) >= {"upsize": 1, "blur16": 2}.get(kind, 3), kernels
