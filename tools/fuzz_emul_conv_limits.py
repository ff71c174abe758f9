#!/usr/bin/env python3
"""CPU fuzz of the integer convolutions' DISPATCH LIMITS on host fibers (tests/emul) against the compiled reference:
masks whose coefficients, sums, scales and offsets sit on and beyond what the matrix-core and packed-byte kernels
accept (an exact half: |c| < 2048; numerators below 2^24; 1 <= scale <= 8000; offset 0; folded edge coefficients),
on uchar and ushort, saturated / half-saturated / noise images.  Whatever kernel the dispatcher picks -- the fast one
inside its limits, the general `convi` beyond them -- must give the reference's pixels.

usage:  LD_PRELOAD=tests/mock_hip/_build/libmockhip.so VIPS_HIP_LIBRARY=tests/emul/_build/libvipship_emul.so \
        python tools/fuzz_emul_conv_limits.py [seed] [cases] [wild|edge]
  wild: coefficients up to 70000, any scale and offset (most cases must be REFUSED by the fast kernels)
  edge: |c| <= 2047 / 2048, sums of magnitudes around 65793 (= 2^24 / 255), scales around 8000, offset 0
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import libvips_amd  # noqa: E402
from libvips_amd import Image  # noqa: E402
from tests import helpers  # noqa: E402

libvips_amd.init(0)
lib = libvips_amd.lib


def wild(seed, cases):
    rng = np.random.default_rng(seed)
    bad = 0
    ran = {}
    for case in range(cases):
        dt = np.uint16 if rng.random() < 0.4 else np.uint8
        bands = int(rng.choice([1, 3, 4]))
        w = int(rng.integers(8, 200)); h = int(rng.integers(4, 120))
        src = helpers.lcg_image(w, h, bands, dt, 7000 + case)
        mode = rng.choice(["sat", "noise", "zero"])
        if mode == "sat":
            src[:] = np.iinfo(dt).max
        im = Image.new_from_array(src)
        sep = rng.random() < 0.5
        big = int(rng.choice([40, 127, 255, 1000, 2047, 2049, 5000, 70000]))
        lib.vips_hip_gate_reset(); lib.vips_hip_gate_enable(1)
        try:
            if sep:
                n = int(rng.choice([3, 5, 9, 15, 33, 35]))
                mask = rng.integers(-big, big + 1, size=n).astype(np.float64)
                if rng.random() < 0.5: mask = np.abs(mask)
                scale = int(max(1, abs(mask.sum()))) if rng.random() < 0.6 else int(rng.integers(1, 100000))
                off = float(rng.choice([0, 0, 3, -7, 128]))
                try:
                    got = im.convsep(list(mask), scale=scale, offset=off, precision="integer").numpy()
                except Exception as e:
                    print("refused sep", n, big, str(e)[:80]); continue
                want = lambda: helpers.Ref.run_mask("convsep", src, mask[None, :], scale, off, "precision=integer")
                desc = ("sep", n, big, scale, off)
            else:
                mw, mh = int(rng.choice([1, 3, 5, 7, 9])), int(rng.choice([1, 3, 5, 9]))
                if mw == 1 and mh == 1: mw = 3
                mask = rng.integers(-big, big + 1, size=(mh, mw)).astype(np.float64)
                if rng.random() < 0.5: mask = np.abs(mask)
                scale = int(max(1, abs(mask.sum()))) if rng.random() < 0.6 else int(rng.integers(1, 100000))
                off = float(rng.choice([0, 0, 3, -7, 128]))
                try:
                    got = im.conv(mask, scale=scale, offset=off, precision="integer").numpy()
                except Exception as e:
                    print("refused 2d", (mw, mh), big, str(e)[:80]); continue
                want = lambda: helpers.Ref.run_mask("conv", src, mask, scale, off, "precision=integer")
                desc = ("2d", mw, mh, big, scale, off)
        finally:
            report = list(libvips_amd.gate_report()); lib.vips_hip_gate_enable(0)
        ref = want()
        ok = got.shape == ref.shape and got.dtype == ref.dtype and np.array_equal(got, ref)
        for g in report: ran[g] = ran.get(g, 0) + 1
        if not ok:
            bad += 1
            d = np.argwhere(got != ref) if got.shape == ref.shape else []
            print("MISMATCH", desc, dt.__name__, (w, h, bands), mode, report, len(d), d[:3] if len(d) else (got.shape, ref.shape, got.dtype, ref.dtype))
    print("cases done, bad =", bad, ran)
    return bad


def edge(seed, cases):
    rng = np.random.default_rng(seed)
    bad = 0
    ran = {}
    for case in range(cases):
        dt = np.uint16 if rng.random() < 0.4 else np.uint8
        bands = int(rng.choice([1, 3, 4]))
        w = int(rng.integers(32, 140)); h = int(rng.integers(8, 100))
        if rng.random() < 0.5: w = (w + 3) & ~3
        src = helpers.lcg_image(w, h, bands, dt, 7000 + case)
        mode = rng.choice(["sat", "noise", "half"])
        if mode == "sat":
            src[:] = np.iinfo(dt).max
        elif mode == "half":
            src[:, : w // 2] = np.iinfo(dt).max
        im = Image.new_from_array(src)
        sep = rng.random() < 0.6
        target = int(rng.choice([2000, 20000, 60000, 65000, 65700, 65792, 65793, 66000, 80000]))
        lib.vips_hip_gate_reset(); lib.vips_hip_gate_enable(1)
        try:
            if sep:
                n = int(rng.choice([3, 5, 9, 15, 33]))
                shape = (n,)
            else:
                mw, mh = int(rng.choice([3, 5, 9, 17, 33])), int(rng.choice([1, 3, 5, 9]))
                shape = (mh, mw)
            cnt = int(np.prod(shape))
            # magnitudes that sum to the target, each below 2048 where that is possible
            mags = rng.dirichlet(np.ones(cnt)) * target
            mags = np.minimum(np.floor(mags), 2047 if rng.random() < 0.8 else 2048)
            mags.flat[int(rng.integers(cnt))] += 0
            signs = np.where(rng.random(cnt) < (0.0 if rng.random() < 0.5 else 0.3), -1.0, 1.0)
            mask = (mags * signs).reshape(shape).astype(np.float64)
            ssum = int(abs(mask.sum()))
            scale = int(rng.choice([max(1, ssum), 7999, 8000, 8001, max(1, min(8000, ssum)), 1, 255]))
            if sep:
                got = im.convsep(list(mask), scale=scale, precision="integer").numpy()
                want = lambda: helpers.Ref.run_mask("convsep", src, mask[None, :], scale, 0.0, "precision=integer")
            else:
                got = im.conv(mask, scale=scale, precision="integer").numpy()
                want = lambda: helpers.Ref.run_mask("conv", src, mask, scale, 0.0, "precision=integer")
            desc = (shape, target, int(np.abs(mask).sum()), scale)
        finally:
            report = list(libvips_amd.gate_report()); lib.vips_hip_gate_enable(0)
        ref = want()
        ok = got.shape == ref.shape and got.dtype == ref.dtype and np.array_equal(got, ref)
        for g in report: ran[g] = ran.get(g, 0) + 1
        if not ok:
            bad += 1
            d = np.argwhere(got != ref) if got.shape == ref.shape else []
            print("MISMATCH", desc, dt.__name__, (w, h, bands), mode, report, len(d), d[:3] if len(d) else (got.shape, ref.shape, got.dtype, ref.dtype))
    print("cases done, bad =", bad, ran)
    return bad


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    mode = sys.argv[3] if len(sys.argv) > 3 else "edge"
    sys.exit(1 if {"wild": wild, "edge": edge}[mode](seed, cases) else 0)
