#!/usr/bin/env python3
"""Randomised parity sweep (device vs oracle/port) beyond the fixed test cases: random sizes,
bands, formats and parameters for reduce / resize (down and up) / gaussblur / conv / shrink.
usage: python tools/fuzz_gpu.py [seconds] [seed] [kind].  Prints every mismatch; exit code 1 if any."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import libvips_amd  # noqa: E402
from libvips_amd import Image  # noqa: E402
from tests import helpers  # noqa: E402
from tests.helpers import Port, PortCC  # noqa: E402

INT_TYPES = [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32]
ALL_TYPES = INT_TYPES + [np.float32]
KERNELS = ["nearest", "linear", "cubic", "mitchell", "lanczos2", "lanczos3", "mks2013", "mks2021"]


def same(a, b):
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype == np.float32:
        return np.array_equal(a.view(np.int32), b.view(np.int32)) or np.array_equal(a, b, equal_nan=True)
    return np.array_equal(a, b)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    libvips_amd.init(0)
    # bit-for-bit comparisons: the library's exact float mode (the default lets the large float
    # convolutions on integer images use fused multiply-adds, <= 1 ULP from the reference)
    libvips_amd.lib.vips_hip_set_exact_float(1)
    t0 = time.time()
    n = bad = 0
    while time.time() - t0 < budget:
        kind = rng.choice(["reduce8", "reduce", "resize", "upsize", "gaussblur", "conv", "shrink", "thumb",
                           "approx", "resize2k", "sharpen"])
        if len(sys.argv) > 3:
            kind = sys.argv[3]
        seed = rng.randrange(1 << 30)
        try:
            if kind == "reduce8":
                w, h = rng.randrange(8, 3000), rng.randrange(8, 2200)
                b = rng.choice([4, 4, 3, 1, 2])
                if rng.random() < 0.5:
                    w = (w + 7) // 8 * 8  # rows of whole 8-byte columns: the streaming vertical kernel
                src = helpers.lcg_image(w, h, b, np.uint8, seed)
                s = rng.choice([2, 4, 8, 8, 8])
                got = Image.new_from_array(src).reduce(s, s, kernel="lanczos3").numpy()
                want = Port.reduce(src, s, s, "lanczos3")
                desc = (kind, w, h, b, s)
            elif kind == "reduce":
                w, h, b = rng.randrange(1, 400), rng.randrange(1, 400), rng.randrange(1, 6)
                dt = rng.choice(ALL_TYPES)
                hs, vs = 1 + rng.random() * 6, 1 + rng.random() * 6
                k = rng.choice(KERNELS)
                src = helpers.lcg_image(w, h, b, dt, seed)
                got = Image.new_from_array(src).reduce(hs, vs, kernel=k).numpy()
                want = Port.reduce(src, hs, vs, k)
                desc = (kind, w, h, b, dt.__name__, hs, vs, k)
            elif kind == "resize2k":
                # uchar by 1 / (2 k) on rows of >= 2 KB: the one-kernel chain (resize_stream.hip), and
                # the one-kernel sharpen behind it on sRGB
                b = rng.choice([1, 2, 3, 3, 4])
                w, h = rng.randrange(2048 // b + 1, 4300), rng.randrange(8, 1500)
                k1, k2 = rng.choice([1, 2, 3, 4, 5, 6, 8]), rng.choice([1, 2, 3, 4, 5, 6, 8])
                if rng.random() < 0.6:
                    k2 = k1
                src = helpers.lcg_image(w, h, b, np.uint8, seed)
                sharpen = b == 3 and rng.random() < 0.5
                im = Image.new_from_array(src, interpretation="srgb" if sharpen else "multiband")
                k = rng.choice(["lanczos3", "lanczos3", "linear", "cubic", "mitchell", "lanczos2", "mks2013", "mks2021"])
                out = im.resize(0.5 / k1, vscale=0.5 / k2, kernel=k)
                want = Port.resize(src, 0.5 / k1, 0.5 / k2, kernel=k)
                if sharpen:
                    out = out.sharpen()
                    want = PortCC.sharpen(want, "srgb")
                got = out.numpy()
                desc = (kind, w, h, b, k1, k2, k, sharpen)
            elif kind in ("resize", "upsize"):
                w, h, b = rng.randrange(2, 300), rng.randrange(2, 300), rng.randrange(1, 5)
                dt = rng.choice(ALL_TYPES)
                lo, hi = (0.05, 1.0) if kind == "resize" else (0.5, 5.0)
                hs, vs = lo + rng.random() * (hi - lo), lo + rng.random() * (hi - lo)
                k = rng.choice(KERNELS)
                src = helpers.lcg_image(w, h, b, dt, seed)
                got = Image.new_from_array(src).resize(hs, vscale=vs, kernel=k).numpy()
                want = Port.resize(src, hs, vs, kernel=k)
                desc = (kind, w, h, b, dt.__name__, hs, vs, k)
            elif kind == "gaussblur":
                w, h, b = rng.randrange(1, 1600), rng.randrange(1, 500), rng.randrange(1, 5)
                dt = rng.choice([np.uint8, np.uint16, np.int16, np.float32])
                sigma = 0.3 + rng.random() * 9
                prec = rng.choice(["integer", "float"])
                src = helpers.lcg_image(w, h, b, dt, seed)
                got = Image.new_from_array(src).gaussblur(sigma, precision=prec).numpy()
                want = PortCC.gaussblur(src, sigma, precision=prec)
                desc = (kind, w, h, b, dt.__name__, sigma, prec)
            elif kind == "conv":
                w, h, b = rng.randrange(1, 300), rng.randrange(1, 300), rng.randrange(1, 4)
                dt = rng.choice(INT_TYPES + [np.float32])
                mw, mh = rng.randrange(1, 34), rng.randrange(1, 6)
                mask = np.round(np.array([[rng.gauss(0, 3) for _ in range(mw)] for _ in range(mh)]), 2)
                if rng.random() < 0.3:
                    mask[rng.randrange(mh), rng.randrange(mw)] = 0.0
                scale = rng.choice([1.0, 2.5, 7.0])
                prec = rng.choice(["integer", "float"])
                src = helpers.lcg_image(w, h, b, dt, seed)
                got = Image.new_from_array(src).conv(mask, scale=scale, offset=1.0, precision=prec).numpy()
                want = PortCC.conv(src, mask, scale, 1.0, prec)
                desc = (kind, w, h, b, dt.__name__, mw, mh, scale, prec)
            elif kind == "approx":
                # precision=approximate: conva / convasep (float sums are exact for these pixels,
                # double convasep is tile-order dependent in the reference: left out)
                w, h, b = rng.randrange(1, 260), rng.randrange(1, 200), rng.randrange(1, 4)
                dt = rng.choice(INT_TYPES + [np.float32])
                layers, cluster = rng.randrange(1, 16), rng.randrange(1, 6)
                src = helpers.lcg_image(w, h, b, dt, seed)
                if rng.random() < 0.5:
                    mw, mh = rng.randrange(1, 14), rng.randrange(1, 14)
                    mask = np.array([[float(rng.randrange(-4, 16)) for _ in range(mw)] for _ in range(mh)])
                    mask[rng.randrange(mh), rng.randrange(mw)] = 17.0
                    scale, offset = float(rng.randrange(1, 50)), float(rng.randrange(-4, 5))
                    got = Image.new_from_array(src).conva(mask, scale, offset, layers, cluster).numpy()
                    want = PortCC.conva(src, mask, scale, offset, layers, cluster)
                    desc = (kind, "conva", w, h, b, dt.__name__, mw, mh, layers, cluster)
                else:
                    sigma = 0.5 + rng.random() * 9
                    got = Image.new_from_array(src).gaussblur(sigma, precision="approximate").numpy()
                    want = PortCC.gaussblur(src, sigma, precision="approximate")
                    desc = (kind, "gaussblur", w, h, b, dt.__name__, sigma)
            elif kind == "sharpen":
                # round 6: vips_sharpen on sRGB uchar over smooth / noisy / striped content (the skip kernel's list
                # from empty to full), sizes either side of the 4 Mpixel switch to the adaptive pair now and then
                big = rng.random() < 0.15
                w = rng.randrange(2048, 2500) if big else rng.randrange(3, 900)
                h = rng.randrange(2048, 2300) if big else rng.randrange(3, 700)
                noise = helpers.lcg_image(w, h, 3, np.uint8, seed)
                small = helpers.lcg_image(w // 8 + 2, h // 8 + 2, 3, np.uint8, seed + 1).astype(np.float32)
                smooth = np.kron(small, np.ones((8, 8, 1), np.float32))[:h, :w]
                smooth = (smooth + np.roll(smooth, 1, 1) + np.roll(smooth, 2, 1) + np.roll(smooth, 1, 0)) / 4.0
                src = smooth.astype(np.uint8)
                cut = rng.randrange(0, w + 1)
                src[:, cut:] = noise[:, cut:]
                if rng.random() < 0.3:
                    src[rng.randrange(0, h):, : w // 2] = (np.arange(h) // 5 % 2 * 180 + 30).astype(np.uint8)[-1, None, None]
                params = {}
                if rng.random() < 0.5:
                    params = dict(sigma=rng.choice([0.3, 0.5, 0.8, 1.0, 1.4]), x1=rng.choice([0.5, 1.0, 2.0, 3.0]),
                                  m1=rng.choice([0.0, 0.0, 0.3]), m2=rng.choice([1.0, 3.0, 5.0]))
                got = Image.new_from_array(src, interpretation="srgb").sharpen(**params).numpy()
                want = PortCC.sharpen(src, "srgb", **params)
                desc = (kind, w, h, cut, params)
            elif kind == "shrink":
                w, h, b = rng.randrange(1, 500), rng.randrange(1, 500), rng.randrange(1, 5)
                dt = rng.choice(ALL_TYPES)
                hs, vs = rng.randrange(1, 9), rng.randrange(1, 9)
                src = helpers.lcg_image(w, h, b, dt, seed)
                ceil = rng.random() < 0.5
                got = Image.new_from_array(src).shrink(hs, vs, ceil=ceil).numpy()
                want = Port.shrink(src, hs, vs, ceil)
                desc = (kind, w, h, b, dt.__name__, hs, vs, ceil)
            else:
                w, h, b = rng.randrange(40, 900), rng.randrange(40, 700), rng.choice([3, 4])
                src = helpers.lcg_image(w, h, b, np.uint8, seed)
                target = rng.randrange(8, 400)
                got = Image.new_from_array(src, interpretation="srgb").thumbnail_image(target).numpy()
                want = PortCC.thumbnail_image(src, "srgb", target)
                desc = (kind, w, h, b, target)
            n += 1
            if not same(got, want):
                bad += 1
                print("MISMATCH", desc, "seed", seed, flush=True)
        except libvips_amd.VipsHipError as exc:
            # loud refusals are fine only where the port refuses too
            try:
                _ = want  # noqa: F841
                print("DEVICE REFUSED", desc, str(exc)[:100], flush=True)
                bad += 1
            except NameError:
                pass
        except Exception as exc:  # port-side refusal of an out-of-scope case
            if "unsupported" not in repr(exc) and "shrunk" not in repr(exc):
                print("ERROR", kind, repr(exc)[:200], flush=True)
        finally:
            want = None
            del want
    print("fuzz: %d cases, %d bad, %.0f s" % (n, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
