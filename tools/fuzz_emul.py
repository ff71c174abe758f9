#!/usr/bin/env python3
"""CPU fuzz of round 4's streaming kernels on host fibers (tests/emul) against the compiled reference:
random sizes, bands and parameters for shrinkh / reduceh / reducev on uchar, conv on ushort and uchar,
gaussblur on uchar.  Whatever kernel the dispatcher picks must give the reference's pixels when it is an
emulated one; cases that land on a kernel the mock runtime cannot run (no pixels) are counted and skipped.

usage:  LD_PRELOAD=tests/mock_hip/_build/libmockhip.so VIPS_HIP_LIBRARY=tests/emul/_build/libvipship_emul.so \
        python tools/fuzz_emul.py [cases] [seed] [kind,kind,...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import libvips_amd  # noqa: E402
from libvips_amd import Image  # noqa: E402
from tests import helpers  # noqa: E402

EMULATED = {"shrinkh_u8_stream", "shrinkv_reducev_u8_band", "reduceh_u8_packed", "reducev_u8_stream", "reducev_u8_band", "reduceh_u8_band", "reducev_u16_band", "reduceh_u16_band", "conv_u16_2d", "conv_u8_2d", "conv_u8_sep", "conv_u8_mfma_sep", "conv_u8_mfma_2d", "conv_u16_mfma_sep",
            "shrinkv_u16_stream", "shrinkh_u16_stream", "reducev_u16_stream", "reduceh_u16_lds"}
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
# (round 5: ushort gaussblur / convsep on the matrix cores, and vips_resize through the band chain)
KINDS = sys.argv[3].split(",") if len(sys.argv) > 3 else ["shrinkh", "reduceh", "reducev", "conv16", "conv8", "blur8", "blur16", "sep16", "resize"]
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MAX_H = int(os.environ.get("FUZZ_MAX_H", "90"))
MAX_W = int(os.environ.get("FUZZ_MAX_W", "640"))
libvips_amd.init(0)
lib = libvips_amd.lib
ran = {}
skipped = 0
for case in range(n_cases):
    kind = rng.choice(KINDS)
    bands = int(rng.choice([1, 2, 3, 4]))
    # widths that make rows of whole dwords most of the time
    # ($FUZZ_MAX_H / $FUZZ_MAX_W: taller / wider images -- several rows of tiles, the bottom-up walk of every other
    # block of rows, more than one 16-row half of an output tile)
    w = int(rng.integers(1, MAX_W // 4)) * 4 if rng.random() < 0.8 else int(rng.integers(2, MAX_W + 60))
    h = int(rng.integers(1, MAX_H))
    dt = np.uint16 if kind in ("conv16", "blur16", "sep16") else np.uint8
    src = helpers.lcg_image(w, h, bands, dt, 1000 + case)
    if rng.random() < 0.3:
        src[: h // 2] = np.iinfo(dt).max
    im = Image.new_from_array(src)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    try:
        if kind == "shrinkh":
            hs = int(rng.choice([2, 3, 4, 5, 6, 7, 8, 12, 16]))
            ceil = bool(rng.random() < 0.3)
            if w < hs:
                continue
            got = im.shrinkh(hs, ceil=ceil).numpy()
            chain = "shrinkh:hshrink=%d%s" % (hs, ",ceil=true" if ceil else "")
            want = lambda: helpers.Ref.run_chain(chain, src)
        elif kind == "reduceh":
            s = float(rng.choice([4.0, 8.0]))
            k = str(rng.choice(["lanczos3", "cubic", "linear", "mitchell", "lanczos2"]))
            if w < 2 * s:
                continue
            got = im.reduceh(s, kernel=k).numpy()
            chain = "reduceh:hshrink=%r,kernel=%s" % (s, k)
            want = lambda: helpers.Ref.run_chain(chain, src)
        elif kind == "reducev":
            s = float(rng.choice([1.7, 2.5, 3.3, 5.1, 7.3, 9.9]))
            k = str(rng.choice(["lanczos3", "cubic", "linear"]))
            hh = int(rng.integers(int(3 * s) + 2, 400))
            src = helpers.lcg_image(w, hh, bands, np.uint8, 2000 + case)
            im = Image.new_from_array(src)
            got = im.reducev(s, kernel=k).numpy()
            chain = "reducev:vshrink=%r,kernel=%s" % (s, k)
            want = lambda: helpers.Ref.run_chain(chain, src)
        elif kind in ("conv16", "conv8"):
            if bands == 2:
                bands = 3
                src = helpers.lcg_image(w, h, bands, dt, 3000 + case)
                im = Image.new_from_array(src)
            mw, mh = int(rng.choice([1, 3, 5])), int(rng.choice([1, 3, 5]))
            if mw == 1 and mh == 1:
                mw = 3
            mask = rng.integers(-6, 20, size=(mh, mw)).astype(np.float64)
            scale = int(max(1, abs(mask.sum()))) if rng.random() < 0.7 else int(rng.integers(1, 300))
            got = im.conv(mask, scale=scale, precision="integer").numpy()
            want = lambda: helpers.Ref.run_mask("conv", src, mask, scale, 0.0, "precision=integer")
        elif kind == "sep16":
            n = int(rng.choice([3, 5, 7, 9, 13, 21, 33]))
            mask = rng.integers(-3, 30, size=n).astype(np.float64)
            scale = int(max(1, abs(mask.sum()))) if rng.random() < 0.7 else int(rng.integers(1, 500))
            got = im.convsep(list(mask), scale=scale, precision="integer").numpy()
            want = lambda: helpers.Ref.run_mask("convsep", src, mask[None, :], scale, 0.0, "precision=integer")
        elif kind == "resize":
            hh = int(rng.integers(40, 900))
            src = helpers.lcg_image(w, hh, bands, np.uint8, 4000 + case)
            if rng.random() < 0.3:
                src[: hh // 2] = 255
            im = Image.new_from_array(src)
            hscale = float(rng.choice([0.45, 0.3, 0.23, 0.17, 0.11, 0.07]))
            vscale = 1.0 / float(rng.uniform(2.1, 34.0))
            if w * hscale < 2 or hh * vscale < 2:
                continue
            got = im.resize(hscale, vscale=vscale).numpy()
            chain = "resize:scale=%r,vscale=%r" % (hscale, vscale)
            want = lambda: helpers.Ref.run_chain(chain, src)
        else:
            if bands == 2 and kind == "blur8":
                continue
            sigma = float(rng.choice([1.0, 2.0, 3.0, 5.0, 8.0]))
            got = im.gaussblur(sigma).numpy()
            want = lambda: helpers.Ref.run_chain("gaussblur:sigma=%r" % sigma, src)
    finally:
        report = list(libvips_amd.gate_report())
        lib.vips_hip_gate_enable(0)
    if not report or any(g not in EMULATED for g in report):
        skipped += 1
        continue
    ref = want()
    ok = got.shape == ref.shape and got.dtype == ref.dtype and np.array_equal(got, ref)
    for g in report:
        ran[g] = ran.get(g, 0) + 1
    if not ok:
        bad = np.argwhere(got != ref) if got.shape == ref.shape else []
        print("MISMATCH", kind, (w, h, bands), report, len(bad), bad[:4] if len(bad) else (got.shape, ref.shape))
        sys.exit(1)
print("fuzz ok: %d cases on emulated kernels %r, %d on kernels the mock cannot run" % (sum(ran.values()), ran, skipped))
