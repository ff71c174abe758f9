"""Random cases of convsep_stream.hip on host fibers (tests/emul: the kernel file itself under the mock HIP runtime):
sizes around the strip and segment boundaries, 1-4 bands, sigmas from 3 to 29 taps, integer / almost-integer /
float pixels, with and without the colour epilogue -- against the plain-C port, bit for bit.
  usage: LD_PRELOAD=tests/mock_hip/_build/libmockhip.so VIPS_HIP_LIBRARY=tests/emul/_build/libvipship_emul.so \
         python tools/fuzz_emul_convsep.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import libvips_amd
from libvips_amd import Image
from tests import helpers
from tests.helpers import PortCC

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
libvips_amd.init(0)
libvips_amd.lib.vips_hip_set_exact_float(1)
bad = 0
for case in range(n_cases):
    colour = rng.random() < 0.5
    bands = 3 if colour else int(rng.integers(1, 5))
    # widths around multiples of the strip (256 pixels of 3 bands, 768 / bands elements), heights around the steps of 8
    w = int(rng.choice([rng.integers(1, 40), 256 + rng.integers(-9, 10), 512 + rng.integers(-9, 10), rng.integers(40, 1100)]))
    h = int(rng.choice([rng.integers(1, 12), rng.integers(12, 80), 8 * rng.integers(2, 40) + rng.integers(-1, 2), rng.integers(80, 500)]))
    w, h = max(w, 1), max(h, 1)
    sigma = float(rng.choice([0.6, 1.0, 2.0, 3.1, 5.0, 8.0]))
    kind = rng.choice(["integer", "almost", "float"])
    precision = "integer" if kind != "float" or rng.random() < 0.5 else "float"
    if kind == "float":
        src = helpers.lcg_image(w, h, bands, np.float32, 1000 + case)
    else:
        src = helpers.lcg_image(w, h, bands, np.uint8, 1000 + case).astype(np.float32)
        if kind == "almost":
            for _ in range(int(rng.integers(1, 4))):
                src[rng.integers(0, h), rng.integers(0, w), rng.integers(0, bands)] = rng.choice([0.5, 255.5, -3.0, 300.0, 17.25])
    env = {}
    if rng.random() < 0.3:
        env["VIPS_HIP_STREAM_EPI"] = "0"
    if rng.random() < 0.2:
        env["VIPS_HIP_STREAM_INT"] = "0"
    os.environ.update(env)
    try:
        if colour:
            got = Image.new_from_array(src, interpretation="srgb").gaussblur_colourspace(sigma, "lab", precision=precision).numpy()
            want = PortCC.colourspace(PortCC.gaussblur(src, sigma, precision=precision), "lab", "srgb")
        else:
            got = Image.new_from_array(src).gaussblur(sigma, precision=precision).numpy()
            want = PortCC.gaussblur(src, sigma, precision=precision)
    finally:
        for k in env:
            del os.environ[k]
    ok = got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if not ok:
        bad += 1
        print("MISMATCH", case, (w, h, bands), sigma, kind, precision, colour, env, flush=True)
print("%d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
