"""Random cases of the reduce / resize kernels on host fibers (tests/emul: reduce_u8.hip with its matrix instruction,
resize_stream / resize_streamg / resize_tail, the general kernels -- the kernel files themselves under the mock HIP
runtime): vips_reduce by integer and fractional factors, vips_resize by one or two scales, 1-4 bands, uchar --
against the plain-C port, bit for bit.
  usage: LD_PRELOAD=tests/mock_hip/_build/libmockhip.so VIPS_HIP_LIBRARY=tests/emul/_build/libvipship_emul.so \
         python tools/fuzz_emul_resize.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import libvips_amd
from libvips_amd import Image
from tests import helpers
from tests.helpers import Port

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
libvips_amd.init(0)
lib = libvips_amd.lib
bad = 0
ran = {}
for case in range(n_cases):
    bands = int(rng.choice([1, 2, 3, 4, 4, 3]))
    w, h = int(rng.integers(40, 1400)), int(rng.integers(40, 900))
    src = helpers.lcg_image(w, h, bands, np.uint8, 2000 + case)
    im = Image.new_from_array(src)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    if rng.random() < 0.5:
        f = rng.choice([2.0, 4.0, 8.0, 8.0, float(rng.uniform(1.1, 9.0))])
        g = f if rng.random() < 0.7 else float(rng.choice([2.0, 4.0, 8.0, rng.uniform(1.1, 9.0)]))
        what = ("reduce", float(f), float(g))
        got = im.reduce(float(f), float(g), kernel="lanczos3").numpy()
        want = Port.reduce(src, float(f), float(g), "lanczos3")
    else:
        s = float(rng.choice([0.125, 0.25, 0.5, rng.uniform(0.02, 0.9)]))
        kw = {} if rng.random() < 0.6 else {"vscale": float(rng.uniform(0.03, 0.9))}
        what = ("resize", s, kw)
        got = im.resize(s, **kw).numpy()
        want = Port.resize(src, s, **kw)
    for k in libvips_amd.gate_report():
        ran[k] = ran.get(k, 0) + 1
    lib.vips_hip_gate_enable(0)
    if got.shape != want.shape or not np.array_equal(got, want):
        bad += 1
        print("MISMATCH", case, (w, h, bands), what, flush=True)
print("%d cases, %d mismatches; kernels: %s" % (n_cases, bad, dict(sorted(ran.items()))))
sys.exit(1 if bad else 0)
