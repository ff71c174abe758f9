#!/usr/bin/env python3
"""precision=approximate against precision=integer on an 8192^2 uchar RGB image: the fast path
(approximated mask on the convi / fused separable kernels) and the generic box-sum kernels."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import libvips_amd  # noqa: E402
from bench import lcg_image_device  # noqa: E402
from libvips_amd import Image  # noqa: E402

n = int(os.environ.get("TUNE_SIZE", "8192"))
libvips_amd.init(0)
src = lcg_image_device(torch, n, n, 3, 12345, torch.device("cuda", 0))
torch.cuda.synchronize()
im = Image.new_from_tensor(src, interpretation="srgb")
g13 = np.rint(20 * np.exp(-(np.arange(-6, 7)[None, :] ** 2 + np.arange(-6, 7)[:, None] ** 2) / 18.0))


def clock(label, fn, reps=5):
    fn()
    libvips_amd.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    libvips_amd.synchronize()
    print("%-44s %8.3f ms" % (label, (time.perf_counter() - t0) / reps * 1e3), flush=True)


clock("gaussblur sigma 8 integer", lambda: im.gaussblur(8.0, precision="integer"))
clock("gaussblur sigma 8 approximate (fused)", lambda: im.gaussblur(8.0, precision="approximate"))
clock("conv g13 integer", lambda: im.conv(g13, g13.sum(), precision="integer"))
clock("conv g13 approximate (convi kernels)", lambda: im.conv(g13, g13.sum(), precision="approximate"))
os.environ["VIPS_HIP_NO_APPROX_FAST"] = "1"
clock("gaussblur sigma 8 approximate (generic)", lambda: im.gaussblur(8.0, precision="approximate"), 2)
clock("conv g13 approximate (generic)", lambda: im.conv(g13, g13.sum(), precision="approximate"), 1)
