#!/usr/bin/env python3
"""Where and how often does the default float mode of the streaming separable convolution differ
from the exact mode?  usage: python tools/c3_fast_diff.py [size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import libvips_amd  # noqa: E402
from libvips_amd import Image, lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
libvips_amd.init(0)
dev = torch.device("cuda", 0)
src = bench.lcg_image_device(torch, n, n, 3, 12345, dev).float()
torch.cuda.synchronize()
im = Image.new_from_tensor(src, interpretation="srgb")
for what, fn in (("blur", lambda: im.gaussblur(8.0)), ("blur+lab", lambda: im.gaussblur_colourspace(8.0, "lab")),
                 ("blur float", lambda: im.gaussblur(8.0, precision="float"))):
    lib.vips_hip_set_exact_float(1)
    a = fn()
    lib.vips_hip_set_exact_float(0)
    b = fn()
    ta = torch.empty((n, n, 3), dtype=torch.float32, device=dev)
    tb = torch.empty_like(ta)
    lib.vips_hip_memcpy_d2d(ta.data_ptr(), lib.vips_hip_image_get_data(a._h), ta.numel() * 4)
    lib.vips_hip_memcpy_d2d(tb.data_ptr(), lib.vips_hip_image_get_data(b._h), tb.numel() * 4)
    libvips_amd.synchronize()
    ne = ta != tb
    cnt = int(ne.sum().item())
    print("%-10s %d of %d elements differ (%.3g)" % (what, cnt, ta.numel(), cnt / ta.numel()))
    if cnt:
        idx = ne.nonzero()[:8]
        for y, x, c in idx.tolist():
            print("   (%d, %d, %d): exact %r fast %r" % (y, x, c, float(ta[y, x, c]), float(tb[y, x, c])))
        d = (ta - tb).abs()
        print("   max abs diff %g" % float(d.max().item()))
