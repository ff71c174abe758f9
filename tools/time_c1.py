#!/usr/bin/env python3
"""C1's device part (vips_thumbnail_image(512) of a resident 4096^2 x 3 uchar image) under the
streaming resize's launch knobs.  usage: python tools/time_c1.py "NAME:VAR=val,..." ..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import libvips_amd  # noqa: E402
from libvips_amd import Image, lib  # noqa: E402

n = int(os.environ.get("C1_SIZE", "4096"))
libvips_amd.init(0)
dev = torch.device("cuda", 0)
src = bench.lcg_image_device(torch, n, n, 3, 12345, dev)
torch.cuda.synchronize()
im = Image.new_from_tensor(src, interpretation="srgb")
KNOBS = ("VIPS_HIP_STREAM_SEG", "VIPS_HIP_STREAM_BLOCKS", "VIPS_HIP_STREAM_DW", "VIPS_HIP_NO_RESIZE_STREAM")
for spec in sys.argv[1:] or ["default:"]:
    name, _, rest = spec.partition(":")
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(dict(kv.split("=") for kv in rest.split(",") if kv))
    for _ in range(5):
        im.thumbnail_image(512)
    libvips_amd.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        im.thumbnail_image(512)
    libvips_amd.synchronize()
    wall = (time.perf_counter() - t0) / 200 * 1e3
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    for _ in range(8):
        im.thumbnail_image(512)
    libvips_amd.synchronize()
    lib.vips_hip_gate_enable(0)
    rep = {k: round(v[1] / v[0], 4) for k, v in libvips_amd.gate_report().items()}
    lib.vips_hip_gate_reset()
    print("%-12s %.4f ms per thumbnail (200 back to back)  kernels alone %s" % (name, wall, rep), flush=True)
