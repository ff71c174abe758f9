#!/usr/bin/env python3
"""A few vips_sharpen launches on BASELINE config 4's thumbnails (1024 x 1024 x 3 sRGB, the 1/8 resize of LCG noise)
or on an image of TUNE_SIZE^2 of noise (TUNE_KIND=noise): the command rocprofv3 passes wrap; prints the gate times."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import libvips_amd  # noqa: E402
from bench import lcg_image_device  # noqa: E402
from libvips_amd import Image, lib  # noqa: E402

libvips_amd.init(0)
dev = torch.device("cuda", 0)
count = int(os.environ.get("TUNE_IMAGES", "16"))
if os.environ.get("TUNE_KIND", "thumb") == "noise":
    n = int(os.environ.get("TUNE_SIZE", "8192"))
    ims = [Image.new_from_tensor(lcg_image_device(torch, n, n, 3, 12345, dev), interpretation="srgb")]
else:
    ims = []
    for k in range(count):
        big = Image.new_from_tensor(lcg_image_device(torch, 8192, 8192, 3, 12345 + k, dev), interpretation="srgb")
        ims.append(big.resize(0.125))
        del big
libvips_amd.synchronize()
outs = [im.sharpen() for im in ims]  # (tables, the identity proof, the pool)
libvips_amd.synchronize()
lib.vips_hip_gate_reset()
lib.vips_hip_gate_enable(1)
for _ in range(int(os.environ.get("TUNE_LAUNCHES", "4"))):
    outs = [im.sharpen() for im in ims]
libvips_amd.synchronize()
lib.vips_hip_gate_enable(0)
print({k: (v[0], round(v[1] / v[0] * 1e3, 2), "us") for k, v in libvips_amd.gate_report().items()}, flush=True)
