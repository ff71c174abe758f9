#!/usr/bin/env python3
"""A few gaussblur(sigma 8) launches on a float image: the command rocprofv3 PMC passes wrap."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import libvips_amd  # noqa: E402
from bench import lcg_image_device  # noqa: E402
from libvips_amd import Image  # noqa: E402

n = int(os.environ.get("TUNE_SIZE", "8192"))
libvips_amd.init(0)
src = lcg_image_device(torch, n, n, 3, 12345, torch.device("cuda", 0)).float()
torch.cuda.synchronize()
im = Image.new_from_tensor(src, interpretation="srgb")
for _ in range(int(os.environ.get("TUNE_LAUNCHES", "3"))):
    out = im.gaussblur(8.0)
libvips_amd.synchronize()
