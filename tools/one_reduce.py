#!/usr/bin/env python3
"""A handful of C2 launches and nothing else: the command rocprofv3 PMC passes wrap."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import libvips_amd  # noqa: E402
from bench import lcg_image_device  # noqa: E402
from libvips_amd import Image  # noqa: E402

n = int(os.environ.get("TUNE_SIZE", "16384"))
libvips_amd.init(0)
src = lcg_image_device(torch, n, n, 4, 12345, torch.device("cuda", 0))
torch.cuda.synchronize()
im = Image.new_from_tensor(src)
for _ in range(int(os.environ.get("TUNE_LAUNCHES", "4"))):
    im.reduce(8.0, 8.0, kernel="lanczos3")
libvips_amd.synchronize()
