#!/usr/bin/env python3
"""Stage breakdown of the fused resize tail on one BASELINE config 4 image ($VIPS_HIP_TAIL_DEBUG)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import libvips_amd  # noqa: E402
from libvips_amd import Image, lib  # noqa: E402

libvips_amd.init(0)
dev = torch.device("cuda", 0)
n = 8192
im = Image.new_from_tensor(bench.lcg_image_device(torch, n, n, 3, 12345, dev), interpretation="srgb")
for spec in sys.argv[1:] or ["0"]:
    name, _, rest = spec.partition(":")
    for k in ("VIPS_HIP_TAIL_DEBUG", "VIPS_HIP_TAIL_TH", "VIPS_HIP_TAIL_TW", "VIPS_HIP_TAIL_LDS"):
        os.environ.pop(k, None)
    os.environ.update(dict(kv.split("=") for kv in rest.split(",") if kv))
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    for _ in range(6):
        im.resize(0.125)
    libvips_amd.synchronize()
    lib.vips_hip_gate_enable(0)
    rep = {k: round(v[1] / v[0], 4) for k, v in libvips_amd.gate_report().items()}
    print("%-18s %s" % (name, rep), flush=True)
