#!/usr/bin/env python3
"""Is the C2 kernel power-limited?  The same launches on LCG bytes, on zeros and on a constant
image (MI355X clocks to its power budget: MI355X_MICROARCH.md, DVFS give-back)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import libvips_amd  # noqa: E402
from bench import lcg_image_device  # noqa: E402
from libvips_amd import Image, lib  # noqa: E402

dev = torch.device("cuda", 0)
libvips_amd.init(0)
stream = torch.cuda.Stream(device=dev)
lib.vips_hip_set_stream(stream.cuda_stream)
n = 16384
with torch.cuda.stream(stream):
    data = {
        "lcg": lcg_image_device(torch, n, n, 4, 12345, dev),
        "zeros": torch.zeros((n, n, 4), dtype=torch.uint8, device=dev),
        "const200": torch.full((n, n, 4), 200, dtype=torch.uint8, device=dev),
    }
torch.cuda.synchronize()
ims = {k: Image.new_from_tensor(v) for k, v in data.items()}
knobs = [("full", {}), ("loads_only", {"VIPS_HIP_FUSED_DEBUG": "16"}), ("arith_only", {"VIPS_HIP_FUSED_DEBUG": "8"})]
times = {}
with torch.cuda.stream(stream):
    for rnd in range(5):
        for kname, env in knobs:
            os.environ.pop("VIPS_HIP_FUSED_DEBUG", None)
            os.environ.update(env)
            for name, im in ims.items():
                lib.vips_hip_gate_reset()
                lib.vips_hip_gate_enable(1)
                for _ in range(15):
                    im.reduce(8.0, 8.0, kernel="lanczos3")
                torch.cuda.synchronize()
                lib.vips_hip_gate_enable(0)
                (k, (launches, total)), = libvips_amd.gate_report().items()
                if rnd:
                    times.setdefault((kname, name), []).append(total / launches)
for key, t in times.items():
    print("%-11s %-9s med %.4f ms" % (key[0], key[1], statistics.median(t)))
