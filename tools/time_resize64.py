#!/usr/bin/env python3
"""The resize chain of C4 alone: one 64-image launch of resize_stream_u8 on 8192 x 8192 x 3 images,
kernel time by HIP events (the library's gates), variants by environment, interleaved and repeated.
usage: python tools/time_resize64.py "NAME:VAR=val,..." ..."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import libvips_amd  # noqa: E402
from libvips_amd import Image, lib  # noqa: E402

KNOBS = ("VIPS_HIP_STREAM_HF", "VIPS_HIP_STREAM_DEBUG", "VIPS_HIP_STREAM_BLOCKS",
         "VIPS_HIP_STREAM_SEG", "VIPS_HIP_STREAM_DW", "VIPS_HIP_STREAM_LDSPAD", "VIPS_HIP_STREAM_BURST", "VIPS_HIP_STREAM_WINDOW")
n, count = 8192, 64
libvips_amd.init(0)
dev = torch.device("cuda", 0)
store = torch.empty((count, n, n, 3), dtype=torch.uint8, device=dev)
for k in range(count):
    bench.lcg_image_device(torch, n, n, 3, 12345 + k, dev, out=store[k])
torch.cuda.synchronize()
ims = [Image.new_from_tensor(store[k], interpretation="srgb") for k in range(count)]
specs = sys.argv[1:] or ["default:"]
ref = None
times = {s.partition(":")[0]: [] for s in specs}
same = {}
for rnd in range(int(os.environ.get("ROUNDS", "6"))):
    for spec in specs:
        name, _, rest = spec.partition(":")
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(dict(kv.split("=") for kv in rest.split(",") if kv))
        outs = libvips_amd.resize_sharpen_batch(ims, 0.125, sharpen=False)
        libvips_amd.synchronize()
        if rnd == 0:
            one = outs[5].numpy()
            if ref is None:
                ref = one
            same[name] = bool((one == ref).all())
        lib.vips_hip_gate_reset()
        lib.vips_hip_gate_enable(1)
        for _ in range(5):
            outs = libvips_amd.resize_sharpen_batch(ims, 0.125, sharpen=False)
        libvips_amd.synchronize()
        lib.vips_hip_gate_enable(0)
        rep = libvips_amd.gate_report()
        lib.vips_hip_gate_reset()
        g = rep["resize_stream_u8"]
        times[name].append(g[1] / g[0] / count)
for name, v in times.items():
    print("%-14s same=%s  per image: min %.4f  med %.4f  max %.4f ms  (%.0f GB/s of the images at the median)" %
          (name, same[name], min(v), statistics.median(v), max(v), n * n * 3 / statistics.median(v) / 1e6), flush=True)
