#!/usr/bin/env python3
"""C4 per-image kernels and the batched rate for variants of the vertical half of resize.
usage: python tools/time_c4.py "NAME:VAR=val,..." ..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import libvips_amd  # noqa: E402
from libvips_amd import Image, lib  # noqa: E402

KNOBS = ("VIPS_HIP_NO_RESIZE_TAIL", "VIPS_HIP_NO_FUSED_SHARPEN", "VIPS_HIP_TAIL_TH", "VIPS_HIP_TAIL_TW",
         "VIPS_HIP_NO_RESIZE_STREAM", "VIPS_HIP_NO_BATCH_LAUNCH", "VIPS_HIP_STREAM_BLOCKS", "VIPS_HIP_STREAM_SEG",
         "VIPS_HIP_STREAM_DEBUG", "VIPS_HIP_BATCH_OVERLAP", "VIPS_HIP_STREAM_DW", "VIPS_HIP_STREAM_HF")
n, count = 8192, int(os.environ.get("C4_IMAGES", "64"))
scale = float(os.environ.get("C4_SCALE", "0.125"))
libvips_amd.init(0)
dev = torch.device("cuda", 0)
store = torch.empty((count, n, n, 3), dtype=torch.uint8, device=dev)
for k in range(count):
    bench.lcg_image_device(torch, n, n, 3, 12345 + k, dev, out=store[k])
torch.cuda.synchronize()
ims = [Image.new_from_tensor(store[k], interpretation="srgb") for k in range(count)]
ref = None
for spec in sys.argv[1:] or ["default:"]:
    name, _, rest = spec.partition(":")
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(dict(kv.split("=") for kv in rest.split(",") if kv))
    one = ims[0].resize(scale).sharpen().numpy()
    if ref is None:
        ref = one
    same = bool((one == ref).all())
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    for _ in range(4):
        ims[0].resize(scale)
    libvips_amd.synchronize()
    lib.vips_hip_gate_enable(0)
    rep = {k: round(v[1] / v[0], 4) for k, v in libvips_amd.gate_report().items()}
    lib.vips_hip_gate_reset()
    libvips_amd.resize_sharpen_batch(ims, scale, threads=8)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        libvips_amd.resize_sharpen_batch(ims, scale, threads=8)
        best = min(best, (time.perf_counter() - t0) / count * 1e3)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    libvips_amd.resize_sharpen_batch(ims, scale, threads=8)
    lib.vips_hip_gate_enable(0)
    inb = {k: round(v[1] / count, 4) for k, v in libvips_amd.gate_report().items()}
    lib.vips_hip_gate_reset()
    print("%-14s same=%s batch %.4f ms/image  resize kernels alone %s  in the batch, per image %s" %
          (name, same, best, rep, inb), flush=True)
