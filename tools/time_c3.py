#!/usr/bin/env python3
"""C3 pieces on the MI355X: blur only / blur + colour, old and new kernels, block sizes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import libvips_amd  # noqa: E402
from libvips_amd import Image, lib  # noqa: E402

n = int(os.environ.get("C3_SIZE", "16384"))
libvips_amd.init(0)
dev = torch.device("cuda", 0)
src = bench.lcg_image_device(torch, n, n, 3, 12345, dev).float()
torch.cuda.synchronize()
im = Image.new_from_tensor(src, interpretation="srgb")
KNOBS = ("VIPS_HIP_STREAM_NT", "VIPS_HIP_NO_STREAM_CONVSEP")


def run(name, env, fn):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    fn()
    libvips_amd.synchronize()
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    for _ in range(3):
        fn()
    libvips_amd.synchronize()
    lib.vips_hip_gate_enable(0)
    rep = libvips_amd.gate_report()
    print("%-34s %s" % (name, {k: round(v[1] / v[0], 3) for k, v in rep.items()}), flush=True)


scale = (32768.0 / n) ** 2
print("image %d^2 (x%.0f for 32768^2)" % (n, scale))
for exact in (1, 0):
    lib.vips_hip_set_exact_float(exact)
    tag = "exact" if exact else "default"
    run("blur stream 768 [%s]" % tag, {}, lambda: im.gaussblur(8.0))
    run("blur+lab stream 768 [%s]" % tag, {}, lambda: im.gaussblur_colourspace(8.0, "lab"))
    run("blur float precision [%s]" % tag, {}, lambda: im.gaussblur(8.0, precision="float"))
lib.vips_hip_set_exact_float(1)
run("blur old kernel", {"VIPS_HIP_NO_STREAM_CONVSEP": "1"}, lambda: im.gaussblur(8.0))
run("blur stream 768", {}, lambda: im.gaussblur(8.0))
run("blur+lab stream 768", {}, lambda: im.gaussblur_colourspace(8.0, "lab"))
run("blur+xyz stream 768", {}, lambda: im.gaussblur_colourspace(8.0, "xyz"))
run("blur sigma 2 stream 768", {}, lambda: im.gaussblur(2.0))
run("blur sigma 2 old", {"VIPS_HIP_NO_STREAM_CONVSEP": "1"}, lambda: im.gaussblur(2.0))
run("colourspace only", {}, lambda: im.colourspace("lab"))
