#!/usr/bin/env python3
"""Does the 64 KB row pitch of the C2 image (16384 RGBA pixels: a power of two) cost the fused
reduce anything?  The same image with rows padded by PAD bytes, through the region ABI
(VipsHipRegion carries the stride).  usage: python tools/tune_c2_pitch.py [pad ...]"""
import ctypes
import math
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import libvips_amd  # noqa: E402
from bench import lcg_image_device  # noqa: E402
from libvips_amd import KERNELS, lib  # noqa: E402
from libvips_amd._ffi import Region, check, check_handle  # noqa: E402

pads = [int(x) for x in sys.argv[1:]] or [0, 128, 256, 512, 1024, 4096]
n = 16384
dev = torch.device("cuda", 0)
libvips_amd.init(0)
stream = torch.cuda.Stream(device=dev)
lib.vips_hip_set_stream(stream.cuda_stream)
with torch.cuda.stream(stream):
    src = lcg_image_device(torch, n, n, 4, 12345, dev)
torch.cuda.synchronize()
rv = check_handle(lib.vips_hip_reduce_new(KERNELS["lanczos3"], 8.0, n, n // 8, math.nan))
rh = check_handle(lib.vips_hip_reduce_new(KERNELS["lanczos3"], 8.0, n, n // 8, math.nan))
out = torch.empty((n // 8, n // 8, 4), dtype=torch.uint8, device=dev)
ref = None
times = {p: [] for p in pads}
bufs = {}
for p in pads:
    pitch = n * 4 + p
    buf = torch.zeros((n, pitch), dtype=torch.uint8, device=dev)
    buf[:, :n * 4] = src.reshape(n, n * 4)
    bufs[p] = buf
torch.cuda.synchronize()
with torch.cuda.stream(stream):
    for rnd in range(6):
        for p in pads:
            pitch = n * 4 + p
            rin = Region(bufs[p].data_ptr(), 0, 0, n, n, n, n, 4, 0, pitch)
            rout = Region(out.data_ptr(), 0, 0, n // 8, n // 8, n // 8, n // 8, 4, 0, n // 8 * 4)
            lib.vips_hip_gate_reset()
            lib.vips_hip_gate_enable(1)
            for _ in range(15):
                check(lib.vips_hip_reduce_gen(rv, rh, ctypes.byref(rin), ctypes.byref(rout)))
            torch.cuda.synchronize()
            lib.vips_hip_gate_enable(0)
            rep = libvips_amd.gate_report()
            (name, (launches, total)), = rep.items()
            if rnd:
                times[p].append(total / launches)
            got = out.clone()
            if ref is None:
                ref = got
            assert torch.equal(got, ref), "padded pitch changed the pixels"
for p in pads:
    t = times[p]
    print("pitch 65536 + %-5d  min %.4f  med %.4f  max %.4f ms" % (p, min(t), statistics.median(t), max(t)))
