#!/usr/bin/env python3
"""vips_reducev(8) / vips_reduceh(8) on an 8192 x 8192 x 3 uchar image: ms a call for each environment given as
NAME=VALUE[,NAME=VALUE] arguments ("-" = the default), outputs compared with the first one's."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import libvips_amd  # noqa: E402
from bench import lcg_image_device  # noqa: E402
from libvips_amd import Image  # noqa: E402

n = int(os.environ.get("TUNE_SIZE", "8192"))
bands = int(os.environ.get("TUNE_BANDS", "3"))
axis = os.environ.get("TUNE_AXIS", "v")
libvips_amd.init(0)
src = lcg_image_device(torch, n, n, bands, 12345, torch.device("cuda", 0))
torch.cuda.synchronize()
im = Image.new_from_tensor(src)
call = (lambda: im.reducev(8.0, kernel="lanczos3")) if axis == "v" else (lambda: im.reduceh(8.0, kernel="lanczos3"))
first = None
for spec in sys.argv[1:] or ["-"]:
    names = []
    if spec != "-":
        for kv in spec.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
            names.append(k)
    libvips_amd.lib.vips_hip_gate_reset()
    libvips_amd.lib.vips_hip_gate_enable(1)
    out = call()
    gates = sorted(libvips_amd.gate_report())
    libvips_amd.lib.vips_hip_gate_enable(0)
    got = out.numpy()
    if first is None:
        first = got
    same = bool((got == first).all())
    for _ in range(10):
        call()
    libvips_amd.synchronize()
    best = []
    for rep in range(3):
        libvips_amd.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            call()
        libvips_amd.synchronize()
        best.append((time.perf_counter() - t0) / 50 * 1e3)
    nbytes = n * n * bands + n * (n // 8) * bands
    ms = min(best)
    print(f"{spec:40s} {ms:.4f} ms  {nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s  same={same}  {gates}", flush=True)
    for k in names:
        del os.environ[k]
