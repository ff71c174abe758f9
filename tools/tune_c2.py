#!/usr/bin/env python3
"""Interleaved timing of the C2 kernel variants (profiling aid; the env knobs are debug-only).

usage: python tools/tune_c2.py "NAME:VAR=val,VAR=val" ...   (NAME: alone = defaults)
Each variant is timed ROUNDS times (interleaved, so box-to-box and clock drift cancel) with the
library's own HIP-event gates; prints min / median kernel ms per variant.
"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import libvips_amd  # noqa: E402
from bench import lcg_image_device  # noqa: E402
from libvips_amd import Image, lib  # noqa: E402

KNOBS = ("VIPS_HIP_FUSED_DEBUG", "VIPS_HIP_FUSED_CAP", "VIPS_HIP_NO_MFMA", "VIPS_HIP_FUSED_ALIGN",
         "VIPS_HIP_FUSED_OWT", "VIPS_HIP_FUSED_STAGGER", "VIPS_HIP_FUSED_BURST", "VIPS_HIP_FUSED_EXCH",
         "VIPS_HIP_FUSED_PLAIN")
ROUNDS = int(os.environ.get("TUNE_ROUNDS", "5"))
LAUNCHES = int(os.environ.get("TUNE_LAUNCHES", "15"))


def main():
    variants = []
    for spec in sys.argv[1:] or ["default:"]:
        name, _, rest = spec.partition(":")
        env = dict(kv.split("=") for kv in rest.split(",") if kv)
        variants.append((name, env))
    device = torch.device("cuda", 0)
    libvips_amd.init(0)
    stream = torch.cuda.Stream(device=device)
    lib.vips_hip_set_stream(stream.cuda_stream)
    n = int(os.environ.get("TUNE_SIZE", "16384"))
    with torch.cuda.stream(stream):
        src = lcg_image_device(torch, n, n, 4, 12345, device)
    torch.cuda.synchronize()
    im = Image.new_from_tensor(src)
    times = {name: [] for name, _ in variants}
    walls = {name: [] for name, _ in variants}
    kernels = {}
    with torch.cuda.stream(stream):
        for rnd in range(ROUNDS + 1):
            for name, env in variants:
                for k in KNOBS:
                    os.environ.pop(k, None)
                os.environ.update(env)
                # back-to-back launches, wall clock (what bench.py's ms_per_step sees) ...
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(LAUNCHES):
                    im.reduce(8.0, 8.0, kernel="lanczos3")
                torch.cuda.synchronize()
                wall_ms = (time.perf_counter() - t0) * 1e3 / LAUNCHES
                if rnd:
                    walls[name].append(wall_ms)
                # ... and every launch between its own pair of HIP events (the gates)
                lib.vips_hip_gate_reset()
                lib.vips_hip_gate_enable(1)
                for _ in range(LAUNCHES):
                    im.reduce(8.0, 8.0, kernel="lanczos3")
                torch.cuda.synchronize()
                lib.vips_hip_gate_enable(0)
                rep = libvips_amd.gate_report()
                kname, (launches, total_ms) = max(rep.items(), key=lambda kv: kv[1][1])
                kernels[name] = kname
                if rnd:  # round 0 = warm-up
                    times[name].append(total_ms / launches)
    for name, _ in variants:
        t = times[name]
        print("%-24s gate min %.4f  med %.4f  max %.4f ms | back-to-back wall med %.4f ms  %s"
              % (name, min(t), statistics.median(t), max(t), statistics.median(walls[name]), kernels[name]))


if __name__ == "__main__":
    main()
