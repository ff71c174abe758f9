#!/usr/bin/env python
"""Timings of the other BASELINE configs (C3, C4 per image, C5 kernel) on one MI355X.

These are parity-test configurations, not bench.py lines (the BASELINE metric is quoted on
config 2); this script records where their kernels stand against the HBM roofline so the
next kernel to tune is picked from numbers.  usage: python tools/bench_configs.py [--quick]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="", help="comma list of c3,c4,c5")
    args = ap.parse_args()

    import numpy as np
    import torch

    import bench
    import libvips_amd
    from libvips_amd import Image, lib

    libvips_amd.init(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    lib.vips_hip_set_stream(stream.cuda_stream)
    results = []

    def timed(name, fn, alg_bytes, reps=3):
        with torch.cuda.stream(stream):
            fn()
            torch.cuda.synchronize()
            lib.vips_hip_gate_reset()
            lib.vips_hip_gate_enable(1)
            t0 = time.perf_counter()
            for _ in range(reps):
                out = fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            lib.vips_hip_gate_enable(0)
            gates = libvips_amd.gate_report()
            lib.vips_hip_gate_reset()
        rec = {
            "config": name,
            "ms": round(dt * 1e3, 3),
            "algorithmic_GBps": round(alg_bytes / dt / 1e9, 1),
            "frac_of_8TBps": round(alg_bytes / dt / 8e12, 4),
            "kernels_ms": {k: round(v[1] / v[0], 3) for k, v in gates.items()},
        }
        results.append(rec)
        print(json.dumps(rec), flush=True)
        return out

    only = set(x for x in args.only.lower().split(",") if x)
    with torch.cuda.stream(stream):
      if not only or "c3" in only:
        # ---- C3: gaussblur(sigma 8) -> colourspace(LAB) on float sRGB
        n = 8192 if args.quick else 32768
        src = bench.lcg_image_device(torch, n, n, 3, 12345, dev).float()
        torch.cuda.synchronize()
        im = Image.new_from_tensor(src, interpretation="srgb")
        timed("C3 gaussblur(sigma=8)+sRGB->Lab %dx%dx3 f32" % (n, n),
              lambda: im.gaussblur(8.0).colourspace("lab"), 2 * n * n * 12, reps=2)
        timed("C3a gaussblur(sigma=8) only", lambda: im.gaussblur(8.0), 2 * n * n * 12, reps=2)
        timed("C3b colourspace sRGB->Lab only", lambda: im.colourspace("lab"), 2 * n * n * 12, reps=2)
        del im, src
        torch.cuda.empty_cache()
        lib.vips_hip_pool_trim()

      if not only or "c4" in only:
        # ---- C4 per image: resize(1/8) -> sharpen on 8192^2 x3 u8
        n = 8192
        src = bench.lcg_image_device(torch, n, n, 3, 12345, dev)
        torch.cuda.synchronize()
        im = Image.new_from_tensor(src, interpretation="srgb")
        timed("C4 resize(1/8)+sharpen %dx%dx3 u8 (per image)" % (n, n),
              lambda: im.resize(0.125).sharpen(), n * n * 3 + (n // 8) ** 2 * 3, reps=5)
        timed("C4a resize(1/8) only", lambda: im.resize(0.125), n * n * 3 + (n // 8) ** 2 * 3, reps=5)
        del im, src

      if not only or "c5" in only:
        # ---- C5 kernel: 31x31 float mask on ushort (one GPU's share is 65536 x 8192)
        w, h = (4096, 1024) if args.quick else (16384, 2048)
        src = bench.lcg_image_device(torch, w, h, 2, 12345, dev).view(torch.uint16).reshape(h, w, 1)
        torch.cuda.synchronize()
        im = Image.new_from_tensor(src.contiguous())
        mask, scale = libvips_amd.gaussmat(5, 0.01, False, "float")
        timed("C5 conv 31x31 float on %dx%d u16" % (w, h),
              lambda: im.conv(mask, scale=scale, precision="float"), w * h * 6, reps=1)
    out_path = os.path.join(ROOT, "gpurun_out", "bench_configs.json")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(results, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
