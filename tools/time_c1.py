#!/usr/bin/env python3
"""BASELINE configs[0]'s device part (bench.run_c1) for variants of the streaming resize's launch
geometry.  usage: python tools/time_c1.py "NAME:VAR=val,..." ..."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

KNOBS = ("VIPS_HIP_STREAM_SEG", "VIPS_HIP_STREAM_BLOCKS", "VIPS_HIP_STREAM_BURST", "VIPS_HIP_STREAM_WINDOW",
         "VIPS_HIP_STREAM_HF")
args = argparse.Namespace(gpus=1, steps=20, warmup=5, no_settle=True)
ctx = bench.Ctx(args)
for rnd in range(2):
    for spec in sys.argv[1:] or ["default:"]:
        name, _, rest = spec.partition(":")
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(dict(kv.split("=") for kv in rest.split(",") if kv))
        e = bench.run_c1(ctx, 200, 20, verify=True, cpu=False)
        print("%-10s %.4f ms  parity %s  %s" % (name, e["ms"], e.get("parity", {}).get("bit_exact"), e.get("kernels")), flush=True)
