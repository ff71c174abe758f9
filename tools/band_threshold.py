#!/usr/bin/env python3
"""Where the three-launch band chain of vips_resize (ops_resample.cpp resize_down_u8_stream) overtakes the
one-kernel chain (resize_streamg.hip): square 3-band uchar images of growing size at a scale that leaves a
fractional reduce on both axes, each timed both ways (VIPS_HIP_RESIZE_BAND_MIN=0 / VIPS_HIP_NO_RESIZE_BAND=1).
usage: python tools/band_threshold.py  (on a GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import libvips_amd  # noqa: E402
from libvips_amd import Image  # noqa: E402


def main():
    libvips_amd.init(0)
    for edge in (768, 1024, 1448, 2048, 2896, 4096, 8192):
        t = torch.randint(0, 256, (edge, edge, 3), dtype=torch.uint8, device="cuda")
        im = Image.new_from_tensor(t, interpretation="srgb")
        row = []
        for scale in (0.23, 0.11):
            for env in ({"VIPS_HIP_RESIZE_BAND_MIN": "0"}, {"VIPS_HIP_NO_RESIZE_BAND": "1"}):
                for k in ("VIPS_HIP_RESIZE_BAND_MIN", "VIPS_HIP_NO_RESIZE_BAND"):
                    os.environ.pop(k, None)
                os.environ.update(env)
                for _ in range(5):
                    im.resize(scale)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 200
                for _ in range(n):
                    im.resize(scale)
                torch.cuda.synchronize()
                row.append((time.perf_counter() - t0) / n * 1e6)
        # 1 / 8: the one-kernel chain of resize_stream.hip (the default) against the band chain
        for env in ({}, {"VIPS_HIP_NO_RESIZE_STREAM": "1", "VIPS_HIP_RESIZE_BAND_MIN": "0"}):
            for k in ("VIPS_HIP_RESIZE_BAND_MIN", "VIPS_HIP_NO_RESIZE_BAND", "VIPS_HIP_NO_RESIZE_STREAM"):
                os.environ.pop(k, None)
            os.environ.update(env)
            for _ in range(5):
                im.resize(0.125)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 200
            for _ in range(n):
                im.resize(0.125)
            torch.cuda.synchronize()
            row.append((time.perf_counter() - t0) / n * 1e6)
        os.environ.pop("VIPS_HIP_NO_RESIZE_STREAM", None)
        print("%5d^2 x 3 (%5.1f MB): scale 0.23 band %6.1f us, one kernel %6.1f us; scale 0.11 band %6.1f us, one kernel %6.1f us; "
              "scale 1/8 one kernel %6.1f us, band %6.1f us"
              % (edge, edge * edge * 3 / 1e6, row[0], row[1], row[2], row[3], row[4], row[5]), flush=True)


if __name__ == "__main__":
    main()
