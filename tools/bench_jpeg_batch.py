#!/usr/bin/env python3
"""Thumbnails per second from JPEG files (the shape of BASELINE config C4 when the inputs are
files): N synthetic 4000x3000 JPEGs -> 256x256, shrink-on-load on host threads, everything
after the decode on the device.  usage: python tools/bench_jpeg_batch.py [n_files] [threads...]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from PIL import Image as PILImage  # noqa: E402

import libvips_amd  # noqa: E402
from libvips_amd import Image  # noqa: E402
from tests import helpers  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    threads = [int(v) for v in sys.argv[2:]] or [1, 8, 32, 64]
    libvips_amd.init(0)
    with tempfile.TemporaryDirectory() as d:
        y, x = np.mgrid[0:3000, 0:4000]
        base = np.stack([(np.sin(x / 37.0) + 1) * 127, (np.cos(y / 23.0) + 1) * 127, (x + y) % 256], axis=2)
        base = (base.astype(int) + helpers.lcg_image(4000, 3000, 3) // 8).clip(0, 255).astype(np.uint8)
        one = os.path.join(d, "src.jpg")
        PILImage.fromarray(base).save(one, quality=90)
        paths = []
        for i in range(n):
            p = os.path.join(d, "f%d.jpg" % i)
            os.link(one, p)
            paths.append(p)
        print("%d files of %.1f MB (4000x3000 -> 256x256, shrink-on-load 8)" % (n, os.path.getsize(one) / 1e6))
        for t in threads:
            Image.thumbnail_batch(paths[:min(n, t)], 256, 256, threads=t)  # warm: streams, tables
            t0 = time.perf_counter()
            outs = Image.thumbnail_batch(paths, 256, 256, threads=t)
            dt = time.perf_counter() - t0
            bad = sum(1 for o in outs if isinstance(o, Exception))
            print("threads %3d: %7.1f thumbnails/s  %6.1f Mpixel/s in  (%d failed)"
                  % (t, n / dt, n * 12.0 / dt, bad), flush=True)


if __name__ == "__main__":
    main()
