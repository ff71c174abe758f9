#!/usr/bin/env python3
"""C4 shape on one GPU: independent 8192^2 RGB images through resize(1/8) + sharpen, T host
threads each with its own library stream (the libvips threadpool model: one worker per image).
Reports images/s and input Mpixels/s per thread count.  usage: bench_c4_threads.py [threads...]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import libvips_amd  # noqa: E402
from libvips_amd import Image, lib  # noqa: E402


def main():
    counts = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16]
    n = int(os.environ.get("C4_SIZE", "8192"))
    per_thread = int(os.environ.get("C4_IMAGES", "24"))
    libvips_amd.init(0)
    dev = torch.device("cuda", 0)
    src = bench.lcg_image_device(torch, n, n, 3, 12345, dev)
    torch.cuda.synchronize()
    im = Image.new_from_tensor(src, interpretation="srgb")
    im.resize(0.125).sharpen()
    libvips_amd.synchronize()
    for t in counts:
        barrier = threading.Barrier(t + 1)

        def worker():
            libvips_amd.init(0)
            lib.vips_hip_set_stream(None)  # this thread's own stream
            im.resize(0.125).sharpen()
            libvips_amd.synchronize()
            barrier.wait()
            for _ in range(per_thread):
                out = im.resize(0.125).sharpen()
            libvips_amd.synchronize()
            barrier.wait()
            del out

        threads = [threading.Thread(target=worker) for _ in range(t)]
        for th in threads:
            th.start()
        barrier.wait()
        t0 = time.perf_counter()
        barrier.wait()
        dt = time.perf_counter() - t0
        for th in threads:
            th.join()
        images = t * per_thread
        print("threads %2d: %7.1f images/s  %9.1f Mpixels/s (input)  %.3f ms/image" %
              (t, images / dt, images * n * n / dt / 1e6, dt / images * 1e3), flush=True)


if __name__ == "__main__":
    main()
