#!/usr/bin/env python3
"""C4: do the HBM-bound resize chain and the ALU-bound sharpen fit side by side if each gets its
own CUs?  Times a 64-image launch of each on streams restricted to N of the 256 CUs
(hipExtStreamCreateWithCUMask; bit i of the mask = CU i/8 of XCD i%8 on this part), then the two
together on disjoint masks.  usage: python tools/c4_cumask.py
"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import libvips_amd  # noqa: E402
from libvips_amd import Image, lib  # noqa: E402

hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(lo, hi):
    """A stream that may use CUs lo .. hi-1 (mask bits)."""
    words = (ctypes.c_uint32 * 8)()
    for b in range(lo, hi):
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    r = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert r == 0, r
    return s


def main():
    n, count = 8192, 64
    libvips_amd.init(0)
    dev = torch.device("cuda", 0)
    store = torch.empty((count, n, n, 3), dtype=torch.uint8, device=dev)
    for k in range(count):
        bench.lcg_image_device(torch, n, n, 3, 12345 + k, dev, out=store[k])
    torch.cuda.synchronize()
    ims = [Image.new_from_tensor(store[k], interpretation="srgb") for k in range(count)]
    smalls = libvips_amd.resize_sharpen_batch(ims, 0.125, sharpen=False)
    libvips_amd.synchronize()

    def timed(fn, reps=5):
        fn()
        libvips_amd.synchronize()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            libvips_amd.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3 / count

    def resize():
        return libvips_amd.resize_sharpen_batch(ims, 0.125, sharpen=False)

    def sharpen():
        return [im.sharpen() for im in smalls]  # one launch per thumbnail

    for cus in (256, 224, 192, 160, 128):
        s = masked_stream(0, cus)
        lib.vips_hip_set_stream(s)
        print("resize  on %3d CUs: %.4f ms/image" % (cus, timed(resize)), flush=True)
    for cus in (256, 128, 96, 64, 32):
        s = masked_stream(256 - cus, 256)
        lib.vips_hip_set_stream(s)
        print("sharpen on %3d CUs: %.4f ms/image" % (cus, timed(sharpen)), flush=True)


if __name__ == "__main__":
    main()
