#!/usr/bin/env python3
"""Which CUs does a bit of hipExtStreamCreateWithCUMask name on this part?  The sharpen of 64
thumbnails (per-thumbnail launches) and a 64-image resize launch on masks of 64 CUs made two ways:
bits 192..255, and the bits with i % 8 >= 6 -- one of them is two whole XCDs (L2s of their own), the
other eight CUs of every XCD.  usage: python tools/c4_masks.py"""
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


def stream_of(bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words) == 0
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

    masks = {
        "all 256": range(256),
        "bits 192..255": range(192, 256),
        "bits i % 8 >= 6": [i for i in range(256) if i % 8 >= 6],
        "bits 0..63": range(64),
        "bits i % 8 < 2": [i for i in range(256) if i % 8 < 2],
        "bits i % 4 == 3": [i for i in range(256) if i % 4 == 3],
        "bits i % 32 >= 24": [i for i in range(256) if i % 32 >= 24],
        "bits 128..255": range(128, 256),
        "bits i % 8 >= 4": [i for i in range(256) if i % 8 >= 4],
        "bits 0..191": range(192),
        "bits i % 8 < 6": [i for i in range(256) if i % 8 < 6],
    }
    for name, bits in masks.items():
        lib.vips_hip_set_stream(stream_of(list(bits)))
        s = timed(lambda: [im.sharpen() for im in smalls])
        r = timed(lambda: libvips_amd.resize_sharpen_batch(ims, 0.125, sharpen=False))
        print("%-20s (%3d bits): sharpen %.4f  resize %.4f ms/image" % (name, len(list(bits)), s, r), flush=True)


if __name__ == "__main__":
    main()
