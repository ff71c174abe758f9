#!/usr/bin/env python3
"""Where does a C5 slab step spend its host time?  (conv plan build, output allocation, kernel)"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import libvips_amd  # noqa: E402
from libvips_amd import PRECISIONS, lib, sharding  # noqa: E402
from libvips_amd._ffi import check_handle  # noqa: E402

libvips_amd.init(0)
dev = torch.device("cuda", 0)
mask, scale = libvips_amd.gaussmat(5, 0.01, False, "float")
plan = sharding.StripPlan(65536, 65536, 8, sharding.conv_need(31, 65536))
w0, w1 = plan.windows[4]
window = bench.c5_rows_device(torch, 65536, w0, w1 - w0, dev)
torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    m = np.ascontiguousarray(np.asarray(mask, dtype=np.float64))
    conv = check_handle(lib.vips_hip_conv_new(m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 31, 31, float(scale), 0.0,
                                              PRECISIONS["float"]))
    t1 = time.perf_counter()
    lib.vips_hip_conv_free(conv)
    t2 = time.perf_counter()
    out = sharding.conv_strip(window, w0, plan, 4, mask, scale=scale, precision="float")
    t3 = time.perf_counter()
    print("conv_new %.2f ms  conv_free %.2f ms  conv_strip (all) %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
