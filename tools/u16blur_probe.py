import sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import libvips_amd
from libvips_amd import Image
libvips_amd.init(0)
lib = libvips_amd.lib
t = torch.randint(0, 65536, (8192, 8192, 3), dtype=torch.int32, device="cuda").to(torch.uint16)
im = Image.new_from_tensor(t, interpretation="rgb16")
for sigma in (2.0, 8.0):
    lib.vips_hip_gate_reset(); lib.vips_hip_gate_enable(1)
    im.gaussblur(sigma)
    rep = libvips_amd.gate_report(); lib.vips_hip_gate_enable(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        im.gaussblur(sigma)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print("gaussblur sigma %g on 8192^2 x 3 ushort: %.3f ms = %.1f %% of 8 TB/s; %s" % (sigma, ms, 805.3e6 / (ms * 1e-3) / 8e12 * 100, rep), flush=True)
