"""Host-side cost per operation (enqueue time vs enqueue + sync), C4 shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, libvips_amd
from libvips_amd import Image, lib
libvips_amd.init(0)
dev = torch.device("cuda", 0)
src = bench.lcg_image_device(torch, 8192, 8192, 3, 12345, dev)
torch.cuda.synchronize()
im = Image.new_from_tensor(src, interpretation="srgb")
small = im.resize(0.125)
libvips_amd.synchronize()
def t(name, fn, n=50):
    fn(); libvips_amd.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): o = fn()
    t1 = time.perf_counter()
    libvips_amd.synchronize()
    t2 = time.perf_counter()
    print("%-28s host %.3f ms/call, +sync %.3f ms/call" % (name, (t1-t0)/n*1e3, (t2-t0)/n*1e3))
t("resize(1/8)", lambda: im.resize(0.125))
t("sharpen 1024^2", lambda: small.sharpen())
t("colourspace labs", lambda: small.colourspace("labs"))
t("gaussblur(0.5) u8", lambda: small.gaussblur(0.5))
t("cast float", lambda: small.cast("float"))
t("shrinkv 4", lambda: im.shrinkv(4))
t("reducev 2 (on 8192x2048)", lambda: im.shrinkv(4).reducev(2.0))
