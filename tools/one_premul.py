#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, libvips_amd
from libvips_amd import Image
libvips_amd.init(0)
dev = torch.device("cuda", 0)
t = bench.lcg_image_device(torch, 8192, 8192, 4, 7, dev)
torch.cuda.synchronize()
im4 = Image.new_from_tensor(t, interpretation="srgb")
for _ in range(4):
    o = im4.premultiply()
u = t + 1
libvips_amd.synchronize(); torch.cuda.synchronize()
