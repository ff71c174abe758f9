#!/usr/bin/env python3
"""The boundary hand-off of reduce_fused_u8x4_mfma_x under load: many launches of BASELINE config 2 back to back
(every launch's tiles arrive at their boundaries in another order), every output compared word for word with the
kernel that has halos; optionally beside a second stream that keeps some CUs busy (uneven load)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import libvips_amd  # noqa: E402
from bench import lcg_image_device  # noqa: E402
from libvips_amd import Image  # noqa: E402

libvips_amd.init(0)
dev = torch.device("cuda", 0)
n = int(os.environ.get("TUNE_SIZE", "16384"))
rounds = int(os.environ.get("TUNE_ROUNDS", "200"))
src = lcg_image_device(torch, n, n, 4, 12345, dev)
im = Image.new_from_tensor(src)
os.environ["VIPS_HIP_FUSED_EXCH"] = "0"
want = torch.from_numpy(im.reduce(8.0, 8.0, kernel="lanczos3").numpy()).to(dev)
os.environ["VIPS_HIP_FUSED_EXCH"] = "1"
noise = torch.empty((4096, 4096), device=dev)
side = torch.cuda.Stream(device=dev)
bad = 0
for k in range(rounds):
    if os.environ.get("TUNE_LOAD") and k % 3 == 0:
        with torch.cuda.stream(side):  # something else on the part: tiles no longer start and end together
            noise.normal_()
            noise = noise @ noise[:, :512].repeat(1, 8) * 1e-3
    outs = [im.reduce(8.0, 8.0, kernel="lanczos3") for _ in range(4)]
    libvips_amd.synchronize()
    for o in outs:
        got = torch.from_numpy(o.numpy()).to(dev)
        if not torch.equal(got, want):
            bad += 1
            d = (got != want).any(dim=-1).nonzero()
            print("round %d: %d pixels differ, first at %s" % (k, d.shape[0], d[0].tolist()), flush=True)
print("%d launches, %d with a difference" % (4 * rounds, bad))
