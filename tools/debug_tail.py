import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import libvips_amd
from libvips_amd import Image
import helpers
libvips_amd.init(0)
for bands, (w, h), scale in [(1, (512, 384), 0.125), (3, (512, 384), 0.125), (4, (512, 384), 0.125), (1, (512, 384), 0.5), (1, (2048, 1536), 0.125)]:
    src = helpers.lcg_image(w, h, bands, np.uint8, 20)
    got = Image.new_from_array(src).resize(scale).numpy()
    os.environ["VIPS_HIP_NO_RESIZE_TAIL"] = "1"
    exp = Image.new_from_array(src).resize(scale).numpy()
    os.environ.pop("VIPS_HIP_NO_RESIZE_TAIL")
    bad = got != exp
    if bad.ndim == 2:
        bad = bad[:, :, None]
    d = (got.astype(int) - exp.astype(int)).reshape(bad.shape)
    print(bands, (w, h), scale, "bad", int(bad.sum()), "of", bad.size, "rows", np.flatnonzero(bad.any(axis=(1, 2)))[:40],
          "cols", np.flatnonzero(bad.any(axis=(0, 2)))[:80], "bands", np.flatnonzero(bad.any(axis=(0, 1))),
          "diff range", d.min(), d.max())
