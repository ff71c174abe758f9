#!/usr/bin/env python3
"""Bit-equality of the C2 kernel's variants (the env knobs of tools/tune_c2.py) with the default build on a 4096^2
and a ragged image: usage  python tools/check_c2_variants.py "NAME:VAR=val,..." ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import libvips_amd  # noqa: E402
from libvips_amd import Image  # noqa: E402
from tests import helpers  # noqa: E402

libvips_amd.init(0)
for (w, h) in ((4096, 4096), (16384, 2048), (3000, 1234)):
    src = helpers.lcg_image(w, h, 4, np.uint8, 5)
    im = Image.new_from_array(src)
    want = im.reduce(8.0, 8.0, kernel="lanczos3").numpy()
    for spec in sys.argv[1:]:
        name, _, rest = spec.partition(":")
        env = dict(kv.split("=") for kv in rest.split(",") if kv)
        os.environ.update(env)
        got = im.reduce(8.0, 8.0, kernel="lanczos3").numpy()
        for k in env:
            os.environ.pop(k, None)
        print("%dx%d %-20s %s" % (w, h, name, "same" if np.array_equal(got, want) else "DIFFERENT"), flush=True)
