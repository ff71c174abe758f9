"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref).

Run here (the container with /root/reference): `python tests/golden/make_golden.py`.
The vectors pin oracle/port (tests/test_oracle.py) and the HIP path
(tests/test_*_gpu.py) on boxes where neither /root/reference nor oracle/_ref exist.
Inputs are regenerated from (shape, dtype, seed) by tests.helpers.lcg_image, so
only the reference OUTPUTS are stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests import helpers  # noqa: E402
from tests.golden import cases  # noqa: E402


def main():
    out = {}
    for case in cases.RESAMPLE_CASES:
        src = helpers.lcg_image(case["width"], case["height"], case["bands"], case["dtype"], case["seed"])
        got = helpers.Ref.run(case["op"], src, case["args"])
        out[case["name"]] = got
    np.savez_compressed(os.path.join(helpers.GOLDEN, "resample.npz"), **out)
    print("resample.npz: %d cases" % len(out))

    if hasattr(cases, "extra_groups"):
        for fname, fn in cases.extra_groups():
            group = fn()
            np.savez_compressed(os.path.join(helpers.GOLDEN, fname), **group)
            print("%s: %d cases" % (fname, len(group)))


if __name__ == "__main__":
    main()
