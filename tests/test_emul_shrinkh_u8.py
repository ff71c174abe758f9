"""CPU: the packed-byte vips_shrinkh on uchar (libvips_amd/csrc/shrinkh_u8_body.h: a lane owns 4
output pixels, box sums are v_dot4_u32_u8 of planar dwords with byte masks) run thread by thread on
host fibers (tests/emul) under the mock HIP runtime and compared, whole image, bit for bit, with the
compiled reference.  See tests/test_emul_resize_sharpen.py for how the emulation is built."""
import os
import subprocess
import sys

import pytest

from tests import helpers
from tests.test_emul_resize_sharpen import EMUL_SO, _build_emul
from tests.test_host_glue_mock import MOCK_SO, _build_mock, _gpu_present

pytestmark = pytest.mark.skipif(_gpu_present() or not helpers.have_ref() or not _build_mock() or not _build_emul(),
                                reason="a real GPU is present, or the reference / mock runtime / emulation cannot be built")

CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import libvips_amd
from libvips_amd import Image
from tests import helpers

libvips_amd.init(0)
lib = libvips_amd.lib
for (w, h, bands, hs, ceil, gate) in %(cases)r:
    src = helpers.lcg_image(w, h, bands, np.uint8, 11 + w)
    src[: h // 3, : w // 2] = 255          # the largest sums
    src[h // 3: h // 2, w // 2:] = 0
    im = Image.new_from_array(src)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    got = im.shrinkh(hs, ceil=bool(ceil)).numpy()
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    want = helpers.Ref.run_chain("shrinkh:hshrink=%%d%%s" %% (hs, ",ceil=true" if ceil else ""), src)
    assert list(report) == [gate], (w, h, bands, hs, ceil, report)
    if "general" in gate:
        continue  # (the older kernel is not emulated: under the mock runtime it makes no pixels)
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (w, h, bands, hs, len(bad), bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])
print("CHILD-OK")
'''

S, G = "shrinkh_u8_stream", "shrinkh_general"
# (width, height, bands, hshrink, ceil, the kernel that must have run); library-made images have
# tight rows, and the kernel wants rows of whole dwords on both sides
CASES = [
    # every compiled-in factor on 3 bands
    (1024, 37, 3, 2, 0, S), (1020, 21, 3, 3, 0, S), (1024, 40, 3, 4, 0, S), (1040, 11, 3, 5, 0, S), (1104, 9, 3, 6, 0, S),
    (1008, 8, 3, 7, 0, S), (2048, 13, 3, 8, 0, S),
    # a last box that runs over the image's edge (ceil: vips_embed COPY), output rows of whole dwords
    (1036, 9, 3, 5, 1, S), (1020, 7, 3, 8, 1, S), (1004, 6, 1, 7, 1, S),
    # 1, 2 and 4 bands; partial last quads (4 bands: any width; 2 bands: even widths)
    (708, 41, 1, 3, 0, S), (1024, 33, 1, 4, 0, S), (408, 19, 2, 3, 0, S), (412, 19, 2, 2, 0, S), (1024, 16, 4, 4, 0, S),
    (1031, 7, 4, 5, 0, S), (1031, 7, 4, 5, 1, S), (518, 9, 4, 2, 0, S), (93, 5, 4, 3, 1, S),
    # boxes of whole groups (a run-time factor), and factors this kernel leaves to the general one
    (4608, 9, 3, 12, 0, S), (4100, 5, 4, 16, 1, S), (3200, 6, 1, 20, 0, S), (2016, 5, 3, 9, 0, G), (2000, 5, 3, 10, 0, G),
    # rows that are not whole dwords
    (1023, 10, 1, 2, 0, G), (1022, 10, 3, 2, 0, G), (1030, 11, 3, 5, 0, G),
    # an image narrower than one quad's boxes, and one with a single row
    (12, 5, 4, 4, 0, S), (4096, 1, 3, 4, 0, S),
]


def test_shrinkh_u8(tmp_path):
    script = os.path.join(str(tmp_path), "child.py")
    with open(script, "w") as f:
        f.write(CHILD % {"root": helpers.ROOT, "cases": CASES})
    env = dict(os.environ, LD_PRELOAD=MOCK_SO, VIPS_HIP_LIBRARY=EMUL_SO)
    proc = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          env=env, timeout=1800)
    assert proc.returncode == 0 and "CHILD-OK" in proc.stdout, proc.stdout[-3000:]
