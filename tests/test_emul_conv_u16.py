"""CPU: the streaming integer convolution on ushort (libvips_amd/csrc/conv_u16_body.h: vips_conv with
precision=integer and a mask up to 5 x 5, v_dot2_i32_i16 on planar signed lanes, division by the scale
as a multiply-high) run thread by thread on host fibers (tests/emul) under the mock HIP runtime and
compared, whole image, bit for bit, with the compiled reference."""
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
for (w, h, bands, mask, scale, gate, flat) in %(cases)r:
    src = helpers.lcg_image(w, h, bands, np.uint16, 7 + w)
    if flat:
        src[: h // 2] = 65535               # the largest sums: clipping at the top, negative lobes at the edge
        src[h // 2:, : w // 3] = 0
    im = Image.new_from_array(src)
    m = np.asarray(mask, dtype=np.float64)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    got = im.conv(m, scale=scale, precision="integer").numpy()
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    want = helpers.Ref.run_mask("conv", src, m, scale, 0.0, "precision=integer")
    assert list(report) == [gate], (w, h, bands, scale, report)
    if gate != "conv_u16_2d":
        continue  # (the general kernel is not emulated: under the mock runtime it makes no pixels)
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape, got.dtype, want.dtype)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (w, h, bands, scale, len(bad), bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])
print("CHILD-OK")
'''

K3 = [[-1, -1, -1], [-1, 16, -1], [-1, -1, -1]]
K5 = [[1, 4, 6, 4, 1], [4, 16, 24, 16, 4], [6, 24, 36, 24, 6], [4, 16, 24, 16, 4], [1, 4, 6, 4, 1]]
K35 = [[1, 2, 3, 2, 1], [-2, 0, 9, 0, -2], [1, 2, 3, 2, 1]]
K53 = [[1, -2, 1], [2, 0, 2], [3, 9, 3], [2, 0, 2], [1, -2, 1]]
K13 = [[1, 2, 1]]
K7 = [[1, 1, 1, 1, 1, 1, 1]] * 3
C = "conv_u16_2d"
# (width, height, bands, mask, scale, the kernel that must have run, flat areas)
CASES = [
    (1100, 70, 3, K3, 8, C, 0), (332, 41, 1, K5, 256, C, 0), (2070, 37, 4, K3, 8, C, 0), (1028, 50, 3, K5, 256, C, 0),
    (600, 45, 3, K35, 17, C, 0), (400, 60, 4, K53, 21, C, 0), (96, 33, 3, K3, 8, C, 0), (640, 64, 3, K3, 8, C, 1),
    (640, 64, 1, K5, 255, C, 1), (500, 40, 3, K3, 1, C, 1), (310, 30, 3, K13, 4, C, 0), (2, 9, 4, K3, 8, C, 0),
    # wider than 5 columns: the general kernel
    (300, 30, 3, K7, 21, "convi", 0),
]


def _run(cases, tmp_path, extra_env=None):
    script = os.path.join(str(tmp_path), "child.py")
    with open(script, "w") as f:
        f.write(CHILD % {"root": helpers.ROOT, "cases": cases})
    env = dict(os.environ, LD_PRELOAD=MOCK_SO, VIPS_HIP_LIBRARY=EMUL_SO)
    env.update(extra_env or {})
    proc = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          env=env, timeout=1800)
    assert proc.returncode == 0 and "CHILD-OK" in proc.stdout, proc.stdout[-3000:]


def test_conv_u16(tmp_path):
    _run(CASES, tmp_path)


def test_conv_u16_short_segments(tmp_path):
    _run([c for c in CASES if c[5] == C and c[1] > 30], tmp_path, {"VIPS_HIP_CONV_U16_SEG": "7"})
