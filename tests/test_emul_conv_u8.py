"""CPU: the packed-byte integer convolutions on uchar (libvips_amd/csrc/conv_u8_body.h: both passes
of vips_gaussblur / vips_convsep, and vips_conv with a small 2-D mask) run thread by thread on host
fibers (tests/emul) under the mock HIP runtime and compared, whole image, bit for bit, with the
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
for case in %(cases)r:
    kind, w, h, bands = case[:4]
    src = helpers.lcg_image(w, h, bands, np.uint8, 7 + w)
    if len(case) > 5 and case[5] == "flat":
        src[: h // 2] = 255
        src[h // 2:, : w // 3] = 0
    im = Image.new_from_array(src)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    if kind == "blur":
        sigma = case[4]
        got = im.gaussblur(sigma).numpy()
        want = helpers.Ref.run_chain("gaussblur:sigma=%%r" %% sigma, src)
        gate = "conv_u8_sep"
    elif kind == "sep":
        mask, scale = case[4]
        got = im.convsep(mask, scale=scale, precision="integer").numpy()
        want = helpers.Ref.run_mask("convsep", src, np.asarray(mask, dtype=np.float64)[None, :], scale, 0.0, "precision=integer")
        gate = "conv_u8_sep"
    else:
        mask, scale = case[4]
        m = np.asarray(mask, dtype=np.float64)
        got = im.conv(m, scale=scale, precision="integer").numpy()
        want = helpers.Ref.run_mask("conv", src, m, scale, 0.0, "precision=integer")
        gate = "conv_u8_2d"
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    assert list(report) == [gate], (case, report)
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (case[:4], len(bad), bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])
print("CHILD-OK")
'''


def _run(cases, tmp_path, extra_env=None):
    script = os.path.join(str(tmp_path), "child.py")
    with open(script, "w") as f:
        f.write(CHILD % {"root": helpers.ROOT, "cases": cases})
    # (this file is about the packed-byte kernels: the matrix-core one, which takes these masks first, is
    # tests/test_emul_conv_u8_mfma.py's)
    env = dict(os.environ, LD_PRELOAD=MOCK_SO, VIPS_HIP_LIBRARY=EMUL_SO, VIPS_HIP_CONV_U8_MFMA="0")
    env.update(extra_env or {})
    proc = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          env=env, timeout=1800)
    assert proc.returncode == 0 and "CHILD-OK" in proc.stdout, proc.stdout[-3000:]


K3 = [[-1, -1, -1], [-1, 16, -1], [-1, -1, -1]]
K5 = [[1, 4, 6, 4, 1], [4, 16, 24, 16, 4], [6, 24, 36, 24, 6], [4, 16, 24, 16, 4], [1, 4, 6, 4, 1]]
K37 = [[1, 2, 3, 4, 3, 2, 1], [-2, -1, 0, 9, 0, -1, -2], [1, 2, 3, 4, 3, 2, 1]]


def test_gaussblur_and_convsep(tmp_path):
    # sigma 1 .. 8 (3 .. 29 taps: 3, 5, 7 and 9 window dwords), 1 / 3 / 4 bands (rows of whole dwords:
    # anything else takes the older kernels), several strips and segments, images smaller than a strip
    _run([("blur", 1100, 70, 3, 2.0), ("blur", 332, 41, 1, 1.0), ("blur", 2071, 37, 4, 2.0),
          ("blur", 1028, 150, 3, 4.0), ("blur", 600, 130, 3, 6.0), ("blur", 532, 140, 3, 8.0),
          ("blur", 96, 33, 3, 2.0), ("blur", 640, 64, 3, 2.0, "flat"),
          ("sep", 700, 50, 3, ([1, -3, 9, -3, 1], 5)), ("sep", 260, 40, 1, ([5, 1, 5], 11))], tmp_path)


def test_gaussblur_rings_in_lds(tmp_path):
    # the longer masks' rings of transposed quads in LDS (the default keeps them in registers)
    _run([("blur", 1028, 150, 3, 4.0), ("blur", 532, 140, 3, 8.0), ("blur", 600, 90, 1, 6.0)], tmp_path,
         {"VIPS_HIP_CONV_U8_RING": "lds"})


def test_gaussblur_short_segments(tmp_path):
    _run([("blur", 1100, 200, 3, 2.0), ("blur", 532, 200, 3, 8.0)], tmp_path, {"VIPS_HIP_CONV_U8_SEG": "3"})
    _run([("blur", 1100, 100, 3, 2.0)], tmp_path, {"VIPS_HIP_CONV_U8_SEG": "1"})


def test_conv_2d(tmp_path):
    _run([("conv", 1100, 70, 3, (K3, 8)), ("conv", 332, 41, 1, (K5, 256)), ("conv", 2071, 37, 4, (K3, 8)),
          ("conv", 1028, 50, 3, (K5, 256)), ("conv", 600, 45, 3, (K37, 30)), ("conv", 96, 33, 3, (K3, 8)),
          ("conv", 640, 64, 3, (K3, 8), "flat")], tmp_path)
    _run([("conv", 1100, 120, 3, (K5, 256))], tmp_path, {"VIPS_HIP_CONV_U8_SEG": "2"})
