"""CPU: the matrix-core separable convolution on uchar (libvips_amd/csrc/conv_u8_mfma_body.h: both passes of
vips_gaussblur / vips_convsep as Toeplitz products on v_mfma_f32_32x32x16_f16) run thread by thread on host
fibers (tests/emul: the matrix instruction and the LDS-DMA emulated) under the mock HIP runtime and compared,
whole image, bit for bit, with the compiled reference."""
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
    u16 = kind.endswith("16")
    kind = kind[:-2] if u16 else kind
    src = helpers.lcg_image(w, h, bands, np.uint16 if u16 else np.uint8, 7 + w)
    if len(case) > 5 and case[5] == "flat":
        src[: h // 2] = 65535 if u16 else 255
        src[h // 2:, : w // 3] = 0
    im = Image.new_from_array(src)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    gate = "conv_u16_mfma_sep" if u16 else "conv_u8_mfma_sep"
    if kind == "blur":
        sigma = case[4]
        got = im.gaussblur(sigma).numpy()
        want = helpers.Ref.run_chain("gaussblur:sigma=%%r" %% sigma, src)
    elif kind == "conv":
        mask, scale = case[4]
        m = np.asarray(mask, dtype=np.float64)
        got = im.conv(m, scale=scale, precision="integer").numpy()
        want = helpers.Ref.run_mask("conv", src, m, scale, 0.0, "precision=integer")
        gate = "conv_u8_mfma_2d"
    else:
        mask, scale = case[4]
        got = im.convsep(mask, scale=scale, precision="integer").numpy()
        want = helpers.Ref.run_mask("convsep", src, np.asarray(mask, dtype=np.float64)[None, :], scale, 0.0, "precision=integer")
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
    env = dict(os.environ, LD_PRELOAD=MOCK_SO, VIPS_HIP_LIBRARY=EMUL_SO)
    env.pop("VIPS_HIP_CONV_U8_MFMA", None)
    env.update(extra_env or {})
    proc = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          env=env, timeout=1800)
    assert proc.returncode == 0 and "CHILD-OK" in proc.stdout, proc.stdout[-3000:]


def test_gaussblur_and_convsep(tmp_path):
    # sigma 1 .. 8 (3 .. 29 taps), 1 .. 4 bands, several strips (one of them ragged) and chunks, images narrower
    # than a strip, a mask with negative taps, a 33-tap mask (the widest window: hp = half = 16)
    _run([("blur", 300, 70, 3, 2.0), ("blur", 332, 41, 1, 1.0), ("blur", 271, 37, 4, 2.0),
          ("blur", 260, 100, 3, 4.0), ("blur", 200, 80, 2, 6.0), ("blur", 300, 140, 3, 8.0),
          ("blur", 96, 33, 3, 2.0), ("blur", 256, 64, 3, 2.0, "flat"),
          ("sep", 200, 50, 3, ([1, -3, 9, -3, 1], 5)), ("sep", 260, 40, 1, ([5, 1, 5], 11)),
          ("sep", 172, 70, 3, (list(range(1, 18)) + list(range(16, 0, -1)), 289))], tmp_path)


def test_rows_of_whole_16_byte_units(tmp_path):
    # row starts that are multiples of 16 bytes: the staging moves 16 bytes per lane (global_load_lds_dwordx4)
    _run([("blur", 320, 70, 3, 8.0), ("blur", 256, 64, 3, 2.0), ("blur", 272, 45, 4, 4.0), ("blur", 400, 40, 2, 6.0),
          ("blur", 336, 100, 1, 3.0), ("sep", 176, 70, 3, (list(range(1, 18)) + list(range(16, 0, -1)), 289))], tmp_path)
    _run([("blur", 320, 200, 3, 8.0)], tmp_path, {"VIPS_HIP_CONV_MFMA_SEG": "2"})
    # ... and the same images through dword units
    _run([("blur", 320, 70, 3, 8.0), ("blur", 336, 100, 1, 3.0)], tmp_path, {"VIPS_HIP_CONV_MFMA_NARROW": "1"})


K3 = [[-1, -1, -1], [-1, 16, -1], [-1, -1, -1]]
K5 = [[1, 4, 6, 4, 1], [4, 16, 24, 16, 4], [6, 24, 36, 24, 6], [4, 16, 24, 16, 4], [1, 4, 6, 4, 1]]
K37 = [[1, 2, 3, 4, 3, 2, 1], [-2, -1, 0, 9, 0, -1, -2], [1, 2, 3, 4, 3, 2, 1]]
K9x15 = [[(i * 7 + j * 3) % 11 - 2 for j in range(15)] for i in range(9)]


def test_conv_2d(tmp_path):
    # two-dimensional masks: 3 x 3, 5 x 5, 7 wide x 3, 15 wide x 9 (window steps 3 and 4), 1 .. 4 bands, narrow and
    # ragged images, 16-byte and dword staging units, one chunk per segment
    _run([("conv", 300, 70, 3, (K3, 8)), ("conv", 332, 41, 1, (K5, 256)), ("conv", 271, 37, 4, (K3, 8)),
          ("conv", 260, 50, 3, (K5, 256)), ("conv", 200, 45, 2, (K37, 30)), ("conv", 96, 33, 3, (K3, 8)),
          ("conv", 256, 64, 3, (K3, 8), "flat"), ("conv", 320, 100, 3, (K5, 256)),
          ("conv", 176, 80, 3, (K9x15, sum(map(sum, K9x15))))], tmp_path)
    _run([("conv", 300, 120, 3, (K5, 256))], tmp_path, {"VIPS_HIP_CONV_MFMA_SEG": "1"})


def test_short_segments(tmp_path):
    # two chunks per segment: every segment boundary inside the image, top and bottom rows clamped
    _run([("blur", 300, 200, 3, 2.0), ("blur", 160, 230, 3, 8.0)], tmp_path, {"VIPS_HIP_CONV_MFMA_SEG": "2"})


def test_ushort_gaussblur_and_convsep(tmp_path):
    # the same kernel on ushort images (2 x bands byte planes, two exact products per sample and pass, the rounding
    # in integers): sigma 1 .. 8, 1 .. 4 bands, ragged strips, saturated and zero areas, a mask with negative taps,
    # the widest window; 16-byte and dword staging units; segment boundaries inside the image
    _run([("blur16", 300, 70, 3, 2.0), ("blur16", 332, 41, 1, 1.0), ("blur16", 271, 37, 4, 2.0),
          ("blur16", 260, 100, 3, 4.0), ("blur16", 200, 80, 2, 6.0), ("blur16", 300, 140, 3, 8.0),
          ("blur16", 96, 33, 3, 2.0), ("blur16", 256, 64, 3, 2.0, "flat"), ("blur16", 320, 70, 4, 8.0, "flat"),
          ("sep16", 200, 50, 3, ([1, -3, 9, -3, 1], 5)), ("sep16", 260, 40, 1, ([5, 1, 5], 11)),
          ("sep16", 172, 70, 3, (list(range(1, 18)) + list(range(16, 0, -1)), 289))], tmp_path)
    _run([("blur16", 320, 70, 3, 8.0), ("blur16", 336, 100, 1, 3.0)], tmp_path, {"VIPS_HIP_CONV_MFMA_NARROW": "1"})
    _run([("blur16", 300, 200, 3, 2.0), ("blur16", 160, 230, 2, 8.0)], tmp_path, {"VIPS_HIP_CONV_MFMA_SEG": "2"})
