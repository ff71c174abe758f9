"""CPU: vips_reducev on uchar with a coefficient row per output row as a banded matrix product on the matrix
cores (libvips_amd/csrc/reduce_band_body.h) run thread by thread on host fibers (tests/emul: the matrix
instruction emulated as a wave meeting) under the mock HIP runtime and compared, whole image, bit for bit, with
the compiled reference."""
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
for (w, h, bands, shrink, kernel, gate) in %(cases)r:
    dt = np.uint16 if "u16" in gate else np.uint8
    src = helpers.lcg_image(w, h, bands, dt, 11 + w)
    src[: h // 3, : w // 2] = 65535 if dt == np.uint16 else 255
    src[h // 3: h // 2, w // 2:] = 0
    im = Image.new_from_array(src)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    if gate.startswith("reduceh"):
        got = im.reduceh(shrink, kernel=kernel).numpy()
    else:
        got = im.reducev(shrink, kernel=kernel).numpy()
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    if gate.startswith("reduceh"):
        want = helpers.Ref.run_chain("reduceh:hshrink=%%r,kernel=%%s" %% (shrink, kernel), src)
    else:
        want = helpers.Ref.run_chain("reducev:vshrink=%%r,kernel=%%s" %% (shrink, kernel), src)
    assert list(report) == [gate], (w, h, bands, shrink, kernel, report)
    if not gate.endswith("_band"):
        continue
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (w, h, bands, shrink, kernel, len(bad), bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])
print("CHILD-OK")
'''

S = "reducev_u8_band"
# (width, height, bands, shrink, kernel, the kernel that must have run)
CASES = [
    (512, 733, 3, 7.3, "lanczos3", S), (304, 260, 1, 3.7, "linear", S), (300, 200, 4, 2.5, "cubic", S),
    (1024, 333, 2, 5.1, "mitchell", S), (96, 415, 3, 11.7, "lanczos2", S),
    # rows of whole dwords that are not whole 128-byte strips / 8-byte groups; a tall image (many blocks)
    (516, 333, 3, 7.3, "lanczos3", S), (100, 1500, 1, 4.3, "lanczos3", S), (44, 90, 3, 2.2, "lanczos3", S),
    # a shrink whose coefficients are not exact halves (>= 2048): the vector-ALU kernel; a constant phase: the
    # matrix-core kernel of reduce_u8.hip
    (2104, 90, 4, 1.6, "lanczos3", "reducev_u8_stream"), (512, 512, 4, 8.0, "lanczos3", "reducev_u8_mfma"),
]


def _run(cases, tmp_path, extra_env=None):
    script = os.path.join(str(tmp_path), "child.py")
    with open(script, "w") as f:
        f.write(CHILD % {"root": helpers.ROOT, "cases": cases})
    env = dict(os.environ, LD_PRELOAD=MOCK_SO, VIPS_HIP_LIBRARY=EMUL_SO)
    env.pop("VIPS_HIP_REDUCE_BAND", None)
    env.update(extra_env or {})
    proc = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          env=env, timeout=1800)
    assert proc.returncode == 0 and "CHILD-OK" in proc.stdout, proc.stdout[-3000:]


H = "reduceh_u8_band"
HCASES = [
    # (input rows of whole dwords; the output's need not be)
    (732, 100, 3, 7.3, "lanczos3", H), (260, 64, 1, 3.7, "linear", H), (200, 37, 4, 2.5, "cubic", H),
    (334, 50, 2, 5.1, "mitchell", H), (416, 33, 3, 11.7, "lanczos2", H), (1500, 20, 1, 4.3, "lanczos3", H),
    (92, 44, 3, 2.2, "lanczos3", H), (1200, 18, 3, 16.5, "lanczos3", H), (744, 21, 3, 7.3, "lanczos3", H),
    # coefficients that are not exact halves: the byte-at-a-time kernel; a constant phase: the packed-byte kernel
    (300, 40, 4, 1.6, "lanczos3", "reduceh_u8_lds"), (512, 40, 4, 8.0, "lanczos3", "reduceh_u8_packed"),
]


V16, H16 = "reducev_u16_band", "reduceh_u16_band"
CASES16 = [
    (512, 733, 3, 7.3, "lanczos3", V16), (304, 260, 1, 3.7, "linear", V16), (300, 200, 4, 2.5, "cubic", V16),
    (130, 415, 2, 11.7, "lanczos2", V16), (256, 300, 4, 8.0, "lanczos3", V16), (62, 90, 3, 2.2, "lanczos3", V16),
    (732, 100, 3, 7.3, "lanczos3", H16), (260, 64, 2, 3.7, "linear", H16), (200, 37, 4, 2.5, "cubic", H16),
    (416, 33, 3, 11.7, "lanczos2", H16), (1500, 20, 2, 4.3, "lanczos3", H16), (512, 24, 4, 8.0, "lanczos3", H16),
    (92, 44, 3, 2.2, "lanczos3", H16),
]


def test_reducev_band(tmp_path):
    _run(CASES, tmp_path)


def test_reduce_band_ushort(tmp_path):
    _run(CASES16, tmp_path)


def test_reduceh_band(tmp_path):
    _run(HCASES, tmp_path)


RESIZE_CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import libvips_amd
from libvips_amd import Image
from tests import helpers

libvips_amd.init(0)
lib = libvips_amd.lib
for (w, h, bands, scale, vscale, gates) in %(cases)r:
    src = helpers.lcg_image(w, h, bands, np.uint8, 5 + w)
    src[: h // 3, : w // 2] = 255
    src[h // 3: h // 2, w // 2:] = 0
    im = Image.new_from_array(src)
    kw = {} if vscale is None else {"vscale": vscale}
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    got = im.resize(scale, **kw).numpy()
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    chain = "resize:scale=%%r" %% scale + ("" if vscale is None else ",vscale=%%r" %% vscale)
    want = helpers.Ref.run_chain(chain, src)
    assert sorted(report) == sorted(gates), (w, h, bands, scale, vscale, report)
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (w, h, bands, scale, vscale, len(bad), bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])
print("CHILD-OK")
'''

BV, BH, SH = "shrinkv_reducev_u8_band", "reduceh_u8_band", "shrinkh_u8_stream"
# (width, height, bands, scale, vscale, the kernels that must have run): box shrinks 2 .. 16 in front of the
# banded reduce, heights that are not multiples of the shrink (the last box row is clipped), and beside them a box
# shrink the fused kernel does not have (25: the one-kernel chain)
RESIZE_CASES = [
    (512, 733, 3, 0.23, None, [BV, SH, BH]), (304, 1260, 1, 0.07, None, [BV, SH, BH]),
    (300, 611, 4, 0.15, None, [BV, SH, BH]), (1024, 333, 2, 0.3, 0.11, [BV, BH]),
    (96, 815, 3, 0.4, 0.07, [BV, BH]), (516, 1003, 3, 1.0 / 7.3, None, [BV, SH, BH]),
    (640, 480, 4, 0.45, 0.26, [S, BH]), (600, 401, 1, 0.09, 0.19, [BV, SH, BH]),
    (256, 1100, 3, 0.3, 1.0 / 12.5, [BV, BH]), (128, 1531, 4, 0.4, 0.061, [BV, BH]),
    (200, 900, 3, 0.3, 0.09, [BV, BH]), (512, 1290, 3, 0.04, None, [BV, SH, BH]),
    (128, 1800, 3, 0.3, 1 / 18.5, [BV, BH]), (64, 2100, 4, 0.3, 1 / 22.1, [BV, BH]), (64, 2500, 3, 0.3, 1 / 26.3, [BV, BH]),
    (64, 2900, 2, 0.4, 1 / 30.9, [BV, BH]), (60, 3100, 1, 0.4, 1 / 33.0, [BV, BH]), (128, 1900, 3, 0.3, 1 / 28.9, [BV, BH]),
    (512, 1290, 3, 0.0199, None, ["resize_streamg_u8"]),
]


def test_resize_band_chain(tmp_path):
    """vips_resize at a scale that leaves a fractional reduce on both axes: shrinkv + reducev as one banded
    matrix-core kernel, shrinkh, reduceh (ops_resample.cpp), whole image against the compiled reference."""
    script = os.path.join(str(tmp_path), "child.py")
    with open(script, "w") as f:
        f.write(RESIZE_CHILD % {"root": helpers.ROOT, "cases": RESIZE_CASES})
    env = dict(os.environ, LD_PRELOAD=MOCK_SO, VIPS_HIP_LIBRARY=EMUL_SO, VIPS_HIP_RESIZE_BAND_MIN="0")
    for k in ("VIPS_HIP_REDUCE_BAND", "VIPS_HIP_NO_RESIZE_BAND", "VIPS_HIP_STREAMG_ALWAYS"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          env=env, timeout=1800)
    assert proc.returncode == 0 and "CHILD-OK" in proc.stdout, proc.stdout[-3000:]
