"""CPU: the streaming vertical reduce on uchar with a coefficient row per output row
(libvips_amd/csrc/resample16_body.h reducev8_body: the ushort kernel's walk and host-made schedule on
bytes) run thread by thread on host fibers (tests/emul) under the mock HIP runtime and compared,
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
for (w, h, bands, shrink, kernel, gate) in %(cases)r:
    src = helpers.lcg_image(w, h, bands, np.uint8, 11 + w)
    src[: h // 3, : w // 2] = 255
    src[h // 3: h // 2, w // 2:] = 0
    im = Image.new_from_array(src)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    got = im.reducev(shrink, kernel=kernel).numpy()
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    want = helpers.Ref.run_chain("reducev:vshrink=%%r,kernel=%%s" %% (shrink, kernel), src)
    assert list(report) == [gate], (w, h, bands, shrink, kernel, report)
    if gate not in ("reducev_u8_stream", "reducev_u8_mfma"):
        continue  # (the older kernels are not emulated: under the mock runtime they make no pixels)
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (w, h, bands, shrink, kernel, len(bad), bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])
print("CHILD-OK")
'''

S = "reducev_u8_stream"
# (width, height, bands, shrink, kernel, the kernel that must have run)
CASES = [
    (512, 733, 3, 7.3, "lanczos3", S), (304, 260, 1, 3.7, "linear", S), (300, 200, 4, 2.5, "cubic", S), (2104, 90, 4, 1.6, "lanczos3", S),
    (1024, 333, 2, 5.1, "mitchell", S), (96, 415, 3, 11.7, "lanczos2", S),
    # rows that are not whole 8-byte groups; a constant phase (the matrix-core kernel's case)
    (516, 333, 3, 7.3, "lanczos3", "reducev_u8"), (512, 512, 4, 8.0, "lanczos3", "reducev_u8_mfma"),
]


def _run(cases, tmp_path, extra_env=None):
    script = os.path.join(str(tmp_path), "child.py")
    with open(script, "w") as f:
        f.write(CHILD % {"root": helpers.ROOT, "cases": cases})
    # (this file is about the streaming kernel on the vector ALU: the matrix-core one, which takes these images
    # first, is tests/test_emul_reduce_band.py's)
    env = dict(os.environ, LD_PRELOAD=MOCK_SO, VIPS_HIP_LIBRARY=EMUL_SO, VIPS_HIP_REDUCE_BAND="0")
    env.update(extra_env or {})
    proc = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          env=env, timeout=1800)
    assert proc.returncode == 0 and "CHILD-OK" in proc.stdout, proc.stdout[-3000:]


def test_reducev8(tmp_path):
    _run(CASES, tmp_path)


def test_reducev8_short_segments(tmp_path):
    _run([c for c in CASES if c[5] == S], tmp_path, {"VIPS_HIP_R16_SEG": "5"})


def test_reducev8_matrix_core_rows_of_tiles(tmp_path):
    """reducev_u8_mfma (reduce_u8.hip) on images of several rows of tiles: every other row is walked bottom-up
    (the flipped problem, taps reversed), and with VIPS_HIP_BAND_NO_ALTERNATE=1 all of them top-down."""
    m = "reducev_u8_mfma"
    cases = [(600, 1300, 3, 8.0, "lanczos3", m), (512, 523, 4, 8.0, "lanczos3", m), (1000, 2000, 1, 8.0, "lanczos3", m),
             (344, 800, 2, 8.0, "lanczos3", m)]
    _run(cases, tmp_path)
    _run(cases[:2], tmp_path, {"VIPS_HIP_BAND_NO_ALTERNATE": "1"})
