"""CPU: the packed-byte vips_reduceh on uchar (libvips_amd/csrc/reduceh_u8_body.h: planar signed
bytes staged in LDS, a lane owns 4 output pixels, 16-bit coefficients as two v_dot4_i32_i8) run
thread by thread on host fibers (tests/emul) under the mock HIP runtime and compared, whole image,
bit for bit, with the compiled reference.  See tests/test_emul_resize_sharpen.py for how the
emulation is built."""
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
    src[: h // 3, : w // 2] = 255          # clipping at both ends (negative lobes beside flat areas)
    src[h // 3: h // 2, w // 2:] = 0
    im = Image.new_from_array(src)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    got = im.reduceh(shrink, kernel=kernel).numpy()
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    want = helpers.Ref.run_chain("reduceh:hshrink=%%r,kernel=%%s" %% (shrink, kernel), src)
    assert list(report) == [gate], (w, h, bands, shrink, kernel, report)
    if gate != "reduceh_u8_packed":
        continue  # (the older kernels are not emulated: under the mock runtime they make no pixels)
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (w, h, bands, shrink, kernel, len(bad), bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])
print("CHILD-OK")
'''

P, L = "reduceh_u8_packed", "reduceh_u8_lds"
# (width, height, bands, shrink, kernel, the kernel that must have run)
CASES = [
    # the shrinks with first taps 8 and 4 pixels apart, every interpolation kernel's tap count
    (4096, 21, 3, 8.0, "lanczos3", P), (4096, 9, 3, 8.0, "cubic", P), (4096, 9, 3, 8.0, "linear", P), (4096, 9, 3, 8.0, "lanczos2", P),
    (2048, 13, 3, 4.0, "lanczos3", P), (2048, 9, 3, 4.0, "mitchell", P), (2048, 9, 3, 4.0, "linear", P),
    # 1, 2 and 4 bands; several blocks across; output widths with a partial last quad; one row
    (8192, 6, 4, 8.0, "lanczos3", P), (1032, 7, 4, 4.0, "lanczos3", P), (1000, 5, 4, 8.0, "cubic", P), (4128, 5, 1, 8.0, "lanczos3", P),
    (808, 9, 2, 4.0, "lanczos3", P), (16384, 1, 3, 8.0, "lanczos3", P), (40, 3, 4, 8.0, "lanczos3", P),
    # not this kernel's case: a fractional shrink (phases differ), a shrink of 2, rows that are not whole dwords
    (4096, 5, 3, 7.3, "lanczos3", L), (2048, 5, 3, 2.0, "lanczos3", L), (4095, 5, 3, 8.0, "lanczos3", "reduceh_general"),
]


def test_reduceh_u8_packed(tmp_path):
    script = os.path.join(str(tmp_path), "child.py")
    with open(script, "w") as f:
        f.write(CHILD % {"root": helpers.ROOT, "cases": CASES})
    # (the fall-backs of THIS kernel: the banded matrix-core one, tests/test_emul_reduce_band.py, is off, and so is
    # round 6's matrix-core kernel for three bands by 8, tests/test_reduceh_u8_gpu.py)
    env = dict(os.environ, LD_PRELOAD=MOCK_SO, VIPS_HIP_LIBRARY=EMUL_SO, VIPS_HIP_REDUCE_BAND="0", VIPS_HIP_NO_REDUCEH3="1")
    proc = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          env=env, timeout=1800)
    assert proc.returncode == 0 and "CHILD-OK" in proc.stdout, proc.stdout[-3000:]
