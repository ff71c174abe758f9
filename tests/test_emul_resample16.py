"""CPU: the ushort streaming resample kernels (libvips_amd/csrc/resample16_body.h: reducev from a
host-made schedule on v_dot2_i32_i16, shrinkv, reduceh through padded LDS, shrinkh) run thread by
thread on host fibers (tests/emul) under the mock HIP runtime and compared, whole image, bit for
bit, with the compiled reference.  See tests/test_emul_resize_sharpen.py."""
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
for (op, w, h, bands, args, gates) in %(cases)r:
    src = helpers.lcg_image(w, h, bands, np.uint16, 11 + w)
    src[: h // 3, : w // 2] = 65535          # saturating sums
    src[h // 3: h // 2, w // 2:] = 0
    im = Image.new_from_array(src)
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    got = getattr(im, op)(*args).numpy()
    report = libvips_amd.gate_report()
    lib.vips_hip_gate_enable(0)
    names = {"reducev": "vshrink", "reduceh": "hshrink", "shrinkv": "vshrink", "shrinkh": "hshrink"}
    if op in ("reduce", "shrink"):
        chain = "%%s:hshrink=%%r,vshrink=%%r" %% (op, args[0], args[1])
    else:
        chain = "%%s:%%s=%%r" %% (op, names[op], args[0])
    if op.startswith("reduce") and len(args) > (2 if op == "reduce" else 1):
        chain += ",kernel=" + args[-1]
    want = helpers.Ref.run_chain(chain, src)
    assert sorted(report) == sorted(gates), (op, w, h, bands, args, report)
    if any("general" in g for g in gates):
        continue  # (the older kernels are not emulated: under the mock runtime they make no pixels)
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (op, w, h, bands, args, len(bad), bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])
print("CHILD-OK")
'''


def _run(cases, tmp_path, extra_env=None):
    script = os.path.join(str(tmp_path), "child.py")
    with open(script, "w") as f:
        f.write(CHILD % {"root": helpers.ROOT, "cases": cases})
    # (the vector-ALU kernels: the matrix-core ones, tests/test_emul_reduce_band.py, take ushort reduces first)
    env = dict(os.environ, LD_PRELOAD=MOCK_SO, VIPS_HIP_LIBRARY=EMUL_SO, VIPS_HIP_REDUCE_BAND="0", VIPS_HIP_NO_SHRINKBOX16="1")  # (the pair of box kernels; the one-kernel form: test_shrinkbox16)
    env.update(extra_env or {})
    proc = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          env=env, timeout=1800)
    assert proc.returncode == 0 and "CHILD-OK" in proc.stdout, proc.stdout[-3000:]


V, H = ["reducev_u16_stream"], ["reduceh_u16_lds"]
CASES = [
    ("reducev", 1024, 700, 4, (8.0,), V), ("reducev", 517, 333, 3, (7.3,), ["reducev_general"]),  # (3-band rows: not 8-byte groups)
    ("reducev", 512, 333, 3, (7.3,), V), ("reducev", 300, 200, 4, (2.0, "cubic"), V), ("reducev", 301, 260, 1, (3.7, "linear"), ["reducev_general"]),
    ("reducev", 304, 260, 1, (3.7, "linear"), V), ("reducev", 2100, 90, 4, (1.6,), V),
    ("reduceh", 1024, 70, 4, (8.0,), H), ("reduceh", 1031, 33, 3, (7.3,), ["reduceh_general"]), ("reduceh", 1032, 33, 3, (7.3,), H),
    ("reduceh", 700, 41, 1, (2.5, "mitchell"), H), ("reduceh", 402, 19, 2, (3.0,), H),
    ("shrinkv", 1024, 700, 4, (4,), ["shrinkv_u16_stream"]), ("shrinkv", 512, 333, 3, (5,), ["shrinkv_u16_stream"]),
    ("shrinkv", 2100, 37, 2, (2,), ["shrinkv_u16_stream"]),
    ("shrinkh", 1024, 70, 4, (4,), ["shrinkh_u16_stream"]), ("shrinkh", 1032, 33, 3, (5,), ["shrinkh_u16_stream"]), ("shrinkh", 1031, 33, 3, (5,), ["shrinkh_general"]),
    ("shrinkh", 700, 41, 1, (3,), ["shrinkh_u16_stream"]),
    ("reduce", 1024, 512, 4, (8.0, 8.0), V + H), ("shrink", 1024, 512, 4, (4.0, 4.0), ["shrinkv_u16_stream", "shrinkh_u16_stream"]),
]


def test_ushort_streaming(tmp_path):
    _run(CASES, tmp_path)


def test_ushort_streaming_short_segments(tmp_path):
    _run([c for c in CASES if c[0] in ("reducev", "shrinkv", "reduce") and "general" not in c[5][0]], tmp_path,
         {"VIPS_HIP_R16_SEG": "5"})
