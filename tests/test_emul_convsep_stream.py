"""CPU: libvips_amd/csrc/convsep_stream.hip ITSELF -- both passes of a separable float convolution in one
streaming kernel, the colour epilogue behind it (BASELINE config 3: gaussblur sigma 8 + sRGB -> Lab), the
integer horizontal pass and both forms of the epilogue -- compiled for host fibers
(tests/emul/convsep_stream_emul.cpp: the kernel file, not a restatement; the LDS-DMA, the wave vote and the
handful of builtins get host meanings in tests/emul/kernel_prelude.h) and run under the mock HIP runtime
against the oracle (the plain-C port), bit for bit.  What the fibers cannot show -- the order of memory
operations s_waitcnt counts on -- is what the GPU suite is for."""
import os
import subprocess
import sys

import pytest

from tests import helpers
from tests.test_emul_resize_sharpen import EMUL_SO, _build_emul
from tests.test_host_glue_mock import MOCK_SO, _build_mock, _gpu_present

pytestmark = pytest.mark.skipif(_gpu_present() or not _build_mock() or not _build_emul(),
                                reason="a real GPU is present, or the mock runtime / emulation cannot be built")

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import libvips_amd
from libvips_amd import Image, _ffi
from tests import helpers
from tests.helpers import PortCC

libvips_amd.init(0)
lib = libvips_amd.lib
lib.vips_hip_set_exact_float(1)  # (as tests/conftest.py does for the GPU suite; the default mode is asked for per case)


def bits(a):
    return a.view(np.uint32)


def image(w, h, b, kind, seed):
    if kind == "float":
        return helpers.lcg_image(w, h, b, np.float32, seed)
    src = helpers.lcg_image(w, h, b, np.uint8, seed).astype(np.float32)
    if kind == "almost":
        src[h // 3: h // 3 + 5, w // 4: w // 4 + 9] += 0.25
        src[0, 0, 0] = 300.0
        src[h - 1, w - 1, b - 1] = -7.0
        src[h // 2, w // 2, 0] = -0.0
        src[h // 2, (w // 2 + 3) %% w, 0] = 1e-40
    return src


def ulp_distance(a, b):
    def key(x):
        i = x.view(np.int32).astype(np.int64)
        return np.where(i < 0, -(i & 0x7fffffff), i)
    return int(np.max(np.abs(key(a) - key(b))))


def gated(fn):
    lib.vips_hip_gate_reset()
    lib.vips_hip_gate_enable(1)
    try:
        out = fn()
        report = libvips_amd.gate_report()
    finally:
        lib.vips_hip_gate_enable(0)
        lib.vips_hip_gate_reset()
    assert len(report) == 1 and list(report)[0].startswith("convsep_stream"), report
    return out


for case in %(cases)r:
    what = case[0]
    if what == "colour":
        _, w, h, sigma, space, precision, kind, env = case
        src = image(w, h, 3, kind, 68)
        want = PortCC.colourspace(PortCC.gaussblur(src, sigma, precision=precision), space, "srgb")
        os.environ.update(env)
        try:
            got = gated(lambda: Image.new_from_array(src, interpretation="srgb").gaussblur_colourspace(sigma, space, precision=precision).numpy())
        finally:
            for k in env:
                del os.environ[k]
        assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), case
    elif what == "blur":
        _, w, h, b, sigma, precision, kind, exact, env = case
        src = image(w, h, b, kind, 67)
        lib.vips_hip_set_exact_float(1 if exact else 0)
        want = PortCC.gaussblur(src, sigma, precision=precision)
        os.environ.update(env)
        try:
            got = gated(lambda: Image.new_from_array(src).gaussblur(sigma, precision=precision).numpy())
        finally:
            for k in env:
                del os.environ[k]
            lib.vips_hip_set_exact_float(1)
        if env.get("VIPS_HIP_STREAM_INT") == "2" and kind != "integer":
            assert np.isnan(got).sum() > 0, case  # the poisoning mode shows what the window test refused
        elif exact:
            assert np.array_equal(bits(got), bits(want)), case
        else:
            assert ulp_distance(got, want) <= 1, case  # tolerance: 1 ULP (BASELINE.json north_star)
    elif what == "mask":
        _, mask, scale, offset, precision, special = case
        src = helpers.lcg_image(333, 90, 3, np.float32, 68)
        if special == "nonfinite":
            src[17, 40, 1] = np.inf
            src[60, 300, 0] = np.nan
            src[80, 5, 2] = -np.inf
        elif special == "zeros":
            src[:30] = 0.0
            src[30:60, :150] = -0.0
            src[30:60, 150:] = -1e-40
        m = np.array(mask)
        got = gated(lambda: Image.new_from_array(src).convsep(m, scale=scale, offset=offset, precision=precision).numpy())
        want = PortCC.convsep(src, m, scale, offset, precision)
        if special == "nonfinite":
            assert np.array_equal(got, want, equal_nan=True), case
        else:
            assert np.array_equal(bits(got), bits(want)), case
print("CHILD-OK")
'''

NOENV = {}
INT0, INT1, INT2 = ({"VIPS_HIP_STREAM_INT": v} for v in "012")
EPI0 = {"VIPS_HIP_STREAM_EPI": "0"}
M7 = [[1.0, 2.0, 5.0, 7.0, 5.0, 2.0, 1.0]]
MNEG = [[-1.0, 2.0, 5.0, 7.0, 5.0, 3.0, -2.0]]

COLOUR = [
    # (w, h, sigma, space, precision, pixels, environment): several strips, several row segments, narrow and
    # tiny images, 3 .. 29 taps, both precisions, the generic and the spelled-out epilogue
    ("colour", 700, 300, 8.0, "lab", "integer", "float", NOENV), ("colour", 700, 300, 8.0, "lab", "float", "float", NOENV),
    ("colour", 1030, 77, 2.0, "xyz", "integer", "float", NOENV), ("colour", 37, 211, 0.6, "scrgb", "float", "float", NOENV),
    ("colour", 256, 64, 3.1, "lab", "integer", "float", NOENV), ("colour", 5, 3, 8.0, "lab", "integer", "float", NOENV),
    # BASELINE config 3's input: integers; the integer pass off / on / poisoning, both forms of the epilogue
    ("colour", 700, 300, 8.0, "lab", "integer", "integer", INT0), ("colour", 700, 300, 8.0, "lab", "integer", "integer", INT1),
    ("colour", 700, 300, 8.0, "lab", "integer", "integer", INT2), ("colour", 900, 130, 8.0, "lab", "integer", "integer", EPI0),
    ("colour", 37, 211, 8.0, "lab", "integer", "integer", INT2), ("colour", 1030, 77, 3.1, "lab", "integer", "integer", INT2),
    ("colour", 300, 100, 2.0, "xyz", "integer", "integer", INT2),
    # integers almost everywhere: both passes in one image
    ("colour", 800, 200, 8.0, "lab", "integer", "almost", NOENV), ("colour", 800, 200, 8.0, "lab", "integer", "almost", EPI0),
]
BLUR = [
    # (w, h, bands, sigma, precision, pixels, exact float mode, environment)
    ("blur", 700, 200, 3, 8.0, "integer", "float", True, NOENV), ("blur", 700, 200, 3, 8.0, "float", "float", True, NOENV),
    ("blur", 1500, 60, 1, 2.0, "integer", "float", True, NOENV), ("blur", 37, 211, 4, 0.6, "float", "float", True, NOENV),
    ("blur", 1200, 70, 2, 2.0, "float", "float", True, NOENV), ("blur", 5, 3, 3, 8.0, "integer", "float", True, NOENV),
    ("blur", 700, 200, 3, 8.0, "integer", "float", False, NOENV), ("blur", 700, 200, 3, 2.0, "float", "float", False, NOENV),
    ("blur", 700, 200, 3, 8.0, "integer", "integer", True, INT2), ("blur", 1500, 60, 1, 2.0, "integer", "integer", True, INT2),
    ("blur", 260, 120, 4, 0.6, "integer", "integer", True, INT2), ("blur", 700, 200, 3, 8.0, "integer", "almost", True, NOENV),
    # the poisoning mode must show what the window test refuses (without an epilogue: a NaN would index its
    # table, and (int) NaN is the device's 0 only on the device)
    ("blur", 700, 200, 3, 8.0, "integer", "almost", True, INT2), ("blur", 300, 100, 3, 2.0, "integer", "float", True, INT2),
]
MASKS = [
    ("mask", MNEG, 19.0, 3.0, "integer", "nonfinite"), ("mask", MNEG, 18.5, -0.75, "float", "nonfinite"),
    ("mask", M7, 23.0, 0.0, "integer", "zeros"), ("mask", M7, -23.0, -0.0, "integer", "zeros"), ("mask", M7, 1.0, 2.5, "integer", "zeros"),
]


def _run(cases, tmp_path):
    script = os.path.join(str(tmp_path), "child.py")
    with open(script, "w") as f:
        f.write(CHILD % {"root": helpers.ROOT, "cases": cases})
    env = dict(os.environ, LD_PRELOAD=MOCK_SO, VIPS_HIP_LIBRARY=EMUL_SO)
    proc = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          env=env, timeout=1800)
    assert proc.returncode == 0 and "CHILD-OK" in proc.stdout, proc.stdout[-3000:]


def test_blur_and_colourspace_in_one_kernel(tmp_path):
    _run(COLOUR, tmp_path)


def test_blur_alone(tmp_path):
    _run(BLUR, tmp_path)


def test_masks_offsets_nonfinite_and_signed_zeros(tmp_path):
    _run(MASKS, tmp_path)
