"""CPU: the fused resize + sharpen kernel (BASELINE config 4's one kernel, libvips_amd/csrc/
resize_sharpen_body.h) run THREAD BY THREAD on host fibers and compared with the compiled reference.

tests/emul builds libvipship_emul.so: the product's objects, with resize_sharpen.hip replaced by
tests/emul/resize_sharpen_emul.cpp -- the same kernel body and the same host code, the launch
replaced by a fiber run of every workgroup (tests/emul/gcn.h restates the few gfx950 instructions
the body is written in).  Under the mock HIP runtime (device memory = host memory) the library's
batch entry point then makes real thumbnails, and what is checked here is everything about the
kernel that is not the GPU itself: its indexing (strips, segments, halos, rings, the output stage),
its barriers (the fibers of a block are resumed in a different order after every barrier) and its
arithmetic shortcuts (tables, the float quotient), bit for bit against vips_resize + vips_sharpen of
the reference.  The same cases run on the device in tests/test_resample_gpu.py."""
import os
import subprocess
import sys

import pytest

from tests import helpers
from tests.test_host_glue_mock import MOCK_SO, _build_mock, _gpu_present

EMUL_DIR = os.path.join(helpers.ROOT, "tests", "emul")
EMUL_SO = os.path.join(EMUL_DIR, "_build", "libvipship_emul.so")


def _build_emul():
    if not os.path.isdir(os.path.join(helpers.ROOT, "libvips_amd", "csrc", "_obj")):
        return False
    # (a cold build is 26 translation units of whole kernel files: minutes one after another)
    proc = subprocess.run(["make", "-j", str(os.cpu_count() or 4), "-C", EMUL_DIR], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True)
    return proc.returncode == 0 and os.path.exists(EMUL_SO)


pytestmark = pytest.mark.skipif(_gpu_present() or not helpers.have_ref() or not _build_mock() or not _build_emul(),
                                reason="a real GPU is present, or the reference / mock runtime / emulation cannot be built")

CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import libvips_amd
from libvips_amd import Image, resize_sharpen_batch
from tests import helpers

libvips_amd.init(0)
interp = helpers.INTERP["srgb"]
for (w, h, n, scale, sigma, kw) in %(cases)r:
    srcs = [helpers.lcg_image(w, h, 3, np.uint8, 100 + i) for i in range(n)]
    if kw.get("flat"):
        # large flat areas (the LUT's dead zone, dark pixels: Lab2XYZ's linear arms) beside noise
        for s in srcs:
            s[: h // 2, : w // 2] = (s[: h // 2, : w // 2] // 32).astype(np.uint8)
            s[h // 2:, w // 2:] = 250
    ims = [Image.new_from_array(s, interpretation="srgb") for s in srcs]
    extra = {k: v for k, v in kw.items() if k != "flat"}
    outs = resize_sharpen_batch(ims, scale, sigma=sigma, **extra)
    chain = "resize:scale=%%r;sharpen:sigma=%%r" %% (scale, sigma) + "".join(",%%s=%%r" %% kv for kv in sorted(extra.items()))
    for s, o in zip(srcs, outs):
        want = helpers.Ref.run_chain(chain, s, interp)
        got = o.numpy()
        assert got.shape == want.shape, (got.shape, want.shape)
        bad = np.argwhere(got != want)
        assert len(bad) == 0, (w, h, scale, sigma, kw, len(bad), bad[:4])
print("CHILD-OK")
'''


def _run(cases, tmp_path, extra_env=None):
    script = os.path.join(str(tmp_path), "child.py")
    with open(script, "w") as f:
        f.write(CHILD % {"root": helpers.ROOT, "cases": cases})
    env = dict(os.environ, LD_PRELOAD=MOCK_SO, VIPS_HIP_LIBRARY=EMUL_SO, VIPS_HIP_RESIZE_SHARPEN="1")
    env.update(extra_env or {})
    proc = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          env=env, timeout=1200)
    assert proc.returncode == 0 and "CHILD-OK" in proc.stdout, proc.stdout[-3000:]


def test_geometries(tmp_path):
    # (width, height, images, scale, sigma, sharpen arguments): one and several strips and segments,
    # heights that are no multiple of a slab, 3-tap and 5-tap blur, 1/8 and 1/16, a batch
    _run([(704, 512, 2, 0.125, 0.5, {}),
          (1408, 776, 1, 0.125, 0.5, {}),
          (2048, 1000, 2, 0.125, 1.0, {}),
          (2112, 640, 1, 0.0625, 0.5, {}),
          (4096, 256, 1, 0.125, 0.5, {}),
          (1000 * 4, 8 * 23, 1, 0.125, 0.7, {})], tmp_path)


def test_flat_and_dark_areas_and_lut_arguments(tmp_path):
    _run([(1408, 512, 1, 0.125, 0.5, {"flat": True}),
          (1408, 512, 1, 0.125, 0.5, {"flat": True, "m1": 1.0, "m2": 2.0, "x1": 1.0, "y2": 4.0, "y3": 6.0}),
          (704, 320, 1, 0.125, 1.0, {"m2": 5.0, "y2": 30.0, "y3": 40.0})], tmp_path)


def test_segment_and_strip_choices(tmp_path):
    # short segments (the halo rows and the drain of the sharpen stages at every segment end),
    # narrow strips, a stage that bursts every batch
    _run([(1408, 776, 1, 0.125, 0.5, {}), (1408, 776, 1, 0.125, 1.0, {})], tmp_path,
         {"VIPS_HIP_STREAM_SEG": "7", "VIPS_HIP_RSH_TW": "9", "VIPS_HIP_STREAM_BURST": "2"})
    _run([(1408, 776, 2, 0.125, 0.5, {})], tmp_path, {"VIPS_HIP_STREAM_SEG": "49", "VIPS_HIP_RSH_TW": "61"})
