"""CPU: the GPU parity files themselves, run against libvipship_emul.so under the mock HIP runtime.

tests/emul compiles the product's kernel FILES for host fibers (kernel_prelude.h: the .hip sources the GPU
runs, not restatements -- beside the kernel bodies written against gcn.h), so a `-m gpu` test whose kernels are in
that set -- by the end of round 4: all of them, the matrix-core reduce included (v_mfma as a meeting of a
wave's fibers) -- can run here, on the CPU, against the same oracle with the same assertions.  What stays
GPU-only: what needs torch CUDA tensors or the libvips plugin's own library, device-sized cases, the slow
sweeps (a v_mfma is 64 fiber switches).  tools/emul_gpu_suite.sh runs everything that can run."""
import os
import re
import sys

import pytest

from tests import helpers
from tests.test_emul_resize_sharpen import EMUL_SO, _build_emul
from tests.test_host_glue_mock import MOCK_SO, _build_mock, _gpu_present

ENABLED = not (_gpu_present() or not helpers.have_ref() or not _build_mock() or not _build_emul())
pytestmark = pytest.mark.skipif(not ENABLED,
                                reason="a real GPU is present, or the reference / mock runtime / emulation cannot be built")

# test -> (files, -k deselections, at least this many cases must pass).  Independent child processes of minutes
# each: conftest.py starts the selected ones when collection ends (prestart), the test waits for its own.
JOBS = {
    "test_conv_colour_file_on_the_cpu": (["tests/test_conv_colour_gpu.py"], [], 589),
    "test_c1_and_the_new_kernels_files_on_the_cpu": (["tests/test_c1_vipsthumbnail.py", "tests/test_convsep_int_gpu.py"], [], 38),
    "test_resample_file_on_the_cpu": (["tests/test_resample_gpu.py"],
                                      ["resize", "thumbnail", "c2_full", "c2_quarter", "mfma_variants", "region_windows", "any_bands"], 160),
    "test_dispatch_fuzz_file_on_the_cpu": (["tests/test_fuzz_dispatch_gpu.py"], [], 5),
}


def prestart(names):
    if not ENABLED:
        return
    env = dict(os.environ, LD_PRELOAD=MOCK_SO, VIPS_HIP_LIBRARY=EMUL_SO)
    for name in names:
        files, deselect, _ = JOBS[name]
        cmd = [sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + files
        if deselect:
            cmd += ["-k", " and ".join("not " + d for d in deselect)]
        helpers.Background.start("emul_gpu_suite:" + name, cmd, env=env, cwd=helpers.ROOT)


def _run(name):
    prestart([name])  # (no-op when conftest already did)
    rc, text = helpers.Background.wait("emul_gpu_suite:" + name)
    tail = text[-3000:]
    m = re.search(r"(\d+) passed", tail)
    assert rc == 0 and m and "failed" not in tail.splitlines()[-1], tail
    assert int(m.group(1)) >= JOBS[name][2], tail


def test_conv_colour_file_on_the_cpu():
    """tests/test_conv_colour_gpu.py: convi / convf / convsep / gaussblur in every format, the fused blur +
    colourspace kernel, every colour route, cast, premultiply, sharpen, the approximate convolutions --
    the thumbnail goldens and BASELINE config 4's per-image pipeline (the one-kernel resize chains)."""
    _run("test_conv_colour_file_on_the_cpu")


def test_c1_and_the_new_kernels_files_on_the_cpu():
    """BASELINE config 1 (the vipsthumbnail command line against the device part of vips_thumbnail_image) and
    the parity file of the integer horizontal pass of convsep_stream."""
    _run("test_c1_and_the_new_kernels_files_on_the_cpu")


def test_resample_file_on_the_cpu():
    """tests/test_resample_gpu.py: reduce / shrink in every format (the general kernels), the goldens, the
    fused RGBA reduce on the matrix instruction and its VALU sibling, upsizing -- without the cases that go
    through the one-kernel resize chains, the device-sized ones and the slow sweeps of the matrix-core
    kernel's variants (a minute each on fibers: every v_mfma is a meeting of 64 fibers)."""
    _run("test_resample_file_on_the_cpu")


def test_dispatch_fuzz_file_on_the_cpu():
    """tests/test_fuzz_dispatch_gpu.py: the seeded sweep over the dispatch guards of the streaming and matrix-core
    kernels (round 5's conv_u8_mfma / reduce_band included: v_mfma_f32_32x32x16_f16 / 16x16x32 as wave meetings)."""
    _run("test_dispatch_fuzz_file_on_the_cpu")
