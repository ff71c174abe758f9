"""The one-process, several-devices paths of the C library on the GPU box.  With ONE visible device
(the round's boxes) every logical slot is device 0 (VIPS_HIP_DEVICES=0,0 / devices [0, 0, 0]): the
worker threads, the per-device pools and plan caches, the batch scatter, the strip windows and the
peer copies all run for real and the pixels must be the single-device pixels.  With SEVERAL visible
devices (the first multi-GPU lease) the slots are dealt over all of them -- slots() below -- so the
same tests then move real pixels between devices with hipMemcpyPeerAsync, with no new code.  (The
host logic against distinct fake devices is tests/test_host_glue_mock.py.)"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


def slots(n):
    """n logical slots over the visible devices: 0, 1, ... round-robin (all 0 on a one-GPU box)."""
    import torch

    count = max(torch.cuda.device_count(), 1)
    return [k % count for k in range(n)]


def test_conv_strips_on_one_device_match_whole_image():
    """vips_hip_conv_strips: three strips (all on device 0), halos by hipMemcpyPeerAsync into the
    persistent windows, one host thread per strip; bit for bit the whole-image conv.  Twice: the
    windows persist."""
    import libvips_amd
    from libvips_amd import Image, _ffi
    from libvips_amd._ffi import Region, lib

    libvips_amd.init(0)
    height, width, halo = 1031, 2100, 15
    full = helpers.lcg_image(width, height, 1, np.uint16, 92)
    mask, scale = libvips_amd.gaussmat(5, 0.01, False, "float")
    whole = Image.new_from_array(full).conv(mask, scale=scale, precision="float").numpy()
    n = 3
    devices = (ctypes.c_int * n)(*slots(n))
    strips = _ffi.check_handle(lib.vips_hip_strips_new(width, height, 1, 2, n, devices, halo))
    try:
        for k in range(n):
            own = Region()
            assert lib.vips_hip_strips_region(strips, k, None, ctypes.byref(own), None) == 0
            rows = np.ascontiguousarray(full[own.top:own.top + own.height])
            _ffi.check(lib.vips_hip_memcpy_h2d(own.data, rows.ctypes.data, rows.nbytes))
        m = np.ascontiguousarray(mask, dtype=np.float64)
        for step in range(2):
            outs = (ctypes.c_void_p * n)()
            r = lib.vips_hip_conv_strips(strips, outs, m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), m.shape[1],
                                         m.shape[0], scale, 0.0, 1)
            assert r == 0, _ffi.error_buffer()
            row = 0
            for k in range(n):
                got = Image(outs[k]).numpy()
                assert np.array_equal(got.view(np.uint8), whole[row:row + got.shape[0]].view(np.uint8)), (step, k)
                row += got.shape[0]
            assert row == height
    finally:
        lib.vips_hip_strips_free(strips)


CHILD = r'''
import sys, threading
sys.path.insert(0, %(root)r)
import numpy as np
import libvips_amd
from libvips_amd import Image
from libvips_amd._ffi import lib
from tests import helpers
from tests.helpers import Port

# worker threads that never bind themselves: dealt over VIPS_HIP_DEVICES = 0,0
src = helpers.lcg_image(1203, 917, 4, np.uint8, 46)
want = Port.reduce(src, 8, 8, "lanczos3")
results = {}
def worker(i):
    results[i] = (Image.new_from_array(src).reduce(8, 8, kernel="lanczos3").numpy(), lib.vips_hip_current_device())
ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
for t in ts: t.start()
for t in ts: t.join()
want_devs = %(devs)r
for i in range(4):
    assert results[i][1] in want_devs and np.array_equal(results[i][0], want), i
assert sorted(set(r[1] for r in results.values())) == sorted(set(want_devs))   # every slot took work
devs = (libvips_amd._ffi.c_int * 4)()
assert lib.vips_hip_devices(devs, 4) == 2 and list(devs[:2]) == want_devs
# the batch entry point over the configured slots
ims = [Image.new_from_array(helpers.lcg_image(1024, 768, 3, np.uint8, 60 + k), interpretation="srgb") for k in range(5)]
outs = libvips_amd.resize_sharpen_batch(ims, 0.125, threads=2)
for k, o in enumerate(outs):
    one = ims[k].resize(0.125).sharpen().numpy()
    assert np.array_equal(o.numpy(), one), k
print("CHILD-OK")
'''


def test_worker_threads_dealt_over_vips_hip_devices():
    """VIPS_HIP_DEVICES=<two slots>: threads that never call vips_hip_init() are bound round-robin to the
    listed slots, each with its own stream; identical pixels from every thread."""
    two = slots(2)
    env = dict(os.environ, VIPS_HIP_DEVICES=",".join(str(d) for d in two))
    proc = subprocess.run([sys.executable, "-c", CHILD % {"root": helpers.ROOT, "devs": two}], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    assert proc.returncode == 0 and "CHILD-OK" in proc.stdout, proc.stdout[-3000:]


MODULE_CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from tests import helpers
from tests.helpers import Ref

Ref.load_module()
src = helpers.lcg_image(2048, 1536, 3, np.uint8, 65)
for hip, ref, args in (("resize_hip", "resize", "scale=0.125"), ("gaussblur_hip", "gaussblur", "sigma=2"),
                       ("reduce_hip", "reduce", "hshrink=4,vshrink=4")):
    got = Ref.run(hip, src, args, 22)
    want = Ref.run(ref, src, args, 22)
    assert got.shape == want.shape and np.array_equal(got, want), hip
got = Ref.run_chain("resize_hip:scale=0.25;sharpen_hip:", src, 22)
want = Ref.run_chain("resize:scale=0.25;sharpen:", src, 22)
assert np.array_equal(got, want)
print("CHILD-OK")
'''


@pytest.mark.skipif(not helpers.have_module(), reason="oracle/_ref or host/_build missing")
def test_module_with_vips_hip_devices():
    """The libvips module in a process whose worker pool is spread over VIPS_HIP_DEVICES=0,0:
    libvips' own threads evaluate and generate, whichever slot they were dealt; the pixels are the
    built-in operations' (VERDICT round 2, item 5)."""
    env = dict(os.environ, VIPS_HIP_DEVICES=",".join(str(d) for d in slots(2)))
    proc = subprocess.run([sys.executable, "-c", MODULE_CHILD % {"root": helpers.ROOT}], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    assert proc.returncode == 0 and "CHILD-OK" in proc.stdout, proc.stdout[-3000:]
