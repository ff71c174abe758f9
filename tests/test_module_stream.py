"""The OUTPUT side of the libvips module (host/vips_hip_module.c): a
strip-mined result is served to libvips' generate calls as its strips land, from a host ring bounded
by $VIPS_HIP_HOST_BUDGET -- the reference's sinks hold two buffers, not the image
(iofuncs/sinkdisc.c:177-220, sink.c:428-441) -- and a strip that has left the ring is made again
when a consumer comes back for it.

CPU: the module runs end to end against the mock HIP runtime with the packed-byte convolution
kernel on host fibers (tests/emul), so gaussblur_hip makes real pixels here: what is checked is the
producer / cache / generate machinery, bit for bit against the built-in gaussblur under every
access pattern.  GPU: the same on the device, plus the resample family and a large float image with
the process's resident set watched."""
import os
import sys

import numpy as np
import pytest

from tests import helpers
from tests.test_emul_resize_sharpen import EMUL_SO, _build_emul
from tests.test_host_glue_mock import MOCK_SO, _build_mock, _gpu_present

CHILD = r'''
import sys, os, ctypes
sys.path.insert(0, %(root)r)
import numpy as np
from tests import helpers
from tests.helpers import Ref

Ref.load_module()
module = ctypes.CDLL(helpers.MODULE_LIB)

def stats(reset=0):
    a = (ctypes.c_uint64 * 4)()
    module.vips_hip_module_stream_stats(a, reset)
    return list(a)

src = helpers.lcg_image(500, 640, 3, np.uint8, 82)
want = Ref.run("gaussblur", src, "sigma=3", 22)
out_bytes = want.nbytes

# 1. over the HBM budget, under the host budget: every strip made once, everything kept,
#    requests served while strips were still being made
os.environ["VIPS_HIP_BUDGET"] = "200k"
before, s0 = module.vips_hip_module_strips_done(), stats(1)
got = Ref.run("gaussblur_hip", src, "sigma=3", 22)
made, s1 = module.vips_hip_module_strips_done() - before, stats()
assert np.array_equal(got, want)
assert made >= 4, made
assert s1[3] - s0[3] == 1, "one producer run"
assert s1[2] > s0[2], "no request was served before the last strip was made"
assert s1[1] == s0[1], "host memory not given back"
n_strips = made

# 2. the result does not fit the host budget: a ring of strips, peak host memory bounded
os.environ["VIPS_HIP_HOST_BUDGET"] = "150k"
before, s0 = module.vips_hip_module_strips_done(), stats(1)
got = Ref.run("gaussblur_hip", src, "sigma=3", 22)
made, s1 = module.vips_hip_module_strips_done() - before, stats()
assert np.array_equal(got, want)
assert s1[0] - s0[1] <= 150 * 1024, ("host peak over the budget", s1, s0)
assert s1[0] - s0[1] < out_bytes // 4
assert made >= n_strips

# 3. consumers that do not walk top to bottom: a strip that left the ring is made again
for tail in ["flip:direction=vertical", "rot:angle=d90", "extract_area:left=7,top=600,width=50,height=10",
             "shrinkv:vshrink=7"]:
    s0 = stats(1)
    w = Ref.run_chain("gaussblur:sigma=3;" + tail, src, 22)
    g = Ref.run_chain("gaussblur_hip:sigma=3;" + tail, src, 22)
    s1 = stats()
    assert np.array_equal(g, w), tail
    assert s1[0] - s0[1] <= 150 * 1024, (tail, s1, s0)

# 4. one small request at the bottom makes the strips it touches, not the image
before = module.vips_hip_module_strips_done()
Ref.run_chain("gaussblur_hip:sigma=3;extract_area:left=7,top=600,width=50,height=10", src, 22)
assert module.vips_hip_module_strips_done() - before <= 3
# (the producer's walk starts at the strip of the call that started the evaluation, not at strip 0 -- also on an
# image one tile wide, where libvips' workers ask for tiles of different tops at once)
narrow = helpers.lcg_image(100, 600, 3, np.uint8, 83)
for tail in ["extract_area:left=3,top=540,width=40,height=25", "extract_area:left=0,top=300,width=100,height=40"]:
    before = module.vips_hip_module_strips_done()
    g = Ref.run_chain("gaussblur_hip:sigma=2;" + tail, narrow, 22)
    assert np.array_equal(g, Ref.run_chain("gaussblur:sigma=2;" + tail, narrow, 22)), tail
    if tail.startswith("extract_area:left=3"):
        assert module.vips_hip_module_strips_done() - before <= 3, tail

# 5. a partial (pulled) input: the producer's prefetch, and a strip-mined result feeding a
#    following *_hip operation
w = Ref.run_chain("invert;gaussblur:sigma=3;gaussblur:sigma=1.5", src, 22)
g = Ref.run_chain("invert;gaussblur_hip:sigma=3;gaussblur_hip:sigma=1.5", src, 22)
assert np.array_equal(g, w)

# 6. an error inside the producer reaches the caller as a vips error
os.environ["MOCK_HIP_FAIL_D2H_AFTER"] = "2"
try:
    Ref.run("gaussblur_hip", src, "sigma=3", 22)
    raise SystemExit("a failed download went unnoticed")
except RuntimeError as e:
    assert "gaussblur_hip" in str(e), str(e)
del os.environ["MOCK_HIP_FAIL_D2H_AFTER"]
assert np.array_equal(Ref.run("gaussblur_hip", src, "sigma=3", 22), want)
assert stats()[1] == 0, "host memory held after every operation is gone"
print("CHILD-OK")
'''


RING_ENABLED = not (_gpu_present() or not helpers.have_module() or not _build_mock() or not _build_emul())
RING_TEST = "test_strips_stream_through_a_bounded_host_ring"


def prestart(names):
    """A ring run is a child process of half a minute, started by its own test and nothing else (see
    tests/conftest.py: its strip counts need a machine that is not loaded)."""
    if not RING_ENABLED:
        return
    for name in names:
        if not name.startswith(RING_TEST + "["):
            continue
        devices = int(name[len(RING_TEST) + 1:-1])
        env = dict(os.environ, LD_PRELOAD=MOCK_SO + ":" + EMUL_SO)
        if devices > 1:
            env.update(MOCK_HIP_DEVICES=str(devices), VIPS_HIP_DEVICES=",".join(str(d) for d in range(devices)))
        helpers.Background.start("module_stream:" + name, [sys.executable, "-c", CHILD % {"root": helpers.ROOT}], env=env)


@pytest.mark.skipif(not RING_ENABLED,
                    reason="a real GPU is present, or the reference / module / mock runtime / emulation cannot be built")
@pytest.mark.parametrize("devices", [1, 2])
def test_strips_stream_through_a_bounded_host_ring(devices):
    """devices = 2: two fake devices dealt round-robin over the threads that never bind themselves
    ($VIPS_HIP_DEVICES=0,1: libvips' workers AND the strip producers); a producer that is started
    again for an evicted strip goes back to its first run's device (the slot events live there)."""
    name = "%s[%d]" % (RING_TEST, devices)
    prestart([name])
    rc, text = helpers.Background.wait("module_stream:" + name, timeout=1800)
    assert rc == 0 and "CHILD-OK" in text, text[-3000:]


def _stats(module, reset=0):
    import ctypes

    a = (ctypes.c_uint64 * 4)()
    module.vips_hip_module_stream_stats(a, reset)
    return list(a)


def _rss_peak_kb():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmHWM:"):
                return int(line.split()[1])
    return 0


@pytest.mark.gpu
@pytest.mark.skipif(not helpers.have_module(), reason="oracle/_ref or host/_build missing")
class TestStreamOnGpu(object):
    def setup_class(cls):
        from tests.helpers import Ref

        Ref.load_module()

    @pytest.mark.parametrize("hip_op,ref_op,which,args", [
        ("reduce_hip", "reduce", "rgba", "hshrink=8,vshrink=8"),
        ("reduce_hip", "reduce", "rgb", "hshrink=2.5,vshrink=3.3"),
        ("resize_hip", "resize", "rgb", "scale=0.37"),
        ("shrink_hip", "shrink", "rgb", "hshrink=3,vshrink=4"),
        ("gaussblur_hip", "gaussblur", "rgb", "sigma=3"),
        ("sharpen_hip", "sharpen", "rgb", ""),
        ("colourspace_hip", "colourspace", "rgb", "space=lab"),
        ("conv_hip", "conv", "flt", None),
    ])
    def test_ring_equals_whole_equals_builtin(self, hip_op, ref_op, which, args):
        import ctypes

        from tests.golden import cases
        from tests.helpers import Ref

        src = {"rgb": helpers.lcg_image(700, 900, 3, np.uint8, 82), "rgba": helpers.lcg_image(1024, 1203, 4, np.uint8, 81),
               "flt": helpers.lcg_image(300, 500, 2, np.float32, 83)}[which]
        # (an interpretation is set with a header-only vips_copy: the input is then a partial image
        # that the producer pulls through its two staging buffers, which count as host memory and
        # cannot be smaller than the rows a 16-line strip reads -- the resample family is given the
        # memory image itself, so that what is measured here is the ring)
        interp = cases.INTERP["srgb"] if hip_op in ("gaussblur_hip", "sharpen_hip", "colourspace_hip") else 0
        module = ctypes.CDLL(helpers.MODULE_LIB)

        def run(op):
            if args is None:
                mask, scale, offset = cases.MASKS["rand5x7"]
                return Ref.run_mask(op, src, mask, scale, offset, "precision=float")
            return Ref.run(op, src, args, interp)

        want = run(ref_op)
        os.environ["VIPS_HIP_BUDGET"] = "300k"
        host_budget = max(48 * 1024, want.nbytes // 6)
        os.environ["VIPS_HIP_HOST_BUDGET"] = "%d" % host_budget
        before, s0 = module.vips_hip_module_strips_done(), _stats(module, 1)
        try:
            got = run(hip_op)
        finally:
            del os.environ["VIPS_HIP_BUDGET"]
            del os.environ["VIPS_HIP_HOST_BUDGET"]
        s1 = _stats(module)
        assert module.vips_hip_module_strips_done() - before >= 3, "not strip-mined"
        assert got.shape == want.shape and got.dtype == want.dtype
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
        assert s1[0] - s0[1] <= host_budget, ("host peak over the budget", s1, s0, want.nbytes)
        assert s1[1] == s0[1], "host memory not given back"

    def test_random_access_over_a_ring(self):
        from tests.helpers import Ref

        src = helpers.lcg_image(700, 900, 3, np.uint8, 82)
        os.environ["VIPS_HIP_BUDGET"] = "300k"
        os.environ["VIPS_HIP_HOST_BUDGET"] = "200k"
        try:
            for tail in ["flip:direction=vertical", "rot:angle=d90", "shrinkv:vshrink=7"]:
                w = Ref.run_chain("reduce:hshrink=2.5,vshrink=1.5;" + tail, src, 22)
                g = Ref.run_chain("reduce_hip:hshrink=2.5,vshrink=1.5;" + tail, src, 22)
                assert np.array_equal(g, w), tail
        finally:
            del os.environ["VIPS_HIP_BUDGET"]
            del os.environ["VIPS_HIP_HOST_BUDGET"]

    def test_large_float_result_keeps_the_process_small(self):
        """VERDICT round 3 item 6, scaled to the suite's time: reduce_hip behind a cast on a
        12288 x 12288 image -- 604 MB of float input pulled through libvips, 151 MB of float
        output -- under a 128 MB HBM budget and a 48 MB host budget.  The module's own host
        memory (ring + staging) stays under the budget and the process's resident set grows by
        far less than input + output (the final image itself, 151 MB, is the caller's)."""
        import ctypes

        from tests.helpers import Ref

        n = 12288
        src = helpers.lcg_image(n, n, 1, np.uint8, 5)
        module = ctypes.CDLL(helpers.MODULE_LIB)
        os.environ["VIPS_HIP_BUDGET"] = "128m"
        os.environ["VIPS_HIP_HOST_BUDGET"] = "48m"
        s0 = _stats(module, 1)
        rss0 = _rss_peak_kb()
        try:
            got = Ref.run_chain("cast:format=float;reduce_hip:hshrink=2,vshrink=2", src)
        finally:
            del os.environ["VIPS_HIP_BUDGET"]
            del os.environ["VIPS_HIP_HOST_BUDGET"]
        s1 = _stats(module)
        rss1 = _rss_peak_kb()
        assert got.shape[:2] == (n // 2, n // 2)
        assert s1[0] - s0[1] <= 48 << 20, (s1, s0)
        assert s1[2] > s0[2], "no request served while strips were being made"
        # input 604 MB + output 151 MB were never on the host at once
        assert (rss1 - rss0) * 1024 < 420 << 20, (rss0, rss1)
        # spot rows against the built-in (the whole image is compared at small sizes above)
        want = Ref.run_chain("cast:format=float;reduce:hshrink=2,vshrink=2;extract_area:left=0,top=3000,width=%d,height=64" % (n // 2), src)
        assert np.array_equal(got[3000:3064].view(np.uint8), want.view(np.uint8))
