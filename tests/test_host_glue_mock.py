"""CPU: the HOST side of libvipship.so driven end to end against a mock HIP runtime
(tests/mock_hip/mock_hip.cpp: memory is host memory, copies are memcpy, kernels do nothing).

What this can show without a GPU: every operation's plan building, dispatch, output sizes and
formats, the file loaders and savers (pixel-exact: they are copies), extract_area (a copy), the
batch thread pool (and that its streams are given back), the libvips module's build / generate
plumbing -- and that none of it crashes.  What it cannot show is any pixel a kernel makes; that is
what the `-m gpu` tests are for.  The child process is LD_PRELOADed with the mock; nothing here
runs when a real GPU is present."""
import os
import subprocess
import sys

import pytest

from tests import helpers

MOCK_SRC = os.path.join(helpers.ROOT, "tests", "mock_hip", "mock_hip.cpp")
MOCK_SO = os.path.join(helpers.ROOT, "tests", "mock_hip", "_build", "libmockhip.so")


def _build_mock():
    if os.path.exists(MOCK_SO) and os.path.getmtime(MOCK_SO) >= os.path.getmtime(MOCK_SRC):
        return True
    if not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        return False
    os.makedirs(os.path.dirname(MOCK_SO), exist_ok=True)
    # (several pytest-xdist workers may get here at once: build aside, then rename into place)
    tmp = "%s.%d.tmp" % (MOCK_SO, os.getpid())
    proc = subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", "-o", tmp, MOCK_SRC],
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        return False
    os.replace(tmp, MOCK_SO)
    return True


def _gpu_present():
    import torch

    return torch.cuda.is_available()


pytestmark = pytest.mark.skipif(_gpu_present() or not _build_mock(),
                                reason="a real GPU is present, or the mock runtime cannot be built")


def run_child(body, tmp_path, extra_env=None):
    script = os.path.join(str(tmp_path), "child.py")
    with open(script, "w") as f:
        f.write("import sys\nsys.path.insert(0, %r)\n" % helpers.ROOT)
        f.write("TMP = %r\n" % str(tmp_path))
        f.write(body)
        f.write("\nprint('CHILD-OK', flush=True)\n")
    env = dict(os.environ, LD_PRELOAD=MOCK_SO)
    env.update(extra_env or {})
    proc = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          env=env, timeout=600)
    assert proc.returncode == 0 and "CHILD-OK" in proc.stdout, proc.stdout[-3000:]
    return proc.stdout


def test_files_and_copies_are_pixel_exact(tmp_path):
    run_child(r'''
import os, ctypes
import numpy as np
import libvips_amd
from libvips_amd import Image
from tests import helpers

libvips_amd.init(0)
# .v: file -> "HBM" -> file, byte for byte, every format family, several staging chunks
for dtype in (np.uint8, np.uint16, np.float32, np.complex64):
    if dtype == np.complex64:
        re = helpers.lcg_image(50, 40, 2, np.float32, 94)
        src = (re + 1j * re[::-1]).astype(np.complex64)
    else:
        src = helpers.lcg_image(123, 77, 3, dtype, 94)
    a, b = os.path.join(TMP, "a.v"), os.path.join(TMP, "b.v")
    helpers.write_v(a, src, interpretation=22)
    im = Image.new_from_file(a)
    assert np.array_equal(im.numpy(), src)
    im.write_to_file(b)
    assert open(a, "rb").read() == open(b, "rb").read()
big = helpers.lcg_image(4096, 4400, 4, np.uint8, 95)  # 68.75 MiB: three 32 MiB chunks
a, b = os.path.join(TMP, "big.v"), os.path.join(TMP, "big_out.v")
helpers.write_v(a, big, interpretation=22)
im = Image.new_from_file(a)
im.write_to_file(b)
assert open(a, "rb").read() == open(b, "rb").read()
# extract_area is a copy
src = helpers.lcg_image(300, 200, 4, np.uint16, 96)
im = Image.new_from_array(src)
assert np.array_equal(im.extract_area(10, 20, 250, 100).numpy(), src[20:120, 10:260])
try:
    im.extract_area(100, 0, 250, 10)
    raise SystemExit("bad extract area accepted")
except libvips_amd.VipsHipError as e:
    assert "bad extract area" in str(e)
# JPEG: host decode + upload is pixel exact against the host decoder itself
import tests.test_zz_jpeg as J
p = os.path.join(TMP, "t.jpg")
J.make_jpeg(p, 641, 487)
for shrink in (1, 2, 4, 8):
    want, _ = J.product_decode(p, shrink)
    assert np.array_equal(Image.new_from_jpeg(p, shrink).numpy(), want)
''', tmp_path)


def test_sizes_formats_and_refusals(tmp_path):
    run_child(r'''
import os
import numpy as np
import libvips_amd
from libvips_amd import Image
from tests import helpers
from tests.helpers import PortCC, Port
import tests.test_zz_jpeg as J

libvips_amd.init(0)
src = helpers.lcg_image(517, 389, 4, np.uint8, 99)
im = Image.new_from_array(src, interpretation="srgb")
# thumbnail geometry (fit, fill + crop, force, down, linear) against the oracle port's
for crop in ("none", "centre", "low", "high", "all"):
    for (tw, th, size, linear) in ((100, 100, "both", False), (60, 200, "both", False), (400, 50, "down", False),
                                   (90, 90, "both", True), (300, 100, "force", False), (700, 700, "both", False)):
        want = PortCC.thumbnail_image(src, "srgb", tw, th, size=size, linear=linear, crop=crop)
        got = im.thumbnail_image(tw, th, size=size, linear=linear, crop=crop)
        assert (got.height, got.width, got.bands) == want.shape, (crop, tw, th, size, linear)
        assert got.numpy().dtype == want.dtype
# JPEG thumbnails: geometry against the reference CLI's, grey included; refusals
class T:  # tmp_path stand-in for the helper
    def __init__(self, p): self.p = p
    def __str__(self): return self.p
p = os.path.join(TMP, "t.jpg")
for (w, h, grey, size) in ((2000, 1500, False, "200x200"), (1801, 1203, False, "100x100"), (1000, 750, True, "150x150")):
    J.make_jpeg(p, w, h, grey)
    tw, th = [int(v) for v in size.split("x")]
    got = Image.thumbnail(p, tw, th)
    if os.path.exists(J.VIPSTHUMBNAIL) and J._ref_has_jpeg():
        want, _ = J.cli_thumbnail(T(TMP), p, size)
        assert (got.height, got.width, got.bands) == want.shape, (w, h, size)
exif = J.PIL.Exif()
exif[0x0112] = 6
J.make_jpeg(p, 300, 200, exif=exif.tobytes())
for fn, needle in ((lambda: Image.thumbnail(p, 64), "auto-rotation"),):
    try:
        fn()
        raise SystemExit("accepted")
    except libvips_amd.VipsHipError as e:
        assert needle in str(e)
J.make_jpeg(p, 300, 200, icc_profile=b"\0" * 200)
try:
    Image.thumbnail(p, 64, linear=True)
    raise SystemExit("accepted")
except libvips_amd.VipsHipError as e:
    assert "ICC" in str(e)
assert Image.thumbnail(p, 64).width == 64
try:
    im.thumbnail_image(64, crop="attention")
    raise SystemExit("accepted")
except libvips_amd.VipsHipError as e:
    assert "attention" in str(e)
# every precision of the convolution front doors builds its plan and dispatches, every format
for dtype in (np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32, np.float64):
    a = helpers.lcg_image(90, 70, 2, dtype, 1)
    ia = Image.new_from_array(a)
    for prec in ("integer", "float", "approximate"):
        o = ia.gaussblur(3.0, precision=prec)
        assert (o.width, o.height, o.bands) == (90, 70, 2)
        want = PortCC.gaussblur(a, 3.0, precision=prec)
        assert o.numpy().dtype == want.dtype, (dtype, prec)
    m, s = PortCC.gaussmat(2.0, 0.2, False, "integer")
    assert ia.conv(m, scale=s, precision="approximate", layers=7, cluster=2).numpy().shape == a.shape
libvips_amd.vector_set_enabled(True)
assert Image.new_from_array(helpers.lcg_image(64, 48, 3)).gaussblur(2.0).numpy().shape == (48, 64, 3)
libvips_amd.vector_set_enabled(False)
# resize / reduce / shrink geometry against the port
b = helpers.lcg_image(333, 251, 3, np.uint8, 2)
ib = Image.new_from_array(b)
for scale in (0.5, 0.123, 0.77, 1.9, 3.0):
    assert ib.resize(scale).numpy().shape == Port.resize(b, scale).shape, scale
assert ib.reduce(2.5, 3.3).numpy().shape == Port.reduce(b, 2.5, 3.3).shape
assert ib.shrink(3, 4).numpy().shape == Port.shrink(b, 3, 4).shape
''', tmp_path)


def test_batch_threads_give_their_streams_back(tmp_path):
    run_child(r'''
import ctypes, os
import numpy as np
import libvips_amd
from libvips_amd import Image
import tests.test_zz_jpeg as J

mock = ctypes.CDLL(os.environ["LD_PRELOAD"])
mock.mock_hip_launches.restype = ctypes.c_long
libvips_amd.init(0)
paths = []
for i in range(9):
    p = os.path.join(TMP, "b%d.jpg" % i)
    J.make_jpeg(p, 400 + 37 * i, 300 + 11 * i, grey=(i == 3))
    paths.append(p)
paths.insert(4, os.path.join(TMP, "missing.jpg"))
Image.thumbnail(paths[0], 64)  # the main thread's own stream exists from here on
before = mock.mock_hip_live_streams()
for round in range(3):
    outs = Image.thumbnail_batch(paths, 96, 96, crop="centre", threads=4)
    assert len(outs) == len(paths)
    for p, o in zip(paths, outs):
        if p.endswith("missing.jpg"):
            assert isinstance(o, libvips_amd.VipsHipError) and "unable to open" in str(o)
        else:
            assert (o.width, o.height) == (96, 96), p
assert mock.mock_hip_live_streams() == before, (before, mock.mock_hip_live_streams())
assert mock.mock_hip_launches() > 0
''', tmp_path)


def test_uniform_batch_takes_the_batch_launches(tmp_path):
    """vips_hip_resize_sharpen_batch on a uniform batch of more than one launch's worth of images:
    one resize launch and one sharpen launch per 64 images (on the two CU-masked streams the
    library keeps per device, or with $VIPS_HIP_BATCH_OVERLAP=0 on the caller's); with
    $VIPS_HIP_RESIZE_SHARPEN=1 ONE launch per 64 images (resize + sharpen in one kernel, on the
    caller's stream); thumbnails of the right geometry; a mixed batch goes image by image."""
    run_child(r'''
import ctypes, os
import numpy as np
import libvips_amd
from libvips_amd import Image

mock = ctypes.CDLL(os.environ["LD_PRELOAD"])
mock.mock_hip_launches.restype = ctypes.c_long
libvips_amd.init(0)
ims = [Image.new_from_array(np.zeros((64, 688, 3), np.uint8), interpretation="srgb") for _ in range(70)]
ims[0].resize(0.125)
s0 = mock.mock_hip_live_streams()
n0 = mock.mock_hip_launches()
os.environ["VIPS_HIP_RESIZE_SHARPEN"] = "1"
outs = libvips_amd.resize_sharpen_batch(ims, 0.125, threads=4)
assert mock.mock_hip_launches() - n0 == 2, mock.mock_hip_launches() - n0   # 2 chunks x one kernel
assert mock.mock_hip_live_streams() == s0
assert all((o.width, o.height, o.bands) == (86, 8, 3) for o in outs)
del os.environ["VIPS_HIP_RESIZE_SHARPEN"]
libvips_amd.resize_sharpen_batch(ims, 0.125, threads=4)
before = mock.mock_hip_live_streams()
assert before == s0 + 2, (s0, before)  # the two CU partitions, made once and kept
for overlap in ("", "0"):
    os.environ["VIPS_HIP_BATCH_OVERLAP"] = overlap
    if not overlap:
        del os.environ["VIPS_HIP_BATCH_OVERLAP"]
    n0 = mock.mock_hip_launches()
    outs = libvips_amd.resize_sharpen_batch(ims, 0.125, threads=4)
    assert mock.mock_hip_launches() - n0 == 4, mock.mock_hip_launches() - n0   # 2 chunks x (resize + sharpen)
    assert mock.mock_hip_live_streams() == before
    assert all((o.width, o.height, o.bands) == (86, 8, 3) for o in outs)
os.environ.pop("VIPS_HIP_BATCH_OVERLAP", None)
n0 = mock.mock_hip_launches()
outs = libvips_amd.resize_sharpen_batch(ims, 0.125, sharpen=False, threads=4)
assert mock.mock_hip_launches() - n0 == 2
mixed = ims[:3] + [Image.new_from_array(np.zeros((64, 700, 3), np.uint8), interpretation="srgb")]
outs = libvips_amd.resize_sharpen_batch(mixed, 0.125, threads=2)
assert [(o.width, o.height) for o in outs] == [(86, 8)] * 3 + [(88, 8)]
# the C caller's side of a batch: arrays in, arrays out, the results released in one call (their
# memory goes back to the pool: the next batch allocates nothing new)
from libvips_amd._ffi import lib
n = len(ims)
hin = (ctypes.c_void_p * n)(*[im._h.value for im in ims])
hout = (ctypes.c_void_p * n)()
mock.mock_hip_mallocs.restype = ctypes.c_long
for rep in range(3):
    assert lib.vips_hip_resize_sharpen_batch(hin, n, hout, 0.125, 5, 2.0, 0.5, 2.0, 10.0, 20.0, 0.0, 3.0, 4) == 0
    assert all(hout[i] for i in range(n))
    if rep == 1:
        m0 = mock.mock_hip_mallocs()
    lib.vips_hip_image_unref_many(hout, n)
    assert not any(hout[i] for i in range(n))
assert mock.mock_hip_mallocs() == m0, (m0, mock.mock_hip_mallocs())
lib.vips_hip_image_unref_many(hout, n)   # all NULL: nothing to do
lib.vips_hip_image_unref_many(None, 5)
# the queued form of a uniform batch returns without a single host wait (the synchronous form
# waits for the caller's stream and the two partitions); a mixed batch completes before the return
mock.mock_hip_syncs.restype = ctypes.c_long
s0 = mock.mock_hip_syncs()
assert lib.vips_hip_resize_sharpen_batch_queue(hin, n, hout, 0.125, 5, 2.0, 0.5, 2.0, 10.0, 20.0, 0.0, 3.0, 4) == 0
assert mock.mock_hip_syncs() == s0, (s0, mock.mock_hip_syncs())
lib.vips_hip_image_unref_many(hout, n)
assert lib.vips_hip_resize_sharpen_batch(hin, n, hout, 0.125, 5, 2.0, 0.5, 2.0, 10.0, 20.0, 0.0, 3.0, 4) == 0
assert mock.mock_hip_syncs() >= s0 + 1
lib.vips_hip_image_unref_many(hout, n)
s1 = mock.mock_hip_syncs()
outs = libvips_amd.resize_sharpen_batch(mixed, 0.125, threads=2, wait=False)
assert mock.mock_hip_syncs() > s1 and [(o.width, o.height) for o in outs] == [(86, 8)] * 3 + [(88, 8)]
''', tmp_path)


def test_two_devices_in_one_process(tmp_path):
    """The multi-device host logic against TWO fake devices (MOCK_HIP_DEVICES=2): threads that
    never call vips_hip_init() are dealt round-robin over $VIPS_HIP_DEVICES; a thread re-binds with
    vips_hip_init(); images remember their device and an operation runs where its input lives; a
    plan handle belongs to one device and fails loudly on the other; a batch whose images live on
    both devices is scattered to one worker per device; row strips over the devices exchange their
    halos with peer copies straight into persistent windows (pixel-exact: copies are real here) and
    vips_hip_conv_strips() returns one output strip per device."""
    run_child(r'''
import ctypes, math, os, threading
import numpy as np
import libvips_amd
from libvips_amd import Image, _ffi
from libvips_amd._ffi import Region, lib
from tests import helpers

mock = ctypes.CDLL(os.environ["LD_PRELOAD"])
for f in ("mock_hip_set_device_calls", "mock_hip_peer_copies", "mock_hip_peer_bytes"):
    getattr(mock, f).restype = ctypes.c_long
assert lib.vips_hip_device_count() == 2

# threads that never bind themselves: dealt round-robin over $VIPS_HIP_DEVICES
seen = []
def worker():
    im = Image.new_from_array(np.zeros((8, 8, 3), np.uint8))
    seen.append((lib.vips_hip_current_device(), lib.vips_hip_image_get_device(im._h)))
ts = [threading.Thread(target=worker) for _ in range(6)]
for t in ts:
    t.start()
    t.join()
assert sorted(d for d, _ in seen) == [0, 0, 0, 1, 1, 1], seen
assert all(d == i for d, i in seen)
devs = (ctypes.c_int * 8)()
assert lib.vips_hip_devices(devs, 8) == 2 and list(devs[:2]) == [0, 1]

# explicit binding, images remember their device, ops run where the pixels live
libvips_amd.init(1)
assert lib.vips_hip_current_device() == 1
on1 = Image.new_from_array(helpers.lcg_image(640, 64, 3, np.uint8, 7), interpretation="srgb")
assert lib.vips_hip_image_get_device(on1._h) == 1
libvips_amd.init(0)
on0 = Image.new_from_array(helpers.lcg_image(640, 64, 3, np.uint8, 8), interpretation="srgb")
out = on1.resize(0.5)                      # input on device 1: the thread is re-bound
assert lib.vips_hip_current_device() == 1 and lib.vips_hip_image_get_device(out._h) == 1
out = on0.gaussblur(2.0)
assert lib.vips_hip_current_device() == 0 and lib.vips_hip_image_get_device(out._h) == 0

# a plan handle belongs to the device it first ran on
rv = _ffi.check_handle(lib.vips_hip_reduce_new(5, 2.0, 64, 32, math.nan))
src, dst = on0.region(), Image.new_from_array(np.zeros((32, 640, 3), np.uint8)).region()
assert lib.vips_hip_reducev_gen(rv, ctypes.byref(src), ctypes.byref(dst)) == 0
libvips_amd.init(1)
assert lib.vips_hip_reducev_gen(rv, ctypes.byref(src), ctypes.byref(dst)) == -1
assert "device 0" in _ffi.error_buffer() and "device 1" in _ffi.error_buffer(), _ffi.error_buffer()
lib.vips_hip_error_clear()
lib.vips_hip_reduce_free(rv)

# a batch over both devices: one worker per device, outputs stay where their inputs are
ims = []
for k in range(10):
    libvips_amd.init(k % 2)
    ims.append(Image.new_from_array(np.zeros((64, 688, 3), np.uint8), interpretation="srgb"))
libvips_amd.init(0)
before = [mock.mock_hip_set_device_calls(d) for d in (0, 1)]
outs = libvips_amd.resize_sharpen_batch(ims, 0.125, threads=2)
assert [lib.vips_hip_image_get_device(o._h) for o in outs] == [k % 2 for k in range(10)]
assert all((o.width, o.height, o.bands) == (86, 8, 3) for o in outs)
assert all(mock.mock_hip_set_device_calls(d) > before[d] for d in (0, 1))
assert lib.vips_hip_current_device() == 0

# row strips over the devices: persistent windows, halos by peer copies
H, W, halo = 101, 37, 15
full = helpers.lcg_image(W, H, 1, np.uint16, 91)
devices = (ctypes.c_int * 3)(0, 1, 0)
strips = _ffi.check_handle(lib.vips_hip_strips_new(W, H, 1, 2, 3, devices, halo))
assert lib.vips_hip_strips_count(strips) == 3
wins = []
for k in range(3):
    dev, own, win = ctypes.c_int(), Region(), Region()
    assert lib.vips_hip_strips_region(strips, k, ctypes.byref(dev), ctypes.byref(own), ctypes.byref(win)) == 0
    assert dev.value == devices[k] and own.top >= win.top and own.top + own.height <= win.top + win.height
    rows = np.ascontiguousarray(full[own.top:own.top + own.height])
    ctypes.memmove(own.data, rows.ctypes.data, rows.nbytes)   # "device" memory is host memory here
    wins.append((win.top, win.height, win.data))
assert [w[0] for w in wins] == [0, 34 - halo, 68 - halo] and wins[2][0] + wins[2][1] == H
n0, b0 = mock.mock_hip_peer_copies(), mock.mock_hip_peer_bytes()
for step in range(2):                                          # persistent: nothing is rebuilt
    assert lib.vips_hip_strips_exchange(strips) == 0
    for top, height, data in wins:
        got = np.frombuffer((ctypes.c_char * (height * W * 2)).from_address(data), dtype=np.uint16).reshape(height, W, 1)
        assert np.array_equal(got, full[top:top + height]), (step, top)
assert mock.mock_hip_peer_copies() - n0 == 2 * 4                # 2 boundaries x 2 directions, twice
assert mock.mock_hip_peer_bytes() - b0 == 2 * 4 * halo * W * 2
mask, scale = libvips_amd.gaussmat(5, 0.01, False, "float")
m = np.ascontiguousarray(mask, dtype=np.float64)
outs = (ctypes.c_void_p * 3)()
assert lib.vips_hip_conv_strips(strips, outs, m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), m.shape[1], m.shape[0],
                                scale, 0.0, 1) == 0, _ffi.error_buffer()
for k in range(3):
    assert lib.vips_hip_image_get_device(outs[k]) == devices[k]
    assert lib.vips_hip_image_get_width(outs[k]) == W and lib.vips_hip_image_get_format(outs[k]) == 6
    lib.vips_hip_image_unref(outs[k])
short = _ffi.check_handle(lib.vips_hip_strips_new(W, H, 1, 2, 2, devices, 3))
assert lib.vips_hip_conv_strips(short, outs, m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), m.shape[1], m.shape[0],
                                scale, 0.0, 1) == -1 and "halo" in _ffi.error_buffer()
lib.vips_hip_error_clear()
lib.vips_hip_strips_free(short)
lib.vips_hip_strips_free(strips)
assert lib.vips_hip_current_device() == 0
''', tmp_path, {"MOCK_HIP_DEVICES": "2", "VIPS_HIP_DEVICES": "0,1"})


def test_eight_devices_uneven_strips_and_streams(tmp_path):
    """EIGHT fake devices (the node BASELINE's multi-GPU configs name): worker threads dealt over all
    eight; a batch spread over them comes back where its inputs live; an image of 1003 rows as 8
    strips of uneven heights (1003 = 3 x 126 + 5 x 125) with 15-row halos moved by peer copies --
    every window holds exactly the image's rows, 14 boundaries x 2 directions; and a caller's own
    stream (vips_hip_set_stream) survives the library's visits to the other devices (ADVICE r3:
    ScopedDevice used to drop it)."""
    run_child(r'''
import ctypes, os, threading
import numpy as np
import libvips_amd
from libvips_amd import Image, _ffi
from libvips_amd._ffi import Region, lib
from tests import helpers

mock = ctypes.CDLL(os.environ["LD_PRELOAD"])
for f in ("mock_hip_peer_copies", "mock_hip_peer_bytes"):
    getattr(mock, f).restype = ctypes.c_long
assert lib.vips_hip_device_count() == 8
seen = []
def worker():
    Image.new_from_array(np.zeros((8, 8, 3), np.uint8))
    seen.append(lib.vips_hip_current_device())
ts = [threading.Thread(target=worker) for _ in range(16)]
for t in ts:
    t.start()
    t.join()
assert sorted(seen) == sorted(list(range(8)) * 2), seen

ims = []
for k in range(20):
    libvips_amd.init(k % 8)
    ims.append(Image.new_from_array(np.zeros((64, 688, 3), np.uint8), interpretation="srgb"))
libvips_amd.init(3)
# the caller's own stream on device 3
lib.vips_hip_stream_new.restype = ctypes.c_void_p
mine = ctypes.c_void_p(lib.vips_hip_stream_new())
assert lib.vips_hip_set_stream(mine) == 0 and lib.vips_hip_get_stream() == mine.value
outs = libvips_amd.resize_sharpen_batch(ims, 0.125, threads=2)
assert [lib.vips_hip_image_get_device(o._h) for o in outs] == [k % 8 for k in range(20)]
assert lib.vips_hip_current_device() == 3 and lib.vips_hip_get_stream() == mine.value

H, W, halo = 1003, 20, 15
full = helpers.lcg_image(W, H, 1, np.uint16, 93)
devices = (ctypes.c_int * 8)(*range(8))
strips = _ffi.check_handle(lib.vips_hip_strips_new(W, H, 1, 2, 8, devices, halo))
heights, wins = [], []
for k in range(8):
    dev, own, win = ctypes.c_int(), Region(), Region()
    assert lib.vips_hip_strips_region(strips, k, ctypes.byref(dev), ctypes.byref(own), ctypes.byref(win)) == 0
    assert dev.value == k
    heights.append(own.height)
    rows = np.ascontiguousarray(full[own.top:own.top + own.height])
    ctypes.memmove(own.data, rows.ctypes.data, rows.nbytes)
    wins.append((win.top, win.height, win.data))
assert heights == [126, 126, 126, 125, 125, 125, 125, 125] and sum(heights) == H
n0, b0 = mock.mock_hip_peer_copies(), mock.mock_hip_peer_bytes()
assert lib.vips_hip_strips_exchange(strips) == 0
for top, height, data in wins:
    got = np.frombuffer((ctypes.c_char * (height * W * 2)).from_address(data), dtype=np.uint16).reshape(height, W, 1)
    assert np.array_equal(got, full[top:top + height]), top
assert mock.mock_hip_peer_copies() - n0 == 7 * 2
assert mock.mock_hip_peer_bytes() - b0 == 7 * 2 * halo * W * 2
# ... the visit to eight devices left the caller where it was, on its own stream
assert lib.vips_hip_current_device() == 3 and lib.vips_hip_get_stream() == mine.value
lib.vips_hip_strips_free(strips)
lib.vips_hip_set_stream(None)
lib.vips_hip_stream_free(mine)
''', tmp_path, {"MOCK_HIP_DEVICES": "8", "VIPS_HIP_DEVICES": "0,1,2,3,4,5,6,7"})


@pytest.mark.skipif(not helpers.have_module(), reason="oracle/_ref or host/_build missing")
def test_module_plumbing_through_libvips(tmp_path):
    """The *_hip operations driven through the reference's own operation API: build(), the
    device-image link between chained ops, generate; geometry against the built-ins."""
    run_child(r'''
import os
import numpy as np
from tests import helpers
from tests.helpers import Ref
import tests.test_zz_jpeg as J

Ref.load_module()
src = helpers.lcg_image(640, 400, 3, np.uint8, 100)
for hip, ref, args in (("thumbnail_image_hip", "thumbnail_image", "width=100,height=100,crop=centre"),
                       ("thumbnail_image_hip", "thumbnail_image", "width=80,height=200,crop=high"),
                       ("resize_hip", "resize", "scale=0.37"),
                       ("gaussblur_hip", "gaussblur", "sigma=2,precision=approximate"),
                       ("reduce_hip", "reduce", "hshrink=2.5,vshrink=3.3"),
                       ("colourspace_hip", "colourspace", "space=lab"),
                       ("sharpen_hip", "sharpen", "")):
    got = Ref.run(hip, src, args, 22)
    want = Ref.run(ref, src, args, 22)
    assert got.shape == want.shape and got.dtype == want.dtype, (hip, args)
got = Ref.run_chain("resize_hip:scale=0.25;sharpen_hip:;cast_hip:format=ushort", src, 22)
want = Ref.run_chain("resize:scale=0.25;sharpen:;cast:format=ushort", src, 22)
assert got.shape == want.shape and got.dtype == want.dtype
if J._ref_has_jpeg():
    p = os.path.join(TMP, "t.jpg")
    J.make_jpeg(p, 2400, 1600)
    for args in ("width=200", "width=128,height=128,crop=centre", "width=300,height=100,size=force"):
        got, _, _ = Ref.create("thumbnail_hip", "filename=%s,%s" % (p, args))
        want, _, _ = Ref.create("thumbnail", "filename=%s,%s" % (p, args))
        assert got.shape == want.shape and got.dtype == want.dtype, args
''', tmp_path)
