"""CPU: the C-ABI library loads without a GPU and exports every symbol the header
declares; the host-side table builders agree with the oracle; error behaviour."""
import ctypes
import math
import os
import re

import numpy as np
import pytest

import libvips_amd
from libvips_amd import _ffi
from tests import helpers
from tests.helpers import Port


def header_symbols():
    text = open(libvips_amd.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"VIPS_HIP_API[^;(]*?\b(vips_hip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    syms = header_symbols()
    assert len(syms) > 60
    missing = [s for s in syms if not hasattr(_ffi.lib, s)]
    assert missing == []


def test_binding_covers_header():
    assert _ffi.MISSING == []
    unbound = [s for s in header_symbols() if s not in _ffi._SIGNATURES]
    assert unbound == []


def test_reduce_get_points():
    for kernel in range(8):
        for shrink in (1.0, 1.1, 1.5, 2.0, 3.3, 8.0, 100.5):
            assert _ffi.lib.vips_hip_reduce_get_points(kernel, shrink) == \
                Port.lib().port_reduce_get_points(kernel, shrink)
    # reduceh.cpp:129-130: lanczos3 at x8 -> 49 taps
    assert _ffi.lib.vips_hip_reduce_get_points(5, 8.0) == 49


@pytest.mark.parametrize("kernel", range(1, 8))
@pytest.mark.parametrize("shrink", [1.1, 2.0, 2.718, 8.0])
def test_reduce_tables_match_oracle(kernel, shrink):
    r = _ffi.lib.vips_hip_reduce_new(kernel, shrink, 1000, int(1000 / shrink + 0.5), math.nan)
    assert r
    try:
        n = _ffi.lib.vips_hip_reduce_get_n_point(r)
        assert n == Port.lib().port_reduce_get_points(kernel, shrink)
        for phase in range(65):
            cf = (ctypes.c_double * n)()
            cs = (ctypes.c_short * n)()
            assert _ffi.lib.vips_hip_reduce_get_matrixf(r, phase, cf) == 0
            assert _ffi.lib.vips_hip_reduce_get_matrixs(r, phase, cs) == 0
            want = (ctypes.c_double * n)()
            Port.lib().port_reduce_make_mask(want, kernel, n, shrink, np.float32(phase) / 64)
            want = np.array(want[:])
            assert np.array_equal(np.array(cf[:]).view(np.uint64), want.view(np.uint64))
            # reduceh.cpp:497-499: (short)(c * 4096), truncation
            assert np.array_equal(np.array(cs[:]), np.trunc(want * 4096).astype(np.int16))
    finally:
        _ffi.lib.vips_hip_reduce_free(r)


def test_c2_table_shape():
    # x8 lanczos3 -> 49 taps of which tap 48 is 0, offset -0.5
    r = _ffi.lib.vips_hip_reduce_new(5, 8.0, 16384, 2048, math.nan)
    try:
        assert _ffi.lib.vips_hip_reduce_get_n_point(r) == 49
        assert _ffi.lib.vips_hip_reduce_get_offset(r) == -0.5
        cs = (ctypes.c_short * 49)()
        _ffi.lib.vips_hip_reduce_get_matrixs(r, 0, cs)
        assert cs[48] == 0
        assert list(cs[:24]) == list(cs[47:23:-1])  # symmetric
    finally:
        _ffi.lib.vips_hip_reduce_free(r)


def test_reduce_new_errors():
    lib = _ffi.lib
    lib.vips_hip_error_clear()
    assert not lib.vips_hip_reduce_new(5, 0.5, 100, 200, math.nan)
    assert "reduce factor should be >= 1.0" in _ffi.error_buffer()
    lib.vips_hip_error_clear()
    assert not lib.vips_hip_reduce_new(5, 1000.0, 100000, 100, math.nan)
    assert "reduce factor too large" in _ffi.error_buffer()
    lib.vips_hip_error_clear()
    assert not lib.vips_hip_reduce_new(5, 2.0, 1, 0, math.nan)
    assert "image has shrunk to nothing" in _ffi.error_buffer()
    lib.vips_hip_error_clear()


def test_shrink_out_size():
    lib = _ffi.lib
    for size in (1, 9, 10, 11, 4096, 16383):
        for shrink in (2, 3, 4, 7, 8):
            for ceil in (0, 1):
                assert lib.vips_hip_shrink_out_size(size, shrink, ceil) == \
                    Port.lib().port_shrink_out_size(size, shrink, ceil)


def test_no_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _ffi.lib
    lib.vips_hip_error_clear()
    assert lib.vips_hip_init(0) != 0
    assert "no HIP device" in _ffi.error_buffer()
    with pytest.raises(libvips_amd.VipsHipError):
        libvips_amd.Image.new_from_array(np.zeros((4, 4, 3), np.uint8))


def test_affine_out_size():
    # VIPS_ROUND_INT(scale * in_size), resample/transform.c:220-231; host-only entry point
    import ctypes

    from tests.helpers import Port

    lib = _ffi.lib
    port = Port.lib()
    port.port_affine_out_size.argtypes = [ctypes.c_int, ctypes.c_double]
    for size in (1, 2, 7, 100, 4097, 16384):
        for scale in (1.0, 1.01, 1.5, 2.0, 2.3, 2.5, 3.999, 7.3, 10.0):
            assert lib.vips_hip_affine_out_size(size, scale) == port.port_affine_out_size(size, scale)


def test_port_affine_vs_ref_mixed():
    """The upsizing port against the compiled reference beyond the golden cases: every
    interpolator, odd sizes, up / down mixes (the goldens pin it where the reference is absent)."""
    import numpy as np

    from tests import helpers
    from tests.helpers import Port, Ref

    if not helpers.have_ref():
        pytest.skip("oracle/_ref not built")
    for dtype in (np.uint8, np.int16, np.float32):
        for kernel in ("nearest", "linear", "cubic"):
            for (w, h, b, hs, vs) in ((33, 21, 3, 1.7, 2.9), (19, 40, 1, 5.0, 1.2), (64, 9, 4, 2.0, 0.4)):
                if kernel == "nearest" and vs < 1:
                    continue
                src = helpers.lcg_image(w, h, b, dtype, 31)
                want = Ref.run("resize", src, "scale=%r,vscale=%r,kernel=%s" % (hs, vs, kernel))
                got = Port.resize(src, hs, vs, kernel=kernel)
                assert got.shape == want.shape and np.array_equal(got, want), (dtype, kernel, w, h, hs, vs)


def test_every_entry_point_survives_null_arguments():
    """Each C-ABI function called with NULL pointers and zero numbers must return (an error or a
    neutral value), not crash: run in a child so that a crash is a test failure, not the end of
    the suite."""
    import subprocess
    import sys

    code = r'''
import ctypes, sys
sys.path.insert(0, %r)
from libvips_amd import _ffi
# all-zero numbers, then plausible non-zero numbers (2, 2.0) so that no early "bad factor"
# error shields a pointer dereference
for ival, fval in ((0, 0.0), (2, 2.0), (1, 0.5)):
    for name in sorted(_ffi._SIGNATURES):
        if ival and name in ("vips_hip_init", "vips_hip_malloc", "vips_hip_malloc_host"):
            continue  # legitimately act on plain numbers
        restype, argtypes = _ffi._SIGNATURES[name]
        args = []
        for t in argtypes:
            if t in (ctypes.c_int, ctypes.c_size_t, ctypes.c_longlong):
                args.append(ival)
            elif t in (ctypes.c_double, ctypes.c_float):
                args.append(fval)
            else:
                args.append(None)
        print(name, ival, flush=True)
        getattr(_ffi.lib, name)(*args)
        _ffi.lib.vips_hip_error_clear()
print("ALL-RETURNED", flush=True)
''' % helpers.ROOT
    proc = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    tail = proc.stdout.strip().splitlines()[-3:]
    assert proc.returncode == 0 and "ALL-RETURNED" in proc.stdout, tail


def test_device_code_has_no_ashr_pk(tmp_path):
    """ROCm 7.2's clang folds "arithmetic shift, saturate to u8, pack" into v_ashr_pk_u8_i32 and
    then ORs more bytes on top as if the instruction zeroed the upper half of its destination;
    on gfx950 it does not (measured in resize_tail.hip: wrong byte 2).  The kernels that clip
    and pack make the shifted value opaque first -- check that no code object of the library
    holds the instruction."""
    import glob
    import shutil
    import subprocess

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump")
    so = shutil.copy(_ffi.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", os.path.basename(so)], cwd=tmp_path, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objects = glob.glob(str(tmp_path / "lib.so.*gfx950"))
    assert objects, "no gfx950 code objects in the library"
    for path in objects:
        text = subprocess.run([objdump, "-d", path], check=True, capture_output=True, text=True).stdout
        assert "v_ashr_pk_u8_i32" not in text, path
