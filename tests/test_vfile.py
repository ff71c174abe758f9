"""The libvips native .v format next to the device path.

CPU: the header reader (host only) against files written by the REFERENCE's own CLI and against
hand-made headers (big-endian, truncated, wrong magic, bad coding) -- iofuncs/vips.c:301-394,
iofuncs/image.c:966-979.  GPU: file -> HBM -> file round trips, and the reference's CLI reading a
file this library wrote."""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

from tests import helpers

VIPS = os.path.join(helpers.ROOT, "oracle", "_ref", "bin", "vips")
VIPSHEADER = os.path.join(helpers.ROOT, "oracle", "_ref", "bin", "vipsheader")
needs_cli = pytest.mark.skipif(not os.path.exists(VIPS), reason="oracle/_ref/bin not built")


def read_header(path):
    from libvips_amd import _ffi

    h = _ffi.VHeader()
    r = _ffi.lib.vips_hip_vfile_read_header(os.fsencode(path), ctypes.byref(h))
    if r != 0:
        msg = _ffi.error_buffer()
        _ffi.lib.vips_hip_error_clear()
        raise RuntimeError(msg)
    return h


def test_header_of_a_helper_written_file(tmp_path):
    src = helpers.lcg_image(37, 21, 3, np.uint16, 91)
    path = str(tmp_path / "a.v")
    helpers.write_v(path, src, interpretation=25)
    h = read_header(path)
    assert (h.width, h.height, h.bands, h.format, h.coding, h.interpretation) == (37, 21, 3, 2, 0, 25)
    assert (h.xres, h.yres, h.xoffset, h.yoffset, h.msb_first) == (1.0, 1.0, 0, 0, 0)
    assert h.data_offset == 64 and h.data_size == 37 * 21 * 3 * 2


@needs_cli
@pytest.mark.parametrize("dtype,interp", [(np.uint8, 22), (np.int16, 21), (np.float32, 28), (np.float64, 0)])
def test_header_of_a_reference_written_file(tmp_path, dtype, interp):
    """`vips copy` rewrites the file through the reference's own writer (header + pixels + XML)."""
    src = helpers.lcg_image(45, 31, 3, dtype, 92)
    a, b = str(tmp_path / "a.v"), str(tmp_path / "b.v")
    helpers.write_v(a, src, interpretation=interp)
    proc = subprocess.run([VIPS, "copy", a, b, "--xres", "3.5", "--xoffset", "7"], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True)
    assert proc.returncode == 0, proc.stdout
    h = read_header(b)
    assert (h.width, h.height, h.bands, h.format, h.coding, h.interpretation) == \
        (45, 31, 3, helpers.DTYPE_FORMATS[np.dtype(dtype)], 0, interp)
    assert h.xres == 3.5 and h.xoffset == 7
    assert os.path.getsize(b) > 64 + h.data_size  # the XML block sits behind the pixels
    back, _ = helpers.read_v(b)
    assert np.array_equal(back, src)
    # the helper's writer (what the GPU round-trip test compares the product's files with) makes
    # the header the reference's own writer makes
    c = str(tmp_path / "c.v")
    proc = subprocess.run([VIPS, "copy", a, c], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert proc.returncode == 0, proc.stdout
    assert open(c, "rb").read(64) == open(a, "rb").read(64)


def test_header_errors(tmp_path):
    src = helpers.lcg_image(16, 8, 1, np.uint8, 93)
    good = str(tmp_path / "good.v")
    helpers.write_v(good, src)
    raw = open(good, "rb").read()

    short = str(tmp_path / "short.v")
    open(short, "wb").write(raw[:-1])
    with pytest.raises(RuntimeError, match="file too short"):
        read_header(short)

    notvips = str(tmp_path / "not.v")
    open(notvips, "wb").write(b"\x89PNG" + raw[4:])
    with pytest.raises(RuntimeError, match="is not a VIPS image"):
        read_header(notvips)

    coding = str(tmp_path / "coding.v")
    open(coding, "wb").write(raw[:24] + struct.pack("<i", 5) + raw[28:])
    with pytest.raises(RuntimeError, match="unknown coding"):
        read_header(coding)

    with pytest.raises(RuntimeError, match="unable to open"):
        read_header(str(tmp_path / "missing.v"))

    # SPARC order: magic 08 f2 a6 b6 and big-endian fields
    be = str(tmp_path / "be.v")
    header = bytes([0x08, 0xF2, 0xA6, 0xB6]) + struct.pack(">iiiiiiiffiiii", 16, 8, 1, 8, 0, 0, 1, 2.0, 2.0,
                                                          0, 0, 3, 4)
    open(be, "wb").write(header.ljust(64, b"\0") + src.tobytes())
    h = read_header(be)
    assert (h.width, h.height, h.bands, h.msb_first, h.xres, h.xoffset, h.yoffset) == (16, 8, 1, 1, 2.0, 3, 4)

    # a hostile header: 2^23 x 2^23 x 2^14 dpcomplex (16 bytes) wraps a 64-bit byte count to 0 --
    # refused, instead of passing the length check and describing a 256-byte image
    huge = str(tmp_path / "huge.v")
    open(huge, "wb").write(raw[:4] + struct.pack("<iiiii", 1 << 23, 1 << 23, 1 << 14, 128, 9) + raw[24:])
    with pytest.raises(RuntimeError, match="too large"):
        read_header(huge)

    # unknown interpretation value -> VIPS_INTERPRETATION_ERROR, out-of-range sizes are clipped
    odd = str(tmp_path / "odd.v")
    open(odd, "wb").write(raw[:28] + struct.pack("<i", 14) + raw[32:])
    assert read_header(odd).interpretation == -1


@pytest.mark.gpu
def test_large_file_uses_several_chunks(tmp_path):
    """> 2 staging chunks (32 MiB each): the double-buffer hand-over in both directions."""
    import libvips_amd
    from libvips_amd import Image

    libvips_amd.init(0)
    src = helpers.lcg_image(4096, 2800, 4, np.uint8, 95)  # 43.75 MiB... x2.4 -> 3 chunks below
    src = np.concatenate([src, src[::-1], src[:1000]], axis=0)  # 6600 rows x 16 KiB = 103 MiB, 4 chunks
    a, b = str(tmp_path / "big.v"), str(tmp_path / "big_out.v")
    helpers.write_v(a, src, interpretation=22)
    im = Image.new_from_file(a)
    out = im.reduce(8, 8)  # the loaded image is a normal device image
    assert out.width == 512 and out.height == 825
    im.write_to_file(b)
    back, interp = helpers.read_v(b)
    assert interp == 22 and np.array_equal(back, src)


@pytest.mark.gpu
@needs_cli
def test_reference_reads_what_we_write(tmp_path):
    import libvips_amd
    from libvips_amd import Image

    libvips_amd.init(0)
    src = helpers.lcg_image(300, 200, 3, np.uint8, 96)
    path = str(tmp_path / "thumb.v")
    Image.new_from_array(src, interpretation="srgb").thumbnail_image(64).write_to_file(path)
    proc = subprocess.run([VIPSHEADER, path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert proc.returncode == 0 and "64x43 uchar, 3 bands, srgb" in proc.stdout, proc.stdout
    out = str(tmp_path / "inv.v")
    proc = subprocess.run([VIPS, "invert", path, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert proc.returncode == 0, proc.stdout
    ours, _ = helpers.read_v(path)
    inv, _ = helpers.read_v(out)
    assert np.array_equal(inv, 255 - ours)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32, np.complex64])
def test_file_round_trip_through_hbm(tmp_path, dtype):
    import libvips_amd
    from libvips_amd import Image

    libvips_amd.init(0)
    if dtype == np.complex64:
        re = helpers.lcg_image(50, 40, 2, np.float32, 94)
        src = (re + 1j * re[::-1]).astype(np.complex64)
    else:
        src = helpers.lcg_image(123, 77, 3, dtype, 94)
    a, b = str(tmp_path / "a.v"), str(tmp_path / "b.v")
    helpers.write_v(a, src, interpretation=22)
    im = Image.new_from_file(a)
    assert (im.width, im.height, im.bands) == (src.shape[1], src.shape[0], src.shape[2])
    assert np.array_equal(im.numpy(), src)
    im.write_to_file(b)
    ours, theirs = open(b, "rb").read(), open(a, "rb").read()
    assert ours[:64].hex() == theirs[:64].hex()  # the header the reference's writer makes
    assert ours[64:] == theirs[64:]
