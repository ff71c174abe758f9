"""JPEG shrink-on-load in front of the device path.

CPU: the host decode (libjpeg driven the way foreign/jpeg2vips.c drives it) against the
REFERENCE's jpegload for every shrink, and the choice of the block shrink against what the
reference's vipsthumbnail logs.  GPU: vips_hip_thumbnail against the reference's vipsthumbnail
CLI on the same file.  The test JPEGs are made with Pillow (present in the image)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests import helpers
from tests.helpers import Ref

PIL = pytest.importorskip("PIL.Image")
VIPSTHUMBNAIL = os.path.join(helpers.ROOT, "oracle", "_ref", "bin", "vipsthumbnail")


def _ref_has_jpeg():
    if not helpers.have_ref():
        return False
    try:
        Ref.lib()
        lib = ctypes.CDLL(helpers.REF_LIB.replace("libref_shim", "libvips"))
        lib.vips_type_find.restype = ctypes.c_size_t
        lib.vips_type_find.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        return lib.vips_type_find(b"VipsOperation", b"jpegload") != 0
    except Exception:
        return False


needs_ref_jpeg = pytest.mark.skipif(not _ref_has_jpeg(), reason="oracle/_ref built without libjpeg")


def make_jpeg(path, width, height, grey=False, quality=90, subsampling=None, **extra):
    y, x = np.mgrid[0:height, 0:width]
    img = np.stack([(np.sin(x / 37.0) + 1) * 127, (np.cos(y / 23.0) + 1) * 127, (x + y) % 256], axis=2)
    img = (img.astype(int) + helpers.lcg_image(width, height, 3, np.uint8, 97) // 8).clip(0, 255).astype(np.uint8)
    im = PIL.fromarray(img)
    if grey:
        im = im.convert("L")
    kw = dict(extra)
    if subsampling is not None:
        kw["subsampling"] = subsampling
    im.save(path, quality=quality, **kw)


def product_header(path, shrink=1):
    from libvips_amd import _ffi

    h = _ffi.JpegHeader()
    if _ffi.lib.vips_hip_jpeg_read_header(os.fsencode(path), shrink, ctypes.byref(h)) != 0:
        msg = _ffi.error_buffer()
        _ffi.lib.vips_hip_error_clear()
        raise RuntimeError(msg)
    return h


def product_decode(path, shrink):
    from libvips_amd import _ffi

    h = product_header(path, shrink)
    out = np.empty((h.height, h.width, h.bands), np.uint8)
    if _ffi.lib.vips_hip_jpeg_read_to_memory(os.fsencode(path), shrink, out.ctypes.data, out.size) != 0:
        msg = _ffi.error_buffer()
        _ffi.lib.vips_hip_error_clear()
        raise RuntimeError(msg)
    return out, h


CASES = [(641, 487, False, None), (1801, 1203, False, 0), (1000, 750, True, None), (333, 517, False, 2),
         (64, 48, False, None), (17, 9, True, None)]


@needs_ref_jpeg
@pytest.mark.parametrize("width,height,grey,sub", CASES)
def test_decode_matches_reference_jpegload(tmp_path, width, height, grey, sub):
    path = str(tmp_path / "t.jpg")
    make_jpeg(path, width, height, grey, subsampling=sub)
    for shrink in (1, 2, 4, 8):
        if width // shrink < 1 or height // shrink < 1:
            continue
        want, _, _ = Ref.create("jpegload", "filename=%s,shrink=%d" % (path, shrink))
        got, h = product_decode(path, shrink)
        assert (h.image_width, h.image_height) == (width, height)
        assert got.shape == want.shape == (height // shrink, width // shrink, 1 if grey else 3)
        assert h.interpretation == (1 if grey else 22)
        assert np.array_equal(got, want), shrink


@needs_ref_jpeg
def test_progressive_and_quality_variants(tmp_path):
    path = str(tmp_path / "p.jpg")
    for kw in (dict(progressive=True), dict(quality=30), dict(quality=100, subsampling=0), dict(optimize=True)):
        make_jpeg(path, 413, 305, **kw)
        for shrink in (1, 4):
            want, _, _ = Ref.create("jpegload", "filename=%s,shrink=%d" % (path, shrink))
            got, _ = product_decode(path, shrink)
            assert np.array_equal(got, want), (kw, shrink)


def test_header_flags_and_errors(tmp_path):
    path = str(tmp_path / "t.jpg")
    make_jpeg(path, 120, 80)
    h = product_header(path)
    assert (h.width, h.height, h.bands, h.orientation, h.has_icc) == (120, 80, 3, 0, 0)
    # EXIF orientation 6 and an ICC profile are seen (and make vips_hip_thumbnail refuse the file)
    exif = PIL.Exif()
    exif[0x0112] = 6
    make_jpeg(path, 120, 80, exif=exif.tobytes())
    assert product_header(path).orientation == 6
    make_jpeg(path, 120, 80, icc_profile=b"\0" * 200)
    assert product_header(path).has_icc == 1

    make_jpeg(path, 120, 80)
    with pytest.raises(RuntimeError, match="bad shrink factor"):
        product_header(path, 3)
    with pytest.raises(RuntimeError, match="unable to open"):
        product_header(str(tmp_path / "missing.jpg"))
    bad = str(tmp_path / "bad.jpg")
    open(bad, "wb").write(b"\xff\xd8" + b"garbage" * 10)
    with pytest.raises(RuntimeError):
        product_header(bad)
    trunc = str(tmp_path / "trunc.jpg")
    raw = open(path, "rb").read()
    open(trunc, "wb").write(raw[:len(raw) // 2])
    out, _ = product_decode(trunc, 1)  # a truncated scan is a warning, not an error (fail_on none)
    assert out.shape == (80, 120, 3)


@needs_ref_jpeg
def test_truncated_file_decodes_like_the_reference(tmp_path):
    path, trunc = str(tmp_path / "t.jpg"), str(tmp_path / "trunc.jpg")
    make_jpeg(path, 200, 160)
    raw = open(path, "rb").read()
    open(trunc, "wb").write(raw[:len(raw) * 2 // 3])
    want, _, _ = Ref.create("jpegload", "filename=%s" % trunc)
    got, _ = product_decode(trunc, 1)
    assert np.array_equal(got, want)


def cli_thumbnail(tmp_path, src, size, extra=()):
    out_path = os.path.join(str(tmp_path), "cli_out.v")
    env = dict(os.environ, VIPS_INFO="1")
    proc = subprocess.run([VIPSTHUMBNAIL, src, "--size", size, "-o", out_path] + list(extra),
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, text=True)
    assert proc.returncode == 0, proc.stdout
    out, _ = helpers.read_v(out_path)
    factor = [ln for ln in proc.stdout.splitlines() if "pre-shrink" in ln]
    return out, int(factor[0].split("factor ")[1].split(" ")[0]) if factor else None


@needs_ref_jpeg
def test_find_jpegshrink_matches_what_the_reference_logs(tmp_path):
    from libvips_amd import _ffi

    from concurrent.futures import ThreadPoolExecutor

    cases = ((2000, 1500, "200x200", (), 0, 0, 0), (1801, 1203, "100x100", (), 0, 0, 0),
             (640, 480, "300x300", (), 0, 0, 0), (4000, 3000, "128x128", (), 0, 0, 0),
             (1600, 1200, "100x100", ("--linear",), 0, 1, 0), (1600, 400, "100x100!", (), 3, 0, 0),
             (1000, 800, "125x100", (), 0, 0, 0), (1600, 400, "100x100", ("--smartcrop", "centre"), 0, 0, 1),
             (900, 2000, "50x100", ("--smartcrop", "high"), 0, 0, 5))

    def logged(k):  # (two child processes of the reference's command line per case: side by side)
        w, h, size, args = cases[k][:4]
        tmp = tmp_path / ("case%d" % k)
        tmp.mkdir()
        path = str(tmp / "t.jpg")
        make_jpeg(path, w, h, quality=60)
        return cli_thumbnail(tmp, path, size, args)[1]

    with ThreadPoolExecutor(max_workers=min(len(cases), os.cpu_count() or 2)) as pool:
        factors = list(pool.map(logged, range(len(cases))))
    for (w, h, size, args, mode, linear, crop), factor in zip(cases, factors):
        tw, th = [int(v) for v in size.rstrip("!").split("x")]
        assert _ffi.lib.vips_hip_thumbnail_find_jpegshrink(w, h, tw, th, mode, linear, crop) == factor, (w, h, size)


@pytest.mark.gpu
@needs_ref_jpeg
@pytest.mark.parametrize("width,height,grey,size", [(2000, 1500, False, "200x200"), (1801, 1203, False, "100x100"),
                                                     (3001, 1999, True, "150x150"), (640, 480, False, "300x300"),
                                                     (1000, 800, False, "900x900")])
def test_thumbnail_of_a_jpeg_matches_the_reference_cli(tmp_path, width, height, grey, size):
    import libvips_amd
    from libvips_amd import Image

    libvips_amd.init(0)
    path = str(tmp_path / "t.jpg")
    make_jpeg(path, width, height, grey)
    want, _ = cli_thumbnail(tmp_path, path, size)
    tw, th = [int(v) for v in size.split("x")]
    got = Image.thumbnail(path, tw, th).numpy()
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
def test_thumbnail_of_a_v_file_and_refusals(tmp_path):
    import libvips_amd
    from libvips_amd import Image

    libvips_amd.init(0)
    src = helpers.lcg_image(517, 389, 3, np.uint8, 98)
    vpath = str(tmp_path / "src.v")
    helpers.write_v(vpath, src, interpretation=22)
    got = Image.thumbnail(vpath, 100).numpy()
    want = Image.new_from_array(src, interpretation="srgb").thumbnail_image(100).numpy()
    assert np.array_equal(got, want)
    jpath = str(tmp_path / "rot.jpg")
    exif = PIL.Exif()
    exif[0x0112] = 8
    make_jpeg(jpath, 300, 200, exif=exif.tobytes())
    with pytest.raises(libvips_amd.VipsHipError, match="auto-rotation"):
        Image.thumbnail(jpath, 64)
    make_jpeg(jpath, 300, 200, icc_profile=b"\0" * 200)
    with pytest.raises(libvips_amd.VipsHipError, match="ICC"):
        Image.thumbnail(jpath, 64, linear=True)
    # ... but without linear the reference leaves the pixels alone, and so does this path
    plain = str(tmp_path / "plain.jpg")
    make_jpeg(plain, 300, 200)
    assert np.array_equal(Image.thumbnail(jpath, 64).numpy(), Image.thumbnail(plain, 64).numpy())
    # shrink-on-load on its own
    make_jpeg(jpath, 300, 200)
    im = Image.new_from_jpeg(jpath, 2)
    assert (im.width, im.height, im.bands) == (150, 100, 3)


@needs_ref_jpeg
@pytest.mark.parametrize("width,height,grey,size", [(2000, 1500, False, "200x200"), (1801, 1203, False, "100x100"),
                                                     (3001, 1999, True, "150x150"), (640, 480, False, "300x300")])
def test_jpeg_pipeline_through_the_port_matches_the_reference_cli(tmp_path, width, height, grey, size):
    """The CPU twin of the GPU test above: product shrink choice + product host decode + the
    oracle port's thumbnail pipeline == the reference's vipsthumbnail on the same file."""
    from libvips_amd import _ffi

    path = str(tmp_path / "t.jpg")
    make_jpeg(path, width, height, grey)
    want, _ = cli_thumbnail(tmp_path, path, size)
    tw, th = [int(v) for v in size.split("x")]
    factor = _ffi.lib.vips_hip_thumbnail_find_jpegshrink(width, height, tw, th, 0, 0, 0)
    pre, _ = product_decode(path, factor)
    from tests.helpers import PortCC

    got = PortCC.thumbnail_image(pre, "b-w" if grey else "srgb", tw, th)
    assert got.shape == want.shape and np.array_equal(got, want)


@needs_ref_jpeg
@pytest.mark.parametrize("crop", ["centre", "low", "high", "all"])
def test_crop_modes_through_the_port_match_the_reference(crop):
    """thumbnail_image with a positional crop: the port against the compiled reference (the device
    path is compared with the port in the GPU test below)."""
    from tests.helpers import PortCC

    src = helpers.lcg_image(517, 389, 3, np.uint8, 99)
    for (tw, th, size) in ((100, 100, "both"), (60, 200, "both"), (400, 50, "down"), (700, 700, "both")):
        want = Ref.run("thumbnail_image", src, "width=%d,height=%d,size=%s,crop=%s" % (tw, th, size, crop), 22)
        got = PortCC.thumbnail_image(src, "srgb", tw, th, size=size, crop=crop)
        assert got.shape == want.shape and np.array_equal(got, want), (tw, th, size)


@pytest.mark.gpu
@pytest.mark.parametrize("crop", ["centre", "low", "high", "all"])
def test_hip_crop_modes_match_the_port(crop):
    import libvips_amd
    from libvips_amd import Image
    from tests.helpers import PortCC

    libvips_amd.init(0)
    src = helpers.lcg_image(517, 389, 4, np.uint8, 99)
    im = Image.new_from_array(src, interpretation="srgb")
    for (tw, th, size, linear) in ((100, 100, "both", False), (60, 200, "both", False), (400, 50, "down", False),
                                   (90, 90, "both", True)):
        want = PortCC.thumbnail_image(src, "srgb", tw, th, size=size, linear=linear, crop=crop)
        got = im.thumbnail_image(tw, th, size=size, linear=linear, crop=crop).numpy()
        assert got.shape == want.shape and np.array_equal(got, want), (tw, th, size, linear)
    assert np.array_equal(im.extract_area(10, 20, 300, 100).numpy(), src[20:120, 10:310])
    with pytest.raises(libvips_amd.VipsHipError, match="bad extract area"):
        im.extract_area(400, 0, 200, 10)
    with pytest.raises(libvips_amd.VipsHipError, match="attention"):
        im.thumbnail_image(64, crop="attention")


@pytest.mark.gpu
@pytest.mark.skipif(not helpers.have_module(), reason="oracle/_ref or host/_build missing")
def test_module_thumbnail_crop_matches_the_reference():
    Ref.load_module()
    src = helpers.lcg_image(640, 400, 3, np.uint8, 100)
    for args in ("width=100,height=100,crop=centre", "width=80,height=200,crop=high", "width=300,height=50,crop=low"):
        got = Ref.run("thumbnail_image_hip", src, args, 22)
        want = Ref.run("thumbnail_image", src, args, 22)
        assert got.shape == want.shape and np.array_equal(got, want), args


@needs_ref_jpeg
def test_corrupted_files_behave_like_the_reference(tmp_path):
    """Random byte damage: wherever the reference's jpegload still produces an image the host
    decoder produces the same pixels, and wherever it refuses so does the decoder (no crash)."""
    rng = np.random.RandomState(5)
    good = str(tmp_path / "good.jpg")
    make_jpeg(good, 160, 120, quality=75)
    raw = bytearray(open(good, "rb").read())
    bad = str(tmp_path / "bad.jpg")
    agree_ok = agree_fail = 0
    for trial in range(60):
        damaged = bytearray(raw)
        for _ in range(rng.randint(1, 6)):
            at = rng.randint(2, len(damaged))
            if trial % 3 == 0:
                at = rng.randint(2, min(len(damaged), 700))  # the headers and tables
            damaged[at] = rng.randint(0, 256)
        if trial % 10 == 9:
            damaged = damaged[:rng.randint(20, len(damaged))]
        open(bad, "wb").write(bytes(damaged))
        try:
            want, _, _ = Ref.create("jpegload", "filename=%s" % bad)
        except Exception:
            want = None
        try:
            got, _ = product_decode(bad, 1)
        except RuntimeError:
            got = None
        assert (want is None) == (got is None), trial
        if want is not None:
            assert got.shape == want.shape and np.array_equal(got, want), trial
            agree_ok += 1
        else:
            agree_fail += 1
    assert agree_ok > 5 and agree_fail > 0


@pytest.mark.gpu
def test_thumbnail_batch_on_host_threads(tmp_path):
    import libvips_amd
    from libvips_amd import Image

    libvips_amd.init(0)
    paths = []
    for i in range(10):
        p = str(tmp_path / ("b%d.jpg" % i))
        make_jpeg(p, 400 + 37 * i, 300 + 11 * i, grey=(i == 3), quality=70 + i)
        paths.append(p)
    paths.insert(4, str(tmp_path / "missing.jpg"))
    outs = Image.thumbnail_batch(paths, 96, 96, crop="centre", threads=4)
    assert len(outs) == len(paths)
    for p, o in zip(paths, outs):
        if p.endswith("missing.jpg"):
            assert isinstance(o, libvips_amd.VipsHipError) and "unable to open" in str(o)
            continue
        want = Image.thumbnail(p, 96, 96, crop="centre").numpy()
        assert np.array_equal(o.numpy(), want), p


def test_batch_error_plumbing_without_a_device(tmp_path):
    """No GPU here: every file must fail loudly (there is no CPU path), each with its own message,
    from worker threads, without taking the process down."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from libvips_amd import Image, VipsHipError

    good = str(tmp_path / "g.jpg")
    make_jpeg(good, 80, 60)
    outs = Image.thumbnail_batch([good, str(tmp_path / "nope.jpg"), good], 32, threads=3)
    assert all(isinstance(o, VipsHipError) for o in outs)
    assert "no HIP device" in str(outs[0]) and "unable to open" in str(outs[1])


@pytest.mark.skipif(not helpers.have_module(), reason="oracle/_ref or host/_build missing")
def test_module_thumbnail_hip_without_a_device(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    Ref.load_module()
    path = str(tmp_path / "t.jpg")
    make_jpeg(path, 200, 150)
    with pytest.raises(RuntimeError, match="no HIP device"):
        Ref.create("thumbnail_hip", "filename=%s,width=64" % path)


@pytest.mark.gpu
@needs_ref_jpeg
@pytest.mark.skipif(not helpers.have_module(), reason="oracle/_ref or host/_build missing")
def test_module_thumbnail_hip_matches_vips_thumbnail(tmp_path):
    """`vips thumbnail_hip x.jpg ...` against the reference's own `thumbnail` operation."""
    Ref.load_module()
    path = str(tmp_path / "t.jpg")
    make_jpeg(path, 2400, 1600)
    for args in ("width=200", "width=128,height=128,crop=centre", "width=300,height=100,size=force"):
        got, _, _ = Ref.create("thumbnail_hip", "filename=%s,%s" % (path, args))
        want, _, _ = Ref.create("thumbnail", "filename=%s,%s" % (path, args))
        assert got.shape == want.shape and np.array_equal(got, want), args


@needs_ref_jpeg
def test_jpeg_crop_pipeline_through_the_port_matches_vips_thumbnail(tmp_path):
    from libvips_amd import _ffi
    from tests.helpers import PortCC

    path = str(tmp_path / "t.jpg")
    make_jpeg(path, 2400, 1600)
    for (tw, th, size, crop, code) in ((128, 128, "both", "centre", 1), (300, 100, "force", "none", 0),
                                       (100, 300, "both", "high", 5), (200, 200, "down", "low", 4)):
        args = "width=%d,height=%d,size=%s,crop=%s" % (tw, th, size, crop)
        want, _, _ = Ref.create("thumbnail", "filename=%s,%s" % (path, args))
        mode = {"both": 0, "up": 1, "down": 2, "force": 3}[size]
        factor = _ffi.lib.vips_hip_thumbnail_find_jpegshrink(2400, 1600, tw, th, mode, 0, code)
        pre, _ = product_decode(path, factor)
        got = PortCC.thumbnail_image(pre, "srgb", tw, th, size=size, crop=crop)
        assert got.shape == want.shape and np.array_equal(got, want), args


@needs_ref_jpeg
def test_embedded_icc_does_not_change_the_reference_pixels(tmp_path):
    """Why vips_hip_thumbnail only refuses ICC files in linear mode: with no export profile the
    reference carries the profile as metadata and thumbnails the same pixels."""
    a, b = str(tmp_path / "a.jpg"), str(tmp_path / "b.jpg")
    make_jpeg(a, 800, 600)
    make_jpeg(b, 800, 600, icc_profile=b"\0" * 300)
    assert product_header(b).has_icc == 1
    out_a, _ = cli_thumbnail(tmp_path, a, "100x100")
    out_b, _ = cli_thumbnail(tmp_path, b, "100x100")
    assert np.array_equal(out_a, out_b)
