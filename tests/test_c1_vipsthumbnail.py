"""BASELINE config 1: `vipsthumbnail 4096x4096 sRGB uchar -> 512x512` on the reference CPU
path (plumbing; no GPU).  The reference's own CLI, built unchanged in oracle/_ref/bin, is run
on a .v file; the oracle port's thumbnail pipeline must give the same pixels, and -- on the
GPU box -- so must the HIP path."""
import os
import subprocess

import numpy as np
import pytest

from tests import helpers
from tests.helpers import PortCC

VIPSTHUMBNAIL = os.path.join(helpers.ROOT, "oracle", "_ref", "bin", "vipsthumbnail")
needs_cli = pytest.mark.skipif(not os.path.exists(VIPSTHUMBNAIL), reason="oracle/_ref/bin not built")


def run_cli(tmp_path, src, size):
    src_path = os.path.join(str(tmp_path), "src.v")
    out_path = os.path.join(str(tmp_path), "out.v")
    helpers.write_v(src_path, src, interpretation=22)
    env = dict(os.environ, VIPS_INFO="1")
    proc = subprocess.run([VIPSTHUMBNAIL, src_path, "--size", size, "-o", out_path],
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, text=True)
    assert proc.returncode == 0, proc.stdout
    out, interp = helpers.read_v(out_path)
    return out, proc.stdout


@needs_cli
def test_c1_cli_matches_port(tmp_path):
    src = helpers.lcg_image(4096, 4096, 3, np.uint8, 12345)
    out, log = run_cli(tmp_path, src, "512x512")
    assert out.shape == (512, 512, 3)
    # shrinkv 4, reducev 13-point, shrinkh 4, reduceh 13-point
    assert "shrinkv by 4" in log and "shrinkh by 4" in log and "13 point mask" in log
    want = PortCC.thumbnail_image(src, "srgb", 512, 512)
    assert np.array_equal(out, want)


@pytest.mark.gpu
@needs_cli
def test_c1_cli_matches_hip(tmp_path):
    import libvips_amd
    from libvips_amd import Image

    libvips_amd.init(0)
    src = helpers.lcg_image(4096, 4096, 3, np.uint8, 12345)
    out, _ = run_cli(tmp_path, src, "512x512")
    got = Image.new_from_array(src, interpretation="srgb").thumbnail_image(512, 512).numpy()
    assert np.array_equal(got, out)
