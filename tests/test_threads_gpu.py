"""Several host threads, each with its own library stream (the libvips worker model, one image
per worker; SURVEY.md 8(b) "Threading"): results must equal the single-threaded ones.  Guards
the pool (a block freed on one stream must not be reused on another while kernels are in
flight) and the shared operation caches."""
import threading

import numpy as np
import pytest

import libvips_amd
from libvips_amd import Image, lib
from tests import helpers

pytestmark = pytest.mark.gpu


def pipeline(src):
    im = Image.new_from_array(src, interpretation="srgb")
    a = im.resize(0.25).sharpen()
    b = im.gaussblur(2.0).colourspace("lab").colourspace("srgb")
    c = im.reduce(8, 8)
    return a.numpy(), b.numpy(), c.numpy()


def test_threads_match_single_thread():
    libvips_amd.init(0)
    srcs = [helpers.lcg_image(640 + 16 * i, 512 + 8 * i, 3 + (i & 1), np.uint8, 90 + i) for i in range(6)]
    want = [pipeline(s) for s in srcs]
    errors = []

    def worker(k):
        try:
            libvips_amd.init(0)
            lib.vips_hip_set_stream(None)  # this thread's own stream
            for rep in range(12):
                got = pipeline(srcs[k])
                for g, w in zip(got, want[k]):
                    if not np.array_equal(g, w):
                        errors.append((k, rep))
                        return
        except Exception as exc:  # noqa: BLE001
            errors.append((k, repr(exc)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(srcs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
