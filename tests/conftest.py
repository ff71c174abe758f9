import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the compiled reference in oracle/_ref")


# Test files whose cases are child processes of minutes each (see helpers.Background): each has a
# prestart(selected test names) that launches the children of the selected cases.
# (NOT tests/test_module_stream.py's ring runs: they count the strips a producer thread makes before its consumer
# goes away -- a race the producer loses on an idle machine and may win on a loaded one.)
PRESTART = ("test_emul_gpu_suite",)


def pytest_collection_finish(session):
    if hasattr(session.config, "workerinput") or session.config.option.collectonly:
        return  # (pytest-xdist workers start theirs when the test itself runs)
    selected = {}
    for item in session.items:
        mod = getattr(item, "module", None)
        name = getattr(mod, "__name__", "").rsplit(".", 1)[-1]
        if name in PRESTART and hasattr(mod, "prestart"):
            selected.setdefault(mod, []).append(item.name)
    for mod, names in selected.items():
        mod.prestart(names)


def pytest_sessionfinish(session, exitstatus):
    from tests import helpers

    helpers.Background.reap_all()


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the HIP library and the oracle port are built (no-op when fresh)."""
    import __graft_entry__

    __graft_entry__.build()


@pytest.fixture(autouse=True)
def _exact_float(_built):
    """The parity suite pins kernels bit for bit, so it runs the library in its exact float mode
    (every double sum a chain of separately rounded multiplies and adds, as the reference's C);
    the default mode -- fused multiply-adds in the large float convolutions, at most 1 ULP from
    the reference -- is covered by the tests that switch it back on themselves
    (test_conv_colour_gpu.py::test_c5_*_default_mode, bench.py's parity checks, smoke())."""
    from libvips_amd import lib

    before = lib.vips_hip_get_exact_float()
    lib.vips_hip_set_exact_float(1)
    yield
    lib.vips_hip_set_exact_float(before)


@pytest.fixture(params=["exact", "default"])
def float_mode(request, _exact_float):
    """Run a float-convolution parity test in both modes of the library: "exact" (bit for bit against
    the reference) and "default" (the shipped mode: fused multiply-adds, within 1 ULP -- the
    tolerance BASELINE.json's north_star grants float paths)."""
    from libvips_amd import lib

    lib.vips_hip_set_exact_float(1 if request.param == "exact" else 0)
    yield request.param
    lib.vips_hip_set_exact_float(1)
