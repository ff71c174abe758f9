"""CPU: the x87 extended format in integer arithmetic (libvips_amd/csrc/x80.h, what the device sums
double images with -- reduceh.cpp:196-213's long double path) against the host's own long double:
conversions, products, sums with cancellation, double results down to denormals, whole
reduce_sum-shaped chains."""
import os
import subprocess

from tests import helpers


def test_x80_matches_host_long_double(tmp_path):
    exe = os.path.join(str(tmp_path), "x80_check")
    src = os.path.join(helpers.ROOT, "tests", "x80", "x80_check.cpp")
    inc = os.path.join(helpers.ROOT, "libvips_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + inc, src, "-o", exe], check=True)
    proc = subprocess.run([exe], stdout=subprocess.PIPE, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout
    assert proc.stdout.split() == ["from_double", "0", "mul", "0", "add", "0", "to_double", "0", "chains", "0"]
