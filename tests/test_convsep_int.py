"""CPU: the integer horizontal pass of convsep_stream (libvips_amd/csrc/convsep_int_body.h -- packed
bytes, v_dot4_u32_u8, the division by the scale in three single-precision operations) compiled for
the host against tests/emul/gcn.h and compared with the reference's arithmetic (double sum in mask
order, double division, cast: convolution/convi.c:721-741) on 76 000 windows; windows holding
anything but the integers 0 .. 255 and masks outside the bounds must be refused."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CXX = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CXX), reason="no clang++")
def test_integer_hpass_on_the_host(tmp_path):
    exe = str(tmp_path / "convsep_int_check")
    cmd = [CXX, "-std=c++17", "-O2", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(HERE, "emul"),
           "-I" + os.path.join(ROOT, "libvips_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
           "-Wall", "-Wno-unused-function", os.path.join(HERE, "emul", "convsep_int_check.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, timeout=300)
    out = subprocess.run([exe], check=True, timeout=300, capture_output=True, text=True).stdout
    assert out.startswith("OK "), out
    assert int(out.split()[1]) > 70000
