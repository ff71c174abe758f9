"""CPU: the fiber emulator against itself (tests/emul/emul_selftest.cpp): a wave vote only counts the lanes that
make it (a lane that goes to the barrier instead is a masked-off lane), two votes in a row, waves that do not
vote; the quad DPP move; v_mfma_f32_4x4x4_16b_f16 against its definition."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CXX = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CXX), reason="no clang++")
def test_emulator_semantics(tmp_path):
    exe = str(tmp_path / "emul_selftest")
    emul = os.path.join(HERE, "emul")
    cmd = [CXX, "-std=c++17", "-O2", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I" + emul,
           "-I" + os.path.join(ROOT, "libvips_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
           "-Wall", "-Wno-unused-function", os.path.join(emul, "emul_selftest.cpp"), os.path.join(emul, "emul.cpp"),
           "-o", exe, "-lpthread"]
    subprocess.run(cmd, check=True, timeout=600)
    out = subprocess.run([exe], check=True, timeout=300, capture_output=True, text=True).stdout
    assert out.strip() == "OK", out
