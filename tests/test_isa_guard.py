"""CPU: a guard on the compiled device code.  For "shift, clamp to 0..255, pack" hipcc (ROCm 7.2,
gfx950) selects v_ashr_pk_u8_i32 and then takes bits 31:16 of its result for zero, which the
hardware does not clear (profiles/NOTES.md 3.1 "toolchain findings"; round 4's packed reduceh was
bit-exact on host fibers and wrong on the device for exactly this reason).  Kernels keep the shift
and the clamp apart with an empty asm; this test disassembles every built kernel object and fails
if the instruction comes back."""
import glob
import os
import re
import subprocess

import pytest

from tests import helpers

OBJ = os.path.join(helpers.ROOT, "libvips_amd", "csrc", "_obj")
LLVM = "/opt/rocm/lib/llvm/bin"


def _disassemble(obj, tmp):
    fat = os.path.join(tmp, "fat.bin")
    elf = os.path.join(tmp, "dev.elf")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    if not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return ""
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + elf], check=True)
    return subprocess.run([LLVM + "/llvm-objdump", "-d", elf], stdout=subprocess.PIPE, text=True, check=True).stdout


@pytest.mark.skipif(not os.path.isdir(OBJ) or not os.path.exists(LLVM + "/llvm-objdump"),
                    reason="no built kernel objects / no llvm-objdump")
def test_no_packed_shift_clamp_in_device_code(tmp_path):
    objs = sorted(glob.glob(os.path.join(OBJ, "*.hip.o")))
    assert len(objs) >= 15, objs
    bad = {}
    kernels = 0
    for obj in objs:
        text = _disassemble(obj, str(tmp_path))
        kernels += len(re.findall(r"^[0-9a-f]+ <_Z\w+>:", text, flags=re.M))
        n = text.count("v_ashr_pk_u8_i32")
        if n:
            bad[os.path.basename(obj)] = n
    assert kernels > 100, kernels
    assert not bad, "v_ashr_pk_u8_i32 selected (its upper half is not what the compiler assumes): %r" % bad
