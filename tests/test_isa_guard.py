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
    from concurrent.futures import ThreadPoolExecutor

    def one(obj):
        tmp = os.path.join(str(tmp_path), os.path.basename(obj) + ".d")
        os.makedirs(tmp)
        return _disassemble(obj, tmp)

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as pool:
        texts = list(pool.map(one, objs))
    bad = {}
    kernels = 0
    for obj, text in zip(objs, texts):
        kernels += len(re.findall(r"^[0-9a-f]+ <_Z\w+>:", text, flags=re.M))
        n = text.count("v_ashr_pk_u8_i32")
        if n:
            bad[os.path.basename(obj)] = n
    assert kernels > 100, kernels
    assert not bad, "v_ashr_pk_u8_i32 selected (its upper half is not what the compiler assumes): %r" % bad


CSRC = os.path.join(helpers.ROOT, "libvips_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _device_ir(src):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O0", "-std=c++17", "-ffp-contract=off",
                          "-I" + os.path.join(helpers.ROOT, "include"), "-I" + CSRC, "-x", "hip",
                          "--cuda-device-only", "-emit-llvm", "-S", src, "-o", "-"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=True).stdout
    return src, out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_c_cast_of_a_float_to_an_integer_in_device_code():
    """A C cast of NaN or of an out-of-range float to an integer type is undefined; the kernels need what the
    part's converters do (saturate, NaN -> 0), so every such conversion is spelled as the instruction
    (kernel_stmt.h: vh::cvt_i32 / cvt_u32 / cvt_to).  The unoptimised device IR of every kernel file must hold
    no fptosi / fptoui -- neither from our sources nor from a HIP header's helper (__float2int_rz is a C cast)."""
    from concurrent.futures import ThreadPoolExecutor

    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    assert len(srcs) >= 15, srcs
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as pool:
        results = list(pool.map(_device_ir, srcs))
    bad = {}
    defined = 0
    for src, ir in results:
        defined += len(re.findall(r"^define ", ir, flags=re.M))
        hits = re.findall(r"^.*\b(?:fptosi|fptoui)\b.*$", ir, flags=re.M)
        if hits:
            bad[os.path.basename(src)] = hits[:3]
    assert defined > 300, defined
    assert not bad, "C float -> int casts in device code: %r" % bad


# Kernels that are ALLOWED a few bytes of scratch (their register budget is fixed by the launch bounds their LDS
# layout needs; profiles/r05_kernel_resources.txt): anything else with a private segment, or any of these past
# 64 bytes, is a spill nobody chose.
SCRATCH_ALLOWED = ("vh::conv_u8_mfma_sep<4, false, 0>", "vh::conv_u8_mfma_sep<4, true, 0>",
                   "vh::convsep_stream<1, 8, 1, 768, 0>", "vh::convsep_stream<1, 8, 2, 768, 0>",
                   "vh::convsep_stream<1, 8, 2, 768, 1>", "vh::reduce_fused_u8x4<8, 7>", "vh::resize_sharpen_u8<8, 7>")


@pytest.mark.skipif(not os.path.exists(LLVM + "/llvm-readelf"), reason="no llvm-readelf")
def test_no_kernel_spills_to_scratch():
    """The code objects' metadata (tools/kernel_resources.py): every kernel of the built library keeps its registers
    -- no private segment -- but the seven listed, and BASELINE configs[1]'s kernel runs four waves a SIMD."""
    import importlib.util

    from libvips_amd import _ffi

    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(helpers.ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    ks = kr.kernels_of(_ffi.LIB_PATH)
    assert len(ks) > 500, len(ks)
    bad = {k["demangled"]: k["private_segment_fixed_size"] for k in ks
           if k["private_segment_fixed_size"] > (64 if k["demangled"] in SCRATCH_ALLOWED else 0)}
    assert not bad, "kernels with scratch: %r" % bad
    c2 = [k for k in ks if k["demangled"].startswith("vh::reduce_fused_u8x4_mfma<6, 1, 4, true, 0, true, 256,")]
    assert c2 and all(kr.waves_per_simd(k) == 4 and k["vgpr_spill_count"] == 0 for k in c2), c2
