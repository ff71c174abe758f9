#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in the built libvipship.so, read from the code objects' metadata
(no GPU needed): llvm-objdump --offloading unbundles the gfx950 objects, llvm-readelf --notes prints the
amdhsa.kernels records.  One line per kernel: vgpr / agpr / sgpr, LDS bytes (static), scratch bytes, spilled
registers, the workgroup size the kernel was compiled for, and waves per SIMD the registers allow (512 VGPRs a
SIMD lane on gfx950, granule 8; at most 8).

usage:  python tools/kernel_resources.py [libvipship.so] > profiles/<tag>_kernel_resources.txt
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
          "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size")


def kernels_of(so):
    """-> [{name, demangled, vgpr_count, ...}] for every kernel of every gfx950 code object in `so`."""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(so, os.path.join(tmp, "lib.so"))
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", "lib.so"], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for obj in sorted(glob.glob(os.path.join(tmp, "lib.so.*gfx950"))):
            notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", obj], check=True, stdout=subprocess.PIPE,
                                   text=True).stdout
            # one YAML record per kernel, each starting at "  - .agpr_count:" (keys are sorted)
            for rec in re.split(r"^  - (?=\.)", notes, flags=re.M)[1:]:
                if ".vgpr_count:" not in rec or ".name:" not in rec:
                    continue
                k = {}
                for f in FIELDS:
                    m = re.search(r"^\s*\.%s:\s*(\d+)" % f, rec, flags=re.M)
                    k[f] = int(m.group(1)) if m else 0
                k["name"] = re.search(r"^\s*\.name:\s*(\S+)", rec, flags=re.M).group(1)
                out.append(k)
    names = [k["name"] for k in out]
    dem = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for k, d in zip(out, dem):
        k["demangled"] = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", d))
    return out


def waves_per_simd(k):
    regs = k["vgpr_count"] + k["agpr_count"]  # (unified file: the accumulation registers count)
    regs = max(8, (regs + 7) // 8 * 8)
    return min(8, 512 // regs)


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "libvips_amd", "lib", "libvipship.so")
    ks = sorted(kernels_of(so), key=lambda k: k["demangled"])
    print("# %d kernels in %s" % (len(ks), os.path.relpath(so, ROOT)))
    print("# vgpr agpr sgpr   lds_B scratch_B spills  wg  waves/SIMD  kernel")
    for k in ks:
        print("%5d %4d %4d %7d %9d %6d %4d %6d      %s" % (
            k["vgpr_count"], k["agpr_count"], k["sgpr_count"], k["group_segment_fixed_size"],
            k["private_segment_fixed_size"], k["vgpr_spill_count"] + k["sgpr_spill_count"],
            k["max_flat_workgroup_size"], waves_per_simd(k), k["demangled"]))
    spill = [k for k in ks if k["private_segment_fixed_size"] or k["vgpr_spill_count"] or k["sgpr_spill_count"]]
    print("# kernels with scratch or spills: %d" % len(spill))
    for k in spill:
        print("#   %s: scratch %d B, %d vgpr + %d sgpr spilled" % (k["demangled"], k["private_segment_fixed_size"],
                                                                  k["vgpr_spill_count"], k["sgpr_spill_count"]))


if __name__ == "__main__":
    main()
