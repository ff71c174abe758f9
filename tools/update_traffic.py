#!/usr/bin/env python3
"""Stamp profiles/traffic.json with HBM bytes per launch of the C2 kernel from a rocprofv3 PMC
summary (profiles/rocprof_summary.py output) and the hash of the kernel's machine code in the
built library (bench.kernel_isa_sha): comment / rename edits cannot make the counters look stale.

usage: python tools/update_traffic.py profiles/r03_c2_pmc.txt
Reads the FETCH_SIZE / WRITE_SIZE (KB) rows of the reduce_fused_u8x4_mfma kernel; traffic =
FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md section HBM) + WRITE_SIZE.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_isa_sha  # noqa: E402


def main():
    path = sys.argv[1]
    fetch = write = None
    text = open(path).read().splitlines()
    # round 6: BASELINE config 2 runs the kernel without horizontal halos (reduce_fused_u8x4_mfma_x); a summary
    # that holds its rows is stamped for it, an older one for the kernel with halos
    exchange = any("reduce_fused_u8x4_mfma_x" in line for line in text)
    want = "reduce_fused_u8x4_mfma_x" if exchange else "reduce_fused_u8x4_mfma<"
    for line in text:
        if want not in line:
            continue
        m = re.search(r"\b(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([0-9.]+)", line)
        if m:
            kb = float(m.group(3))
            if m.group(1) == "FETCH_SIZE":
                fetch = kb
            else:
                write = kb
    if fetch is None or write is None:
        raise SystemExit("no FETCH_SIZE / WRITE_SIZE rows for reduce_fused_u8x4_mfma in %s" % path)
    src = "libvips_amd/csrc/reduce_u8.hip"
    # the instantiation C2 launches: reduce_fused_u8x4_mfma_x<6, 4, 2> (before round 6:
    # reduce_fused_u8x4_mfma<6, 1, 4, true, 0, true, 256, 1>)
    symbol = "reduce_fused_u8x4_mfma_xILi6ELi4ELi2EE" if exchange else "reduce_fused_u8x4_mfmaILi6ELi1ELi4ELb1ELi0ELb1ELi256ELi1EE"
    sha = kernel_isa_sha(symbol)
    if sha is None:
        raise SystemExit("libvipship.so holds no gfx950 function named *%s*" % symbol)
    fetch_b = int(round(fetch * 2 * 1024))
    write_b = int(round(write * 1024))
    key = "reduce_fused_u8_mfma_x" if exchange else "reduce_fused_u8_mfma"
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (IOError, ValueError):
        table = {}
    table.update({
        "_comment": "HBM bytes per launch from rocprofv3 PMC passes: FETCH_SIZE x 2 (gfx950 correction, "
                    "MI355X_MICROARCH.md section HBM) + WRITE_SIZE.  bench.py reports an entry only while "
                    "isa_sha equals the hash of the kernel's machine code in libvipship.so "
                    "(bench.kernel_isa_sha).",
        key: {
            "traffic_bytes": fetch_b + write_b,
            "fetch_bytes_x2": fetch_b,
            "write_bytes": write_b,
            "algorithmic_bytes": 16384 * 16384 * 4 + 2048 * 2048 * 4,
            "source": src,
            "symbol": symbol,
            "isa_sha": sha,
            "profile": os.path.relpath(os.path.abspath(path), ROOT),
        },
    })
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
        json.dump(table, f, indent=1)
        f.write("\n")
    print(json.dumps(table[key], indent=1))


if __name__ == "__main__":
    main()
