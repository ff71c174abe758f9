#!/usr/bin/env python
"""Per-dispatch timeline out of a rocprofv3 --kernel-trace rocpd database: when every launch of
the kernels matching a pattern started (ms after the first of them), how long it ran and the gap
to the previous launch of the same kernel -- where a batch loses time BETWEEN its kernels.

usage: kernel_timeline.py <kernel-trace.db> <substring>[,<substring>...] [last N launches]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pats = sys.argv[2].split(",")
    last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    if not cols:
        print("no `kernels` view; objects:", [r[0] for r in db.execute("select name from sqlite_master")])
        return
    want = {"name": None, "start": None, "end": None}
    for c in cols:
        lc = c.lower()
        if lc in ("name", "kernel_name") and want["name"] is None:
            want["name"] = c
        elif lc in ("start", "start_timestamp") and want["start"] is None:
            want["start"] = c
        elif lc in ("end", "end_timestamp") and want["end"] is None:
            want["end"] = c
    if None in want.values():
        print("unexpected columns:", cols)
        return
    extra = [c for c in cols if c.lower() in ("stream_id", "queue_id", "stream")]
    q = "select %s, %s, %s%s from kernels order by %s" % (
        want["name"], want["start"], want["end"], "".join(", " + c for c in extra), want["start"])
    rows = [r for r in db.execute(q) if any(p in r[0] for p in pats)]
    if last:
        rows = rows[-last:]
    if not rows:
        print("no launches match", pats)
        return
    t0 = rows[0][1]
    prev_end = {}
    print("%-34s %10s %10s %10s  %s" % ("kernel", "start_ms", "dur_ms", "gap_ms", " ".join(extra)))
    busy = {}
    for r in rows:
        name, s, e = r[0], r[1], r[2]
        short = name.split("(")[0][-34:]
        gap = (s - prev_end[short]) / 1e6 if short in prev_end else 0.0
        prev_end[short] = e
        busy[short] = busy.get(short, 0.0) + (e - s) / 1e6
        print("%-34s %10.3f %10.3f %10.3f  %s" % (short, (s - t0) / 1e6, (e - s) / 1e6, gap, " ".join(str(x) for x in r[3:])))
    print("span %.3f ms; busy per kernel: %s" % ((max(r[2] for r in rows) - t0) / 1e6,
                                                 {k: round(v, 3) for k, v in busy.items()}))


if __name__ == "__main__":
    main()
