#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (.db) outputs into the text files committed under profiles/.

usage: rocprof_summary.py <kernel-trace.db> [<pmc.db> ...]
Kernel trace: per-kernel calls / total / average duration (the `--stats` view).
PMC passes: per-kernel average counter values; FETCH_SIZE/WRITE_SIZE are in KB and
FETCH_SIZE is additionally shown x2 (MI355X_MICROARCH.md section HBM: on gfx950 this
rocprofv3 reports exactly half of the bytes of a wide coalesced streaming read).
"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        print("== %s" % path)
        n = db.execute("select count(*) from counters_collection").fetchone()[0]
        if n == 0:
            print("%-70s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
            for name, calls, total, avg, pct in db.execute(
                    "select name, total_calls, total_duration, average, percentage from top_kernels"):
                print("%-70s %8d %14.3f %12.3f %6.2f%%" % (name[:70], calls, total, avg, pct))
        else:
            print("%-60s %-12s %6s %16s" % ("kernel", "counter", "n", "avg"))
            for name, counter, cnt, avg in db.execute(
                    "select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                    "group by kernel_name, counter_name"):
                extra = ""
                if counter == "FETCH_SIZE":
                    extra = "  KB  (x2 gfx950 correction = %.1f MB)" % (avg * 2 * 1024 / 1e6)
                elif counter == "WRITE_SIZE":
                    extra = "  KB  (= %.1f MB)" % (avg * 1024 / 1e6)
                print("%-60s %-12s %6d %16.3f%s" % (name[:60], counter, cnt, avg, extra))
        print()


if __name__ == "__main__":
    main()
