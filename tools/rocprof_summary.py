#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace as a small text table for profiles/.

usage: python tools/rocprof_summary.py <results.db> [title] > profiles/rNN_xxx_kernel_stats.txt
Equivalent to `rocprofv3 --kernel-trace --stats`' kernel_stats, computed from the same trace.
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else db
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    print("# %s" % title)
    print("# source: rocprofv3 --kernel-trace --stats ; durations in microseconds")
    print("%-78s %8s %14s %12s %12s %12s %7s" % ("Name", "Calls", "TotalDur(us)", "AvgDur(us)", "MinDur(us)", "MaxDur(us)", "Pct"))
    for n, c, t, a, mn, mx in rows:
        print("%-78s %8d %14.2f %12.2f %12.2f %12.2f %6.2f%%" % (n[:78], c, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
    print("# total kernel time %.3f ms" % (tot / 1e6))


if __name__ == "__main__":
    main()
