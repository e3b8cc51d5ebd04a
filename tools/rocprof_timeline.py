#!/usr/bin/env python
"""Timeline of the LAST boosting round in a rocprofv3 (rocpd sqlite) kernel trace: one line per launch with the
start offset, duration and gap to the previous kernel (all in microseconds).

usage: python tools/rocprof_timeline.py <results.db> [first-kernel-substring]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else None
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, start, end from kernels order by start"))
    if first:
        starts = [i for i, r in enumerate(rows) if first in r[0]]
        if not starts:
            print("no kernel matching", first)
            return
        i0 = starts[-1]
    else:       # a round starts with its lambda kernels (whichever variants the data set uses) and k_max_reduce, then k_quantize
        qs = [i for i, r in enumerate(rows) if "k_max_reduce" in r[0] or "k_quantize" in r[0]]      # (the fused root pass has no k_quantize launch)
        if not qs:
            print("no k_max_reduce / k_quantize launch in the trace")
            return
        i0 = qs[-1]
        if "k_quantize" in rows[i0][0] and i0 > 0 and "k_max_reduce" in rows[i0 - 1][0]:
            i0 -= 1
        while i0 > 0 and any(x in rows[i0 - 1][0] for x in ("k_lambda_", "k_pair_terms", "k_mart_residual")):
            i0 -= 1
    t0 = rows[i0][1]
    prev_end = t0
    busy = 0
    for name, st, en in rows[i0:]:
        short = name.replace("rl::", "").replace("void ", "").split("(")[0]
        print("%10.1f %9.2f %7.2f  %s" % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, short))
        busy += en - st
        prev_end = en
    print("# span %.1f us, kernel time %.1f us, launches %d" % ((prev_end - t0) / 1e3, busy / 1e3, len(rows) - i0))


if __name__ == "__main__":
    main()
