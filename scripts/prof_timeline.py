#!/usr/bin/env python3
"""Kernel timeline of one training iteration from a rocprofv3 rocpd database: launches in start order with duration and
the gap to the previous kernel's end, grouped into the captured step graphs (a gap > 30 us starts a new group).

usage: python scripts/prof_timeline.py <results.db> [--iter K] [--full]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    full = "--full" in sys.argv
    # the bench's timed region = the longest run of launches without a > 2 ms pause; take the last 1/3 of the trace
    n = len(rows)
    rows = rows[n * 2 // 3:]
    groups, cur = [], [rows[0]]
    for prev, r in zip(rows[:-1], rows[1:]):
        if r[1] - prev[2] > 30000:
            groups.append(cur)
            cur = []
        cur.append(r)
    groups.append(cur)
    for gi, g in enumerate(groups[:40]):
        busy = sum(r[2] - r[1] for r in g)
        span = g[-1][2] - g[0][1]
        print("group %2d: %3d launches, span %8.1f us, kernel time %8.1f us, idle inside %6.1f us" % (gi, len(g), span / 1e3, busy / 1e3, (span - busy) / 1e3))
        if full:
            for prev, r in zip([None] + g[:-1], g):
                gap = (r[1] - prev[2]) / 1e3 if prev else 0.0
                print("      %-60s %8.1f us  gap %6.1f" % (re.sub(r"\(.*", "", r[0].replace("(anonymous namespace)::", ""))[:60], (r[2] - r[1]) / 1e3, gap))


if __name__ == "__main__":
    main()
