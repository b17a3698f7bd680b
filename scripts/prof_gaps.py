#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 rocpd database, by the kernel that FOLLOWS the gap.

usage: python scripts/prof_gaps.py <results.db> <iterations> [skip first N launches]
"""
import collections
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    iters = float(sys.argv[2])
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = db.execute("select name, start, end from kernels order by start").fetchall()[skip:]
    by = collections.defaultdict(lambda: [0, 0.0])
    tot = 0.0
    prev_end = rows[0][1]
    for name, s, e in rows:
        g = (s - prev_end) / 1e3
        if 0.5 < g < 2000.0:
            n = re.sub(r"\(.*", "", name)
            by[n][0] += 1
            by[n][1] += g
            tot += g
        prev_end = max(prev_end, e)
    span = (rows[-1][2] - rows[0][1]) / 1e3
    print("span %.1f us (%.1f us / iteration), gaps 0.5 us .. 2 ms: %.1f us / iteration" % (span, span / iters, tot / iters))
    for n, (c, g) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
        print("%-70s %6.1f gaps/it %8.1f us/it  avg %6.1f us" % (n[:70], c / iters, g / iters, g / c))


if __name__ == "__main__":
    main()
