#!/usr/bin/env python3
"""Launch sequence of the LAST iteration of a bench run from a rocprofv3 rocpd database: start offset, duration, gap to the
previous kernel's end, kernel name (an iteration starts at the rng_fill / first kernel after the longest gaps).

usage: python scripts/prof_sequence.py <results.db> <launches per iteration> [name filter]
"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    per = int(sys.argv[2])
    flt = sys.argv[3] if len(sys.argv) > 3 else None
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    rows = rows[-per:]
    t0 = rows[0][1]
    prev_end = t0
    for i, (name, s, e) in enumerate(rows):
        n = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))
        n = re.sub(r"^void ", "", n)
        line = "%4d %9.1f us  dur %7.1f  gap %6.1f  %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n[:90])
        if flt is None or flt in n:
            print(line)
        prev_end = e


if __name__ == "__main__":
    main()
