#!/usr/bin/env python3
"""Launch order of the last `n` kernels of a rocprofv3 rocpd database (short names, grid, start gap to the previous kernel's end).

usage: python scripts/prof_sequence.py <results.db> [n = 300] [name filter: only print launches whose neighbourhood (+-2) contains it]
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
flt = sys.argv[3] if len(sys.argv) > 3 else None
rows = db.execute("select name, grid_x / workgroup_x, grid_y / workgroup_y, start, end from kernels order by start").fetchall()[-n:]
short = [re.sub(r"\(.*", "", r[0].replace("(anonymous namespace)::", "").replace("void ", ""))[:48] for r in rows]
for i, r in enumerate(rows):
    if flt and not any(flt in short[j] for j in range(max(0, i - 2), min(len(rows), i + 3))):
        continue
    gap = (r[3] - rows[i - 1][4]) / 1e3 if i else 0.0
    print("%4d %-48s (%d,%d) %7.1f us  gap %6.1f" % (i, short[i], r[1], r[2], (r[4] - r[3]) / 1e3, gap))
