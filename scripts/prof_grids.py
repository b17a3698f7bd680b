#!/usr/bin/env python3
"""Per-(kernel, grid) summary of a rocprofv3 rocpd database: python scripts/prof_grids.py <results.db> [iterations] [name-filter]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
iters = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
flt = "%" + (sys.argv[3] if len(sys.argv) > 3 else "") + "%"
rows = db.execute("select name, grid_x/workgroup_x, grid_y/workgroup_y, grid_z/workgroup_z, count(*), avg(end-start), sum(end-start) "
                  "from kernels where name like ? group by 1,2,3,4 order by 7 desc", (flt,)).fetchall()
tot = db.execute("select sum(end-start) from kernels").fetchone()[0]
print("total %.3f ms/it" % (tot / 1e6 / iters))
for r in rows[:40]:
    n = re.sub(r"^void ", "", re.sub(r"\(.*", "", r[0]))
    print("%-56s %-16s %6.1f/it %9.1f us %8.3f ms/it" % (n[:56], str(tuple(r[1:4])), r[4] / iters, r[5] / 1e3, r[6] / 1e6 / iters))
