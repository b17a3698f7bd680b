#!/usr/bin/env python3
"""Per-(kernel, grid) averages of the hardware counters in rocprofv3 --pmc rocpd databases.

usage: python scripts/pmc_summary.py <name-filter> <results.db> [<results.db> ...]
"""
import re
import sqlite3
import sys

flt = "%" + sys.argv[1] + "%"
for path in sys.argv[2:]:
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, grid_size_x/workgroup_size_x, grid_size_y/workgroup_size_y, grid_size_z/workgroup_size_z, "
                      "counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by 1,2,3,4,5 order by 1,2,3,4,5",
                      (flt,)).fetchall()
    last = None
    for r in rows:
        key = (re.sub(r"^void ", "", re.sub(r"\(.*", "", r[0].replace("(anonymous namespace)::", "")))[:60], r[1], r[2], r[3])
        if key != last:
            print("%s grid %s (%d dispatches)" % (key[0], key[1:], r[6]))
            last = key
        print("    %-32s %16.1f" % (r[4], r[5]))
