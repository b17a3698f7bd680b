#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (the default output format of ROCm 7.2's rocprofv3).

usage: python scripts/prof_summary.py <results.db> [iterations] [--csv out.csv] [--by-grid]
--by-grid: one row per (kernel, grid) instead of per kernel (which launches of a kernel the time goes to).
"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    iters = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 1.0
    if "--by-grid" in sys.argv:
        # rocpd reports the grid in work-items: / workgroup size = workgroups
        rows = db.execute("select name || ' grid (' || (grid_x / workgroup_x) || ',' || (grid_y / workgroup_y) || ')', count(*), sum(end-start), "
                          "avg(end-start), min(end-start), max(end-start) from kernels group by 1 order by 3 desc").fetchall()
    else:
        rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                          "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    out = None
    if "--csv" in sys.argv:
        out = open(sys.argv[sys.argv.index("--csv") + 1], "w")
        out.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    print("total kernel time %.3f ms (%.3f ms / iteration), %d launches / iteration" %
          (tot / 1e6, tot / 1e6 / iters, sum(r[1] for r in rows) / iters))
    for r in rows:
        n = r[0].replace("(anonymous namespace)::", "")
        n = re.sub(r"\(MfmaConvArgs\)|\(MfmaWgradArgs[^)]*\)|\([A-Za-z_][^)]*\)(?= grid|$)", "", n) if "--by-grid" in sys.argv else re.sub(r"\(.*", "", n)
        if out:
            out.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (n.replace("void ", "") if "--by-grid" in sys.argv else n, r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5]))
        print("%-72s %7.1f /it %9.3f ms/it %8.1f us %5.1f%%" % (n[:72], r[1] / iters, r[2] / 1e6 / iters, r[3] / 1e3, 100 * r[2] / tot))


if __name__ == "__main__":
    main()
