# per kernel of a short bench run: grid, LDS, registers -> how many workgroups fit a CU (diagnostic)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/occ_kt -o kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/occ.err
DB=$(find $R/gpurun_out/occ_kt -name "*.db" | head -1)
python3 - "$DB" > $R/gpurun_out/occupancy.txt <<'PY'
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, end-start from kernels").fetchall()
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = re.sub(r"\(.*", "", r[0].replace("(anonymous namespace)::", "")); n = re.sub(r"^void ", "", n)
    wgs = (r[1] // max(r[4], 1)) * max(r[2], 1) * max(r[3], 1)
    key = (n[:60], wgs, r[4], r[5], r[6], r[7])
    agg[key][0] += 1; agg[key][1] += r[9]
out = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("%-60s %7s %5s %7s %5s %5s %6s %9s %8s  wg/CU(lds) waves/SIMD(regs)" % ("kernel", "wgs", "thr", "lds", "vgpr", "agpr", "calls", "avg us", "tot ms"))
for (n, wgs, thr, lds, v, a), (c, t) in out[:60]:
    per_lds = 160 * 1024 // lds if lds else 99
    regs = v + a
    wps = 512 // regs if regs else 8
    print("%-60s %7d %5d %7d %5d %5d %6d %9.1f %8.2f  %d  %d" % (n, wgs, thr, lds, v, a, c, t / c / 1e3, t / 1e6, min(per_lds, 16), min(wps, 8)))
PY
rm -rf $R/gpurun_out/occ_kt
