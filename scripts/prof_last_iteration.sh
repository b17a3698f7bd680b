cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/seq_kt -o kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/seq_bench.json 2> $R/gpurun_out/seq.err
DB=$(find $R/gpurun_out/seq_kt -name "*.db" | head -1)
python3 $R/scripts/prof_sequence.py $DB 269 > $R/gpurun_out/seq.txt
python3 - "$DB" > $R/gpurun_out/seq_grids.txt <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print(cols)
gc = [c for c in cols if 'grid' in c.lower() or 'workgroup' in c.lower()]
rows = db.execute("select name, start, end, %s from kernels order by start" % ",".join(gc)).fetchall()[-269:]
for i, r in enumerate(rows):
    n = re.sub(r"\(.*", "", r[0].replace("(anonymous namespace)::", "")); n = re.sub(r"^void ", "", n)
    print(i, "%.1f" % ((r[2]-r[1])/1e3), n[:70], r[3:])
PY
rm -rf $R/gpurun_out/seq_kt
