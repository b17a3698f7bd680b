cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/kt -o kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
DB=$(find $R/gpurun_out/kt -name "*.db" | head -1)
python3 $R/scripts/prof_summary.py $DB 24 --csv $R/gpurun_out/kstats_now.csv > $R/gpurun_out/kstats_now.txt
python3 $R/scripts/prof_summary.py $DB 24 --by-grid > $R/gpurun_out/kstats_now_by_grid.txt 2>&1
python3 -c "import sqlite3,sys; db=sqlite3.connect(sys.argv[1]); print([r[1] for r in db.execute(\"pragma table_info(kernels)\")])" $DB > $R/gpurun_out/kernels_schema.txt 2>&1
rm -rf $R/gpurun_out/kt
head -40 $R/gpurun_out/kstats_now_by_grid.txt; cat $R/gpurun_out/kernels_schema.txt
