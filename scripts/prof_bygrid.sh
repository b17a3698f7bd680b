#!/bin/bash
# usage (through gpurun, from the repo root): bash scripts/prof_bygrid.sh <tag> [env assignments...]
# kernel trace of the default bench workload -> per-(kernel, grid) time table, per-kernel table, launch sequence of the last iteration
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt
DB=$(find $OUT/kt -name "*.db" | head -1)
python3 $ROOT/scripts/prof_summary.py $DB 14 --csv $OUT/stats.csv > $OUT/stats.txt
python3 $ROOT/scripts/prof_summary.py $DB 14 --by-grid > $OUT/by_grid.txt
python3 $ROOT/scripts/prof_sequence.py $DB 560 > $OUT/seq.txt 2>&1
rm -rf $OUT/kt
head -60 $OUT/by_grid.txt
