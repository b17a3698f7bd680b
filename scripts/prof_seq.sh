#!/bin/bash
# launch order around the device-side copies of one bench iteration (scripts/prof_sequence.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace -d $R/gpurun_out/kts -o kt -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
DB=$(find $R/gpurun_out/kts -name "*.db" | head -1)
python3 $R/scripts/prof_sequence.py $DB 520 copyBuffer > $R/gpurun_out/seq_copy.txt
rm -rf $R/gpurun_out/kts
