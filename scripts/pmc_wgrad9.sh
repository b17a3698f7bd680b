#!/bin/bash
# LDS / MFMA counters of the nine-tap filter-gradient kernel on the 256-channel 32x32 layer (one rocprofv3 --pmc pass per counter group)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export WGRAD_SETS="${WGRAD_SETS:-32x32 alone}"
i=0
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_w9 -o p -- python3 $R/scripts/bench_wgrad_group.py 64 > /dev/null 2>&1
DB=$(find $R/gpurun_out/kt_w9 -name "*.db" | head -1)
python3 $R/scripts/prof_summary.py $DB 1 2>&1 | head -12
rm -rf $R/gpurun_out/kt_w9
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pmc_w9_$i -o p -- python3 $R/scripts/bench_wgrad_group.py 64 > /dev/null 2>&1
  DB=$(find $R/gpurun_out/pmc_w9_$i -name "*.db" | head -1)
  python3 $R/scripts/pmc_summary.py wgrad9 $DB 2>&1 | head -40
  rm -rf $R/gpurun_out/pmc_w9_$i
done
