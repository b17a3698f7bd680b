#!/bin/bash
# MFMA-pipe utilisation of the matrix-core kernels under a given environment, on the GPU box:
#   bash scripts/pmc_busy.sh <tag> [ENV=VALUE ...]      -> gpurun_out/<tag>_pmc_mfma_busy.txt
# (counters in their own pass, kernel trace only: no other trace domain beside --pmc)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS \
  --kernel-trace -d "$OUT/pmc_$TAG" -o pmc -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_$TAG.err"
DB=$(find "$OUT/pmc_$TAG" -name "*.db" | head -1)
python3 "$ROOT/scripts/pmc_summary.py" conv_ "$DB" > "$OUT/${TAG}_pmc_raw.txt"
python3 "$ROOT/scripts/pmc_busy_table.py" "$OUT/${TAG}_pmc_raw.txt" > "$OUT/${TAG}_pmc_mfma_busy.txt"
rm -rf "$OUT/pmc_$TAG"
grep -E "p8|h8|wgrad3_group|trunk|kernel  " "$OUT/${TAG}_pmc_mfma_busy.txt"
