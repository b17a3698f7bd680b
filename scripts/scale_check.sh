#!/bin/bash
# First contact with a multi-GPU node (the development boxes have one GPU; no process with more than one rank has run yet).
#   bash scripts/scale_check.sh            # from the repo root, on a node with >= 2 MI355X
# 1. the two-process RCCL test (ranks fed different shards end with bit-identical weights);
# 2. bench.py --gpus N for N = 1, 2, 4, 8 (as many as the node has), every N a FRESH process tree -- bench.py starts its ranks as child
#    processes before it touches the GPU, nothing here re-executes a GPU-initialised process;
# 3. per N: ms per iteration, images/s, efficiency against N = 1, whether every rank ended with the same weight hash
#    (config.rank_weights_bit_identical), the measured all-reduce time per iteration (HIP events around every group) --
# 4. -- beside what the one-GPU link model predicts for that N (bench.py --dp-stub N under 40 us + 2(N-1)/N * bytes / 200 GB/s per
#    group: ASSUMPTIONS, profiles/r04_bench_dpstub8_model_f32.json); the point of this script is to replace them with measurements.
# Reference semantics being checked: cifar10/gan_resnet.py:183-192 (batch x N, iterations / N), :529-546 (contiguous split, per-tower batch
# statistics), :697, :786 (mean of the tower losses).
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${1:-gpurun_out/scale_check}
mkdir -p "$OUT"
NG=$(python -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $NG"
STEPS=${STEPS:-50}
if [ "$NG" -ge 2 ]; then
  timeout 900 python -m pytest tests/test_gpu_dp.py -q -x -k "two_ranks_over_rccl" 2>&1 | tail -3 | tee "$OUT/two_rank_test.txt"
else
  echo "one GPU: the two-rank RCCL test and the N > 1 bench lines are skipped; the link-model predictions below still run"
fi
for N in 1 2 4 8; do
  if [ "$N" -le "$NG" ]; then
    timeout 900 python bench.py --gpus $N --steps $STEPS --warmup 5 --no-cpu-baseline 2> "$OUT/bench_n$N.err" | tail -1 > "$OUT/bench_n$N.json"
    # (round 6) the two-bucket schedule overlapped with the backward pass, the SAME node: the A/B the link model cannot give
    if [ "$N" -gt 1 ]; then
      RCGAN_DP_OVERLAP=1 timeout 900 python bench.py --gpus $N --steps $STEPS --warmup 5 --no-cpu-baseline 2> "$OUT/bench_overlap_n$N.err" | tail -1 > "$OUT/bench_overlap_n$N.json"
    fi
  fi
  if [ "$N" -gt 1 ]; then
    timeout 600 python bench.py --gpus 1 --steps $STEPS --warmup 5 --no-cpu-baseline --dp-stub $N --dp-stub-gbps 200 --dp-stub-lat-us 40 \
      2> /dev/null | tail -1 > "$OUT/model_n$N.json"
  fi
done
python - "$OUT" <<'PY'
import json, os, sys
out = sys.argv[1]
def load(name):
    try:
        with open(os.path.join(out, name)) as f:
            return json.loads(f.read().strip().splitlines()[-1])
    except Exception:
        return None
base = load("bench_n1.json")
print("%-3s %10s %12s %8s %10s %16s %16s %18s %18s" % ("N", "ms/iter", "images/s", "eff", "weights==", "allreduce ms", "model ms/iter", "model allreduce ms", "overlapped ms/iter"))
for n in (1, 2, 4, 8):
    b, mo = load("bench_n%d.json" % n), load("model_n%d.json" % n)
    row = ["%-3d" % n]
    if b:
        eff = b["value"] / (n * base["value"]) if base else float("nan")
        c = b["config"]
        row += ["%10.3f" % b["ms_per_step"], "%12.0f" % b["value"], "%8.3f" % eff, "%10s" % c.get("rank_weights_bit_identical", "-"),
                "%16s" % c.get("allreduce_ms_per_iteration", "-")]
    else:
        row += ["%10s" % "-", "%12s" % "-", "%8s" % "-", "%10s" % "-", "%16s" % "-"]
    if mo:
        row += ["%16.3f" % mo["ms_per_step"], "%18s" % mo["config"].get("allreduce_ms_per_iteration", "-")]
    else:
        row += ["%16s" % "-", "%18s" % "-"]
    ov = load("bench_overlap_n%d.json" % n)
    row += ["%18.3f" % ov["ms_per_step"] if ov else "%18s" % "-"]
    print(" ".join(row))
PY
