#!/bin/bash
# Collect the round's measurement artefacts on the GPU box (run through gpurun from the repo root):
#   bash scripts/make_profiles.sh r02
# Writes text / csv / json summaries under gpurun_out/<tag>/ (the rocpd databases themselves are deleted); the ones to be judged
# are then copied into profiles/.
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"

# the hash of the kernel sources THIS run measures (bench.py attaches the committed counters to a build by it): recorded here, at
# measurement time -- scripts/publish_profiles.sh refuses to publish counters whose hash is not the tree's
( cd "$ROOT" && python3 -c "import rcgan_amd; from rcgan_amd import _lib; print(_lib.source_hash())" ) > "$OUT/source_sha16.txt"
python3 "$B" > "$OUT/bench_n1_default.json" 2> "$OUT/bench_n1_default.err"

rocprofv3 --kernel-trace -d "$OUT/prof_kt" -o kt -- python3 "$B" --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_n1_under_rocprof.json" 2> "$OUT/prof_kt.err"
DB=$(find "$OUT/prof_kt" -name "*.db" | head -1)
python3 "$ROOT/scripts/prof_summary.py" "$DB" 24 --csv "$OUT/bench_n1_kernel_stats.csv" > "$OUT/bench_n1_kernel_stats.txt"
python3 "$ROOT/scripts/prof_summary.py" "$DB" 24 --by-grid --csv "$OUT/bench_n1_kernel_stats_by_grid.csv" > "$OUT/bench_n1_kernel_stats_by_grid.txt"
python3 "$ROOT/scripts/prof_sequence.py" "$DB" 300 > "$OUT/bench_n1_last_iteration_sequence.txt"
python3 "$ROOT/scripts/prof_timeline.py" "$DB" > "$OUT/bench_n1_timeline_groups.txt"
rm -rf "$OUT/prof_kt"

for CTR in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CTR --kernel-trace -d "$OUT/pmc_$CTR" -o pmc -- python3 "$B" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_$CTR.err"
  DB=$(find "$OUT/pmc_$CTR" -name "*.db" | head -1)
  python3 "$ROOT/scripts/pmc_summary.py" conv_mfma_h8_kernel "$DB" > "$OUT/pmc_${CTR}_conv_h8.txt"
  rm -rf "$OUT/pmc_$CTR"
done

rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS \
  --kernel-trace -d "$OUT/pmc_mfma" -o pmc -- python3 "$B" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_mfma.err"
DB=$(find "$OUT/pmc_mfma" -name "*.db" | head -1)
python3 "$ROOT/scripts/pmc_summary.py" conv_ "$DB" > "$OUT/pmc_mfma_busy_raw.txt"
rm -rf "$OUT/pmc_mfma"

cd "$ROOT"
# `make_profiles.sh <tag> pmc`: the bench line, the kernel trace and the counter passes only (a source change that does not touch what the
# micro-benchmarks below measure still changes the source hash the committed counters are attached by)
[ "${2:-all}" = pmc ] && { ls -la "$OUT"; exit 0; }
if [ "${2:-all}" = lean ]; then
  # `make_profiles.sh <tag> lean` (round 5): the above + per-layer / per-graph micro-benchmarks, the other BASELINE configurations, and this
  # round's switches one at a time (no ablation builds, no re-run of earlier rounds' switches)
  python3 scripts/bench_conv.py > "$OUT/microbench_conv.txt" 2>&1
  python3 scripts/bench_bn.py 128 > "$OUT/microbench_bn.txt" 2>&1
  python3 scripts/step_times.py > "$OUT/step_times.txt" 2>&1
  python3 scripts/bench_mnist.py 256 f32 > "$OUT/bench_mnist.txt" 2>&1
  python3 scripts/bench_wgrad_group.py > "$OUT/wgrad_group.txt" 2>&1
  python3 bench.py --no-cpu-baseline --dtype f32 --steps 20 > "$OUT/bench_f32.json" 2> /dev/null
  python3 bench.py --no-cpu-baseline --batch 512 --steps 8 > "$OUT/bench_b512.json" 2> /dev/null
  python3 bench.py --no-cpu-baseline --dtype f16 > "$OUT/bench_f16.json" 2> /dev/null
  python3 bench.py --no-cpu-baseline --algorithm rcgan-u > "$OUT/bench_rcganu.json" 2> /dev/null
  python3 bench.py --no-cpu-baseline --dp-stub 8 --dp-stub-gbps 200 --dp-stub-lat-us 40 > "$OUT/bench_dpstub8_model_f32.json" 2> /dev/null
  python3 bench.py --no-cpu-baseline --dp-stub 8 --dp-stub-gbps 200 --dp-stub-lat-us 40 --bucket-dtype bf16 > "$OUT/bench_dpstub8_model_bf16.json" 2> /dev/null
  # BASELINE configs[4]'s single-GPU leg: per-GPU batch 512, fp16 activations (round 6)
  python3 bench.py --no-cpu-baseline --dtype f16 --batch 512 --steps 8 > "$OUT/bench_f16_b512.json" 2> /dev/null
  # MNIST cfg2 under the profiler: per-kernel summary (55 iterations) and launches per iteration
  ( cd /tmp && rocprofv3 --kernel-trace -d "$OUT/prof_mnist" -o kt -- python3 "$ROOT/scripts/bench_mnist.py" 256 f32 > /dev/null 2>&1 )
  DBM=$(find "$OUT/prof_mnist" -name "*.db" | head -1)
  python3 scripts/prof_summary.py "$DBM" 55 --csv "$OUT/mnist_b256_f32_kernel_stats.csv" > "$OUT/mnist_b256_f32_kernel_stats.txt"
  python3 scripts/prof_summary.py "$DBM" 55 --by-grid > "$OUT/mnist_b256_f32_kernel_stats_by_grid.txt"
  rm -rf "$OUT/prof_mnist"
  { RCGAN_GG_WGRAD_FIT=0 RCGAN_S2_LPT=0 python3 scripts/bench_mnist.py 256 f32 2> /dev/null | sed "s/^/no_fit no_lpt : /"
    RCGAN_GG_WGRAD_FIT=1 RCGAN_S2_LPT=0 python3 scripts/bench_mnist.py 256 f32 2> /dev/null | sed "s/^/fit    no_lpt : /"
    RCGAN_GG_WGRAD_FIT=0 RCGAN_S2_LPT=1 python3 scripts/bench_mnist.py 256 f32 2> /dev/null | sed "s/^/no_fit lpt    : /"
    RCGAN_CONCAT_WGRAD=0 python3 scripts/bench_mnist.py 256 f32 2> /dev/null | sed "s/^/fit lpt, label columns inside the filter-gradient GEMM (RCGAN_CONCAT_WGRAD=0) : /"; } > "$OUT/bench_mnist_switches.txt"
  for rep in 1 2; do
    python3 bench.py --no-cpu-baseline > "$OUT/bench_default_rep$rep.json" 2> /dev/null
    # this round's switches, one at a time (the gather form's switch really switches it off since round 6: ADVICE r05)
    RCGAN_H8N_GATHER_MINBLK=100000 python3 bench.py --no-cpu-baseline > "$OUT/bench_no_gather_rep$rep.json" 2> /dev/null
    RCGAN_SN_ADAM=0 python3 bench.py --no-cpu-baseline > "$OUT/bench_no_sn_adam_rep$rep.json" 2> /dev/null
    RCGAN_OVERLAP_GF=1 python3 bench.py --no-cpu-baseline > "$OUT/bench_overlap_gf_rep$rep.json" 2> /dev/null
    RCGAN_CRITIC_GRAPH=0 python3 bench.py --no-cpu-baseline > "$OUT/bench_no_critic_graph_rep$rep.json" 2> /dev/null
    RCGAN_BN_INTO_PATCH=0 python3 bench.py --no-cpu-baseline > "$OUT/bench_no_bn_into_patch_rep$rep.json" 2> /dev/null
  done
  # the two data-parallel schedules under the same link model (ASSUMPTIONS: 40 us + 2(N-1)/N * bytes / 200 GB/s per all-reduce group)
  RCGAN_DP_OVERLAP=1 python3 bench.py --no-cpu-baseline --dp-stub 8 --dp-stub-gbps 200 --dp-stub-lat-us 40 > "$OUT/bench_dpstub8_model_f32_overlap.json" 2> /dev/null
  # the gather form where it could pay: per-GPU batch 512
  python3 bench.py --no-cpu-baseline --batch 512 --steps 8 > "$OUT/bench_b512_rep2.json" 2> /dev/null
  RCGAN_H8N_GATHER_MINBLK=100000 python3 bench.py --no-cpu-baseline --batch 512 --steps 8 > "$OUT/bench_b512_no_gather.json" 2> /dev/null
  ls -la "$OUT"; exit 0
fi
python3 scripts/bench_conv.py > "$OUT/microbench_conv.txt" 2>&1
python3 scripts/bench_bn.py 128 > "$OUT/microbench_bn.txt" 2>&1
python3 scripts/bench_trunk.py 128 > "$OUT/microbench_trunk.txt" 2>&1
STAMPS=1 python3 scripts/bench_rf.py 128 > "$OUT/microbench_rf.txt" 2>&1
python3 scripts/exp_p8_fixed_cost.py > "$OUT/exp_p8_fixed_cost.txt" 2>&1
python3 scripts/exp_p8_timeline.py 320 256 3 > "$OUT/exp_p8_timeline.txt" 2>&1
python3 scripts/step_times.py > "$OUT/step_times.txt" 2>&1
python3 scripts/bench_mnist.py 256 f32 > "$OUT/bench_mnist.txt" 2>&1
# MNIST cfg2 under the profiler: per-kernel summary (55 iterations) and launches per iteration
( cd /tmp && rocprofv3 --kernel-trace -d "$OUT/prof_mnist" -o kt -- python3 "$ROOT/scripts/bench_mnist.py" 256 f32 > /dev/null 2>&1 )
DBM=$(find "$OUT/prof_mnist" -name "*.db" | head -1)
python3 scripts/prof_summary.py "$DBM" 55 --csv "$OUT/mnist_b256_f32_kernel_stats.csv" > "$OUT/mnist_b256_f32_kernel_stats.txt"
python3 scripts/prof_summary.py "$DBM" 55 --by-grid > "$OUT/mnist_b256_f32_kernel_stats_by_grid.txt"
rm -rf "$OUT/prof_mnist"
# the reference's own precision: every layer on the fp32 matrix cores
python3 bench.py --no-cpu-baseline --dtype f32 --steps 20 > "$OUT/bench_f32.json" 2> /dev/null
# what bounds the 256 x 256 kernels (timing-only ablation builds; scripts/build_p8_ablate.sh ran before the snapshot)
{ echo "# per K-tile cost of the tile-per-tap 256x256 kernel (conv_mfma_p8_kernel, ping-pong schedule) and of the halo-patch kernel (conv_mfma_h8_kernel):"
  echo "# scripts/exp_p8_fixed_cost.py quick (n = 320, 3x3, Cin 128 / 256 / 512: fit time = fixed + per-K-tile x KT); ablation builds are wrong by construction"
  echo "halo-patch kernel as built:                 $(python3 scripts/exp_p8_fixed_cost.py quick 2>/dev/null | tail -1)"
  echo "tile-per-tap kernel (RCGAN_P8_HALO=0):      $(RCGAN_P8_HALO=0 python3 scripts/exp_p8_fixed_cost.py quick 2>/dev/null | tail -1)"
  echo "tile-per-tap, ping-pong (RCGAN_P8_PP=1):    $(RCGAN_P8_HALO=0 RCGAN_P8_PP=1 python3 scripts/exp_p8_fixed_cost.py quick 2>/dev/null | tail -1)"
  for k in 1 2 4 3 16 h1 h2 h4 h6 h8 h16; do
    L=scripts/probes/_bin/librcgan_abl$k.so
    [ -f $L ] && echo "ablation $k: $(RCGAN_LIB_PATH=$PWD/$L RCGAN_P8_PP=1 RCGAN_P8_HALO=$([ "${k#h}" != "$k" ] && echo 1 || echo 0) python3 scripts/exp_p8_fixed_cost.py quick 2>/dev/null | tail -1)"
  done; } > "$OUT/exp_p8_ablation.txt"
python3 scripts/bench_wgrad_group.py > "$OUT/wgrad_group.txt" 2>&1
# what bounds the filter-gradient kernels (scripts/build_p8_ablate.sh w1 w2 w4 w8 w3 w6 n1 n2 n3 ran before the snapshot)
{ echo "# rcgan_conv2d_bwd_weight_group on plain 3x3 layers at n = 128 (HIP events, launch + slab reduction), the kernel as built and with one part of its loop removed"
  echo "# (timing-only builds, wrong by construction).  THREE-TAP kernel (conv_mfma_wgrad3_group_kernel, RCGAN_WGRAD9=0):"
  RCGAN_WGRAD9=0 bash scripts/exp_wgrad_ablation.sh
  echo "# NINE-TAP kernel (conv_wgrad9_group_kernel; RCGAN_WGRAD9_GROUP_MINWORK=0: the grouped entry point takes it whatever the size):"
  RCGAN_WGRAD9_GROUP_MINWORK=0 bash scripts/exp_wgrad9_ablation.sh; } > "$OUT/exp_wgrad_ablation.txt" 2>&1
RCGAN_WGRAD9=0 python3 bench.py --no-cpu-baseline > "$OUT/bench_no_wgrad9.json" 2> /dev/null
RCGAN_WGRAD9_GROUP_MINWORK=0 python3 bench.py --no-cpu-baseline > "$OUT/bench_wgrad9_everywhere.json" 2> /dev/null
python3 bench.py --no-cpu-baseline --batch 512 --steps 8 > "$OUT/bench_b512.json" 2> /dev/null
python3 bench.py --no-cpu-baseline --dtype f16 > "$OUT/bench_f16.json" 2> /dev/null
python3 bench.py --no-cpu-baseline --dtype f16 --batch 512 --steps 8 > "$OUT/bench_f16_b512.json" 2> /dev/null
python3 bench.py --no-cpu-baseline --algorithm rcgan-u > "$OUT/bench_rcganu.json" 2> /dev/null
# the world-size-8 step schedule against the in-ABI test-double communicator (no traffic): what the schedule itself costs
python3 bench.py --no-cpu-baseline --dp-stub 8 > "$OUT/bench_dpstub8.json" 2> /dev/null
# ... and under a stated link model (40 us + 2(N-1)/N * bytes / 200 GB/s per all-reduce group): the predicted 8-rank iteration, fp32 / bf16 buckets
python3 bench.py --no-cpu-baseline --dp-stub 8 --dp-stub-gbps 200 --dp-stub-lat-us 40 > "$OUT/bench_dpstub8_model_f32.json" 2> /dev/null
python3 bench.py --no-cpu-baseline --dp-stub 8 --dp-stub-gbps 200 --dp-stub-lat-us 40 --bucket-dtype bf16 > "$OUT/bench_dpstub8_model_bf16.json" 2> /dev/null
RCGAN_FUSE_BN_STATS=1 python3 bench.py --no-cpu-baseline > "$OUT/bench_fuse_bn_stats.json" 2> /dev/null
RCGAN_HEAD_RIDERS=0 python3 bench.py --no-cpu-baseline > "$OUT/bench_no_head_riders.json" 2> /dev/null
RCGAN_POOL_IN_TRUNK=0 python3 bench.py --no-cpu-baseline > "$OUT/bench_no_pool_in_trunk.json" 2> /dev/null
RCGAN_RF_CONV=0 python3 bench.py --no-cpu-baseline > "$OUT/bench_no_rf_conv.json" 2> /dev/null
RCGAN_FUSED_TRUNK=0 python3 bench.py --no-cpu-baseline > "$OUT/bench_no_fused_trunk.json" 2> /dev/null
RCGAN_BN_INTO_CONV=0 python3 bench.py --no-cpu-baseline > "$OUT/bench_no_bn_into_conv.json" 2> /dev/null
RCGAN_LINEAR_MFMA=0 python3 bench.py --no-cpu-baseline > "$OUT/bench_no_linear_mfma.json" 2> /dev/null
RCGAN_IMG_OUT_STAGES=2 python3 bench.py --no-cpu-baseline > "$OUT/bench_img_out_stages2.json" 2> /dev/null
python3 scripts/exp_bench_data.py 60 32 > "$OUT/bench_data_smooth.txt" 2>&1
RCGAN_BENCH_IMAGES=uniform python3 scripts/exp_bench_data.py 60 32 > "$OUT/bench_data_uniform.txt" 2>&1
[ -x scripts/probes/_bin/epilogue_store ] && ./scripts/probes/_bin/epilogue_store > "$OUT/probe_epilogue_store.txt" 2>&1
[ -x scripts/probes/_bin/filter_fetch ] && ./scripts/probes/_bin/filter_fetch > "$OUT/probe_filter_fetch.txt" 2>&1
[ -x scripts/probes/_bin/lds_dma_fetch ] && ./scripts/probes/_bin/lds_dma_fetch > "$OUT/probe_lds_dma_fetch.txt" 2>&1
ls -la "$OUT"
