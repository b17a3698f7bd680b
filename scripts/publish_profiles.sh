#!/bin/bash
# Copy the summaries of one scripts/make_profiles.sh run (gpurun_out/<tag>/) into profiles/ under the round's names.
#   bash scripts/publish_profiles.sh r02h r02
set -e
R=gpurun_out/$1; P=profiles/$2
c() { if [ -f "$1" ]; then cp "$1" "$2"; else echo "(missing: $1)"; fi; }       # every input is optional: publish what the run produced
c $R/bench_n1_default.json ${P}_bench_n1_default.json
c $R/bench_n1_under_rocprof.json ${P}_bench_n1_under_rocprof.json
c $R/bench_n1_kernel_stats.csv ${P}_bench_n1_kernel_stats.csv
c $R/bench_n1_kernel_stats_by_grid.csv ${P}_bench_n1_kernel_stats_by_grid.csv
c $R/bench_n1_last_iteration_sequence.txt ${P}_bench_n1_last_iteration_sequence.txt
c $R/exp_p8_fixed_cost.txt ${P}_exp_p8_fixed_cost.txt
c $R/exp_p8_timeline.txt ${P}_exp_p8_timeline.txt
c $R/probe_epilogue_store.txt ${P}_probe_epilogue_store.txt
c $R/probe_filter_fetch.txt ${P}_probe_filter_fetch.txt
c $R/probe_lds_dma_fetch.txt ${P}_probe_lds_dma_fetch.txt
c $R/exp_p8_ablation.txt ${P}_exp_p8_ablation.txt
c $R/exp_wgrad_ablation.txt ${P}_exp_wgrad_ablation.txt
c $R/mnist_b256_f32_kernel_stats.csv ${P}_mnist_b256_f32_kernel_stats.csv
c $R/mnist_b256_f32_kernel_stats_by_grid.txt ${P}_mnist_b256_f32_kernel_stats_by_grid.txt
c $R/bench_f32.json ${P}_bench_n1_f32.json
c $R/bench_dpstub8_model_f32.json ${P}_bench_dpstub8_model_f32.json
c $R/bench_dpstub8_model_bf16.json ${P}_bench_dpstub8_model_bf16.json
# counters are attached to a build by the hash make_profiles.sh recorded when it measured (source_sha16.txt): both scripts refuse a
# run whose hash is not the tree's
python3 scripts/pmc_traffic_json.py $R/pmc_FETCH_SIZE_conv_h8.txt $R/pmc_WRITE_SIZE_conv_h8.txt ${P}_pmc_traffic_conv_h8.json $R/source_sha16.txt > /dev/null
python3 scripts/pmc_busy_table.py $R/pmc_mfma_busy_raw.txt ${P}_pmc_mfma_busy.json $R/bench_n1_kernel_stats_by_grid.csv $R/source_sha16.txt > ${P}_pmc_mfma_busy.txt
F=${P}_microbench.txt
g() { if [ -f "$1" ]; then grep -v "amdgpu.ids" "$1"; else echo "(not collected in this run)"; fi; }
echo "# scripts/bench_conv.py 64  (HIP events, 20 launches each, eager; n = 2B = 128 unless noted; every layer as the reference poses it -- the up blocks' shortcuts at full resolution, D.Block.1/2.Conv2 without their pool)" > $F; g $R/microbench_conv.txt >> $F
echo "# scripts/bench_bn.py 128  (conditional batch norm entry points, bf16: statistics / apply+ReLU / backward = 2 launches)" >> $F; g $R/microbench_bn.txt >> $F
echo "# scripts/bench_trunk.py 128  (the fused 8x8 stage, RCGAN_FUSED_TRUNK, against its eight launches)" >> $F; g $R/microbench_trunk.txt >> $F
echo "# STAMPS=1 scripts/bench_rf.py 128  (the register-filter convolution, rcgan_conv2d_rf, against the tile-per-tap kernels; per-workgroup s_memtime segments)" >> $F; g $R/microbench_rf.txt >> $F
echo "# scripts/step_times.py  (HIP-graph replays, B = 64)" >> $F; g $R/step_times.txt >> $F
echo "# scripts/bench_mnist.py 256 f32" >> $F; g $R/bench_mnist.txt >> $F
[ -f $R/bench_mnist_switches.txt ] && { echo "# ... with round 6's switches off, one or two at a time (RCGAN_GG_WGRAD_FIT: filter-gradient splits fitted to whole rounds; RCGAN_S2_LPT: stride-2 parity classes longest first; RCGAN_CONCAT_WGRAD: the label columns of the transposed convolutions' filter gradient from per-sample sums)" >> $F; cat $R/bench_mnist_switches.txt >> $F; }
echo "# scripts/bench_wgrad_group.py  (rcgan_conv2d_bwd_weight_group on the critic step's layer set as PLAIN 3x3 layers, n = 128; grouped launches + grouped reduction)" >> $F; g $R/wgrad_group.txt >> $F
echo "# python bench.py --no-cpu-baseline --batch 512 --steps 8 | --dtype f16 | --algorithm rcgan-u   (ms per iteration, images/s, sustained TFLOP/s at the reference's FLOP count, dominant kernel: fraction of peak at the reference's count / executed)" >> $F
for f in b512 f16 f16_b512 rcganu dpstub8 dpstub8_model_f32 dpstub8_model_bf16 dpstub8_model_f32_overlap no_wgrad9 wgrad9_everywhere fuse_bn_stats no_head_riders no_pool_in_trunk no_rf_conv no_fused_trunk no_bn_into_conv no_linear_mfma img_out_stages2 \
         default_rep1 default_rep2 no_p8n_halo_rep1 no_p8n_halo_rep2 no_two_pass_rep1 no_two_pass_rep2 no_gather_rep1 no_gather_rep2 no_bn_into_patch_rep1 no_bn_into_patch_rep2 graph_adam_rep1 graph_adam_rep2 \
         no_sn_adam_rep1 no_sn_adam_rep2 no_critic_graph_rep1 no_critic_graph_rep2 overlap_gf_rep1 overlap_gf_rep2 b512_rep2 b512_no_gather; do [ -s $R/bench_$f.json ] && python3 -c "
import json
d=json.load(open('$R/bench_$f.json')); print('%-16s %8.3f ms %10.1f img/s %8.1f TFLOP/s   %.3f / %.3f' % ('$f', d['ms_per_step'], d['value'], d['config']['sustained_tflops'], d['roofline']['frac'], d['roofline']['executed_frac']))" >> $F; done
echo "# scripts/exp_bench_data.py 60 32: d_loss / g_loss of the bench workload, smooth class-conditional images (default) vs uniform noise (rounds 1-2)" >> $F
for k in smooth uniform; do [ -f $R/bench_data_$k.txt ] && { echo "## $k"; grep "^it" $R/bench_data_$k.txt | awk '{printf "%s:%s/%s  ", $2, $4, $6} END {print ""}'; } >> $F; done
python3 - <<PY
import csv
rows=list(csv.DictReader(open('${P}_bench_n1_kernel_stats.csv')))
its=24
cat={}
def c(n):
    if 'wgrad' in n or 'slab_reduce' in n: return 'filter gradients (+ slab reduction)'
    if 'conv_mfma' in n or 'conv_trunk' in n or 'conv_rf' in n: return 'MFMA convolutions fwd/dgrad'
    if 'bn_' in n: return 'batch norm'
    if 'conv_img' in n: return 'image-end convolutions'
    if 'prepare' in n or n.startswith('sn_'): return 'spectral norm + filter preparation (+ riders)'
    if 'head' in n or 'meanhw' in n: return 'projection head'
    return 'other'
tot=0
for r in rows:
    t=float(r['TotalDurationNs'])/1e6/its
    cat[c(r['Name'])]=cat.get(c(r['Name']),0)+t; tot+=t
for k,v in sorted(cat.items(), key=lambda x:-x[1]): print("%-48s %.3f ms  %.1f%%"%(k,v,100*v/tot))
print("total %.3f ms" % tot)
PY
grep "h8_kernel\|p8_kernel" ${P}_pmc_mfma_busy.txt; grep "traffic_bytes_per_launch\|algorithmic_bytes_per_launch" ${P}_pmc_traffic_conv_h8.json
