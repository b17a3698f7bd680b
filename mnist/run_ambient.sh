#!/bin/bash
# ambient on MNIST: the reference preset (mnist/run_ambient.sh), one MI355X.
# Multi-GPU: NGPUS=8 ./run_ambient.sh starts one rank per GPU (RCCL gradient all-reduce); extra flags pass through.
script_file='run_ambient.sh'
checkpoint_dir='ambient'
trial=0
alpha=0.6
epoch=100
ngpus=${NGPUS:-1}
mkdir -p "$checkpoint_dir"
launch="python -u"
if [ "$ngpus" -gt 1 ]; then
  launch="python -m torch.distributed.run --nnodes=1 --nproc-per-node $ngpus --master-addr 127.0.0.1"
fi
$launch main.py \
    --algorithm "ambient" --alpha $alpha --disc_type "vanilla" \
    --loss_fn "ce" --real_match \
    --noestimate_confuse --noaux_classifier \
    --noadd_noise --noconcat_y \
    --nospectral_norm --nomax_norm \
    --checkpoint_dir $checkpoint_dir --script_file ${script_file} \
    --epoch $epoch "$@" 2>&1 | tee -a ${checkpoint_dir}/ambient_alpha${alpha}_epoch${epoch}_${trial}.txt
