#!/bin/bash
# biased on CIFAR-10, alpha = 0.6 (label-noise level 0.4): the reference preset, one MI355X.
# Multi-GPU: NGPUS=8 ./run_biased.sh starts one rank per GPU (RCCL gradient all-reduce).
out=biased
run_id=0
alpha=0.6
ngpus=${NGPUS:-1}
mkdir -p "$out"
log="$out/biased_alpha${alpha}_${run_id}_log.txt"
launch="python"
if [ "$ngpus" -gt 1 ]; then
  launch="python -m torch.distributed.run --nnodes=1 --nproc-per-node $ngpus --master-addr 127.0.0.1"
fi
$launch gan_resnet.py --dataset cifar --algorithm biased --alpha $alpha --run $run_id \
  --log_file "$log" --parent_dir "$out" --ngpus $ngpus --multi_gpu_multi_batch "$@"
