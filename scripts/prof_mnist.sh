#!/bin/bash
# MNIST cfg2 (B = 256, fp32) under the profiler: per-kernel summary of `iters` iterations -> gpurun_out/mnist_kstats.{txt,csv}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace -d $R/gpurun_out/ktm -o kt -- python3 $R/scripts/bench_mnist.py 256 f32 > $R/gpurun_out/mnist_under_rocprof.txt 2>&1
DB=$(find $R/gpurun_out/ktm -name "*.db" | head -1)
python3 $R/scripts/prof_summary.py $DB 55 --csv $R/gpurun_out/mnist_kstats.csv > $R/gpurun_out/mnist_kstats.txt
python3 $R/scripts/prof_summary.py $DB 55 --by-grid > $R/gpurun_out/mnist_kstats_by_grid.txt
rm -rf $R/gpurun_out/ktm
head -50 $R/gpurun_out/mnist_kstats_by_grid.txt
