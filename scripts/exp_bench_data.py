#!/usr/bin/env python3
"""d_loss / g_loss trajectory of the bench workload under the two synthetic image kinds (RCGAN_BENCH_IMAGES=smooth|uniform):
   python scripts/exp_bench_data.py [iterations] [pool]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rcgan_amd  # noqa: F401,E402
from rcgan_amd.cifar import CifarRCGAN  # noqa: E402

its = int(sys.argv[1]) if len(sys.argv) > 1 else 60
if len(sys.argv) > 2:
    bench.POOL = int(sys.argv[2])
m = CifarRCGAN(algorithm="rcgan", alpha=0.6, batch_size=64, dtype="bf16", seed=0, use_graphs=True, device_rng=True)
pool = bench.build_pool(m, 0, 0.6)
dc = [0]
for it in range(its):
    bench.iteration(m, pool, it, dc)
    if it % 5 == 4 or it < 3:
        print("it %3d  d_loss %.4f  g_loss %.4f" % ((it,) + m.losses()), flush=True)
m.ctx.close()
