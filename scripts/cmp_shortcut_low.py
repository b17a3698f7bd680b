import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, rcgan_amd
from rcgan_amd.cifar import CifarRCGAN
m = CifarRCGAN(algorithm="rcgan", alpha=0.6, batch_size=64, dtype="bf16", seed=0, device=0, device_rng=True)
pool = bench.build_pool(m, 0, 0.6)
dc = [0]
for it in range(8):
    bench.iteration(m, pool, it, dc)
    d, g = m.losses()
    pg = m.PG.value.double()
    pd = m.PD.value.double()
    print("it %d d_loss %.5f g_loss %.5f |G| %.6f |D| %.6f" % (it, d, g, float(pg.norm()), float(pd.norm())))
