import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
os.environ["RCGAN_DP_OVERLAP"] = "1"
from tests.test_gpu_dp import _feeds, _model
import rcgan_amd
from rcgan_amd import ops as O
from rcgan_amd.cifar import N_CRITIC
for rf in (True, False):
    O.RF_CONV = rf
    for seed in (12, 13):
        rs = np.random.RandomState(seed)
        its = _feeds(rs, 8, 3, "rcgan-u")
        res = []
        for w in (1, 2):
            m = _model("rcgan-u", "bf16", 8, world_size=w, comm=("stub" if w > 1 else None))
            ls = []
            for it, (lra, ds, g) in enumerate(its):
                m.set_feed("gf", m.pack_feed("gf", labels_random_all=lra)); m.prepare_critic_fakes()
                for k, d in enumerate(ds):
                    m.set_feed("d", m.pack_feed("d", **d)); m.d_step(iteration=it)
                m.set_feed("g", m.pack_feed("g", **g)); m.g_step(iteration=it + 1)
                ls.append(tuple(round(v, 4) for v in m.losses()))
            res.append(ls); m.ctx.close()
        print("rf", rf, "seed", seed, "w1", res[0], "w2", res[1])
