#!/usr/bin/env python3
"""Wall time of one D step and one G step (hipGraph replay, bench inputs): where the iteration goes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import rcgan_amd  # noqa: E402,F401
from rcgan_amd.cifar import CifarRCGAN  # noqa: E402


def main():
    m = CifarRCGAN(algorithm="rcgan", alpha=0.6, batch_size=64, dtype="bf16", seed=0, device=0)
    pool = bench.build_pool(m, 0, 0.6)
    dc = [0]
    for it in range(3):
        bench.iteration(m, pool, it, dc)
    torch.cuda.synchronize()
    def d5(i):
        m.set_feed("gf", pool["feed_gf"][i % bench.POOL])
        m.prepare_critic_fakes()
        for k in range(5):
            bench.feed_d(m, pool, i + k)
            m.d_step(iteration=5)

    def gf(i):
        m.set_feed("gf", pool["feed_gf"][i % bench.POOL])
        m.prepare_critic_fakes()
        m._fakes_left = 0

    for name, fn, reps in (("prepare_critic_fakes (G forward, 5B samples)", gf, 20),
                           ("prepare_critic_fakes + 5 d_steps", d5, 20),
                           ("d_step incl. its own G forward", lambda i: (bench.feed_d(m, pool, i), m.d_step(iteration=5)), 50),
                           ("g_step", lambda i: (bench.feed_g(m, pool, i), m.g_step(iteration=5)), 20)):
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(reps):
            fn(i)
        torch.cuda.synchronize()
        print("%s: %.3f ms" % (name, (time.time() - t0) * 1e3 / reps))


if __name__ == "__main__":
    main()
