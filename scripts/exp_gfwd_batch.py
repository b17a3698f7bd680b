#!/usr/bin/env python3
"""Experiment: generator forward (no grad) as 5 graph replays at n=64 vs one replay at n=320 (timing only; the batched
variant here takes its batch-norm statistics over all 320 samples, which is NOT the reference semantics)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import rcgan_amd  # noqa: E402,F401
from rcgan_amd.cifar import CifarRCGAN, Generator  # noqa: E402


def main():
    m = CifarRCGAN(algorithm="rcgan", batch_size=64, dtype="bf16", arena_bytes=6 << 30)
    ctx, g = m.ctx, m.graph
    rs = np.random.RandomState(0)
    for n in (64, 128, 320):
        lab = ctx.persistent((n,), "i32")
        ctx.view(lab).copy_(torch.from_numpy(rs.randint(10, size=n).astype(np.int32)))
        z = ctx.persistent((n, 128), ctx.act_dtype)
        ctx.view(z).copy_(torch.from_numpy(rs.randn(n, 128).astype(np.float32)))

        def body():
            m._refresh_generator_filters()
            ctx.new_step()
            g.begin_step(set())
            rec, ctx.recording = ctx.recording, False
            try:
                m._prepare_all((m.PG,))
                Generator(n, lab, z)
            finally:
                ctx.recording = rec
        body()
        ctx.sync()
        ctx.graph_begin()
        body()
        gid = ctx.graph_end()
        for _ in range(3):
            ctx.graph_launch(gid)
        ctx.sync()
        reps = 50
        t0 = time.time()
        for _ in range(reps):
            ctx.graph_launch(gid)
        ctx.sync()
        us = (time.time() - t0) / reps * 1e6
        print("G forward n=%d: %.1f us per replay, %.2f us per sample" % (n, us, us / n))


if __name__ == "__main__":
    main()
