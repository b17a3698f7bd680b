#!/usr/bin/env python3
"""Premise test (round 6): do the critic steps' generator forwards hide under the critic steps when they run on a SECOND stream?

The generator does not change during the N_CRITIC critic updates of an iteration (gan_resnet.py:928-947), so the fakes of step
t + 1 can be produced while step t runs.  A critic step is a chain of small-grid launches (half the chip idle most of the time);
the generator forward is big-grid work.  Two engines in one process = two contexts = two streams, separate arenas / workspaces /
counters; no dependencies between them here (timing only, the critic reads stale fakes):

  serial      : engine A alone: prepare_critic_fakes (n = 5B) + 5 critic steps           (what bench.py's iteration does)
  concurrent  : engine A: 5 critic steps   ||   engine B: prepare_critic_fakes (n = 5B)  (one big pass next to the chain)
  pipelined   : engine A: 5 critic steps   ||   engine C (batch B/4... see --chunk): 5B/chunk passes of `chunk` images

  python scripts/exp_overlap_streams.py [--batch 64] [--reps 30]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--chunk_batch", type=int, default=16, help="engine C's batch: its generator pass covers 5 x this many images")
    a = ap.parse_args()
    import rcgan_amd  # noqa: F401
    from rcgan_amd.cifar import N_CRITIC, CifarRCGAN
    B = a.batch
    mk = lambda b: CifarRCGAN(algorithm="rcgan", alpha=0.6, batch_size=b, dtype="bf16", seed=0)
    A, Bm, Cm = mk(B), mk(B), mk(a.chunk_batch)
    pools = {id(m): bench.build_pool(m, 0, 0.6) for m in (A, Bm, Cm)}
    dc = {id(m): [0] for m in (A, Bm, Cm)}

    def gf(m):
        m.set_feed("gf", pools[id(m)]["feed_gf"][dc[id(m)][0] % bench.POOL])
        m.prepare_critic_fakes()

    def critics(m):
        for _ in range(N_CRITIC):
            bench.feed_d(m, pools[id(m)], dc[id(m)][0])
            dc[id(m)][0] += 1
            m.d_step(iteration=1)

    # warm-up: captures every graph (full iterations so that d_fakes / gf / g exist)
    for m in (A, Bm, Cm):
        for it in range(3):
            bench.iteration(m, pools[id(m)], it, dc[id(m)])
    torch.cuda.synchronize()

    def timed(fn, reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def serial():
        gf(A)
        critics(A)

    def only_critics():
        A._fakes_left = N_CRITIC
        A._slice_mirror = 0
        with torch.cuda.stream(A.ctx.stream):
            A.slice_ctr.zero_()
        critics(A)

    def only_gf():
        gf(Bm)

    def concurrent():
        gf(Bm)
        only_critics()

    n_chunks = max(1, (N_CRITIC * B) // (N_CRITIC * a.chunk_batch))

    def only_gf_chunks():
        for _ in range(n_chunks):
            gf(Cm)

    def pipelined():
        # interleave the launches so that neither stream's queue runs dry: one chunk pass per critic step
        A._fakes_left = N_CRITIC
        A._slice_mirror = 0
        with torch.cuda.stream(A.ctx.stream):
            A.slice_ctr.zero_()
        per = max(1, n_chunks // N_CRITIC)
        done = 0
        for s in range(N_CRITIC):
            for _ in range(per):
                if done < n_chunks:
                    gf(Cm)
                    done += 1
            bench.feed_d(A, pools[id(A)], dc[id(A)][0])
            dc[id(A)][0] += 1
            A.d_step(iteration=1)
        while done < n_chunks:
            gf(Cm)
            done += 1

    res = {}
    for name, fn in (("serial: gf(5B) + 5 critic steps, one stream", serial), ("5 critic steps alone", only_critics),
                     ("gf(5B) alone", only_gf), ("concurrent: gf(5B) || 5 critic steps", concurrent),
                     ("gf in %d passes of %d images alone" % (n_chunks, N_CRITIC * a.chunk_batch), only_gf_chunks),
                     ("pipelined: %d passes of %d images || 5 critic steps" % (n_chunks, N_CRITIC * a.chunk_batch), pipelined)):
        fn()
        ms = [timed(fn, a.reps) for _ in range(3)]
        res[name] = ms
        print("%-62s %s ms" % (name, " / ".join("%.3f" % v for v in ms)), flush=True)
    for m in (A, Bm, Cm):
        m.ctx.close()


if __name__ == "__main__":
    main()
