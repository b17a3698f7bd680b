#!/usr/bin/env python3
"""Micro-benchmark of the batch-norm entry points on the CIFAR generator shapes (HIP events on the launch stream)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

import rcgan_amd  # noqa: E402,F401
from rcgan_amd import _lib as L  # noqa: E402
from rcgan_amd.runtime import Context  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    ctx = Context(0, "bf16", arena_bytes=4 << 30, ws_bytes=1 << 30)
    lib, h = ctx.lib, ctx.h
    P = C.c_void_p
    reps = 20
    print("%-22s %10s %10s %10s" % ("shape", "stats us", "apply us", "bwd us"))
    for (n, hw, c) in [(B, 4, 1024), (B, 8, 256), (B, 16, 256), (B, 32, 256)]:
        ctx.new_step()
        rps = hw * hw
        rows = n * rps
        x = ctx.empty((rows, c)); y = ctx.empty((rows, c)); dy = ctx.empty((rows, c)); dx = ctx.empty((rows, c))
        for t in (x, dy):
            ctx.check(lib.rcgan_rng_fill(h, t.size, t.dtype, 1, 0.0, 1.0, 7, None, P(t.ptr)))
        gamma = ctx.upload(np.ones((10, c), np.float32), L.F32); beta = ctx.upload(np.zeros((10, c), np.float32), L.F32)
        dg = ctx.empty((10, c), L.F32); db = ctx.empty((10, c), L.F32)
        mean = ctx.empty((c,), L.F32); rstd = ctx.empty((c,), L.F32)
        labels = ctx.upload(np.random.RandomState(0).randint(10, size=n).astype(np.int32))
        wsb = lib.rcgan_bn_workspace_bytes(rows, c)
        ws = ctx.arena.alloc(wsb)

        def stats():
            ctx.check(lib.rcgan_bn_stats(h, rows, c, L.BF16, P(x.ptr), 1e-5, P(mean.ptr), P(rstd.ptr), None, None, 0.9, P(ws), wsb))

        def apply():
            ctx.check(lib.rcgan_bn_apply_fwd(h, n, rps, c, 10, L.BF16, P(x.ptr), P(labels.ptr), P(gamma.ptr), P(beta.ptr), P(mean.ptr),
                                             P(rstd.ptr), L.ACT_RELU, P(y.ptr), P(ws), wsb))

        def bwd():
            ctx.check(lib.rcgan_bn_bwd2(h, n, rps, c, 10, L.BF16, P(x.ptr), P(y.ptr), P(dy.ptr), P(labels.ptr), P(gamma.ptr), P(beta.ptr), P(mean.ptr),
                                       P(rstd.ptr), L.ACT_RELU, P(dx.ptr), 0, P(dg.ptr), P(db.ptr), 0, P(ws), wsb))
        res = []
        for fn in (stats, apply, bwd):
            fn(); fn()
            ctx.event_record(0)
            for _ in range(reps):
                fn()
            ctx.event_record(1)
            res.append(ctx.event_elapsed_ms(0, 1) * 1e3 / reps)
        print("%-22s %10.1f %10.1f %10.1f" % ("[%d,%d,%d,%d]" % (n, hw, hw, c), *res))


if __name__ == "__main__":
    main()
