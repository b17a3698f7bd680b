#!/usr/bin/env python3
"""Experiment: the small filter gradients of the discriminator (8x8 / 16x16, 128 channels, n=128) launched back to back on
one stream vs spread over several streams -- how much of their time is latency that concurrency would hide."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import rcgan_amd  # noqa: E402,F401
from rcgan_amd import _lib as L  # noqa: E402
from rcgan_amd.runtime import Context  # noqa: E402


def setup(ctx, n, hw, c):
    lib, h = ctx.lib, ctx.h
    x = ctx.empty((n, hw, hw, c)); dy = ctx.empty((n, hw, hw, c))
    dw = ctx.empty((3, 3, c, c), L.F32)
    for t in (x, dy):
        ctx.check(lib.rcgan_rng_fill(h, t.size, t.dtype, 1, 0.0, 1.0, 7, None, C.c_void_p(t.ptr)))
    desc = L.ConvDesc(n, hw, hw, c, c, 3, 3, 1, L.BF16, L.CONV_IN_RELU)
    def call():
        ctx.check(lib.rcgan_conv2d_bwd_weight(h, C.byref(desc), C.c_void_p(x.ptr), C.c_void_p(dy.ptr), C.c_void_p(dw.ptr), None, 0,
                                              C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
    return call


def main():
    nctx = 4
    ctxs = [Context(0, "bf16", arena_bytes=1 << 29, ws_bytes=1 << 28) for _ in range(nctx)]
    for hw in (8, 16):
        calls = [setup(c, 128, hw, 128) for c in ctxs]
        for c in calls:
            c()
        torch.cuda.synchronize()
        reps = 50
        t0 = time.time()
        for _ in range(reps):
            for _ in range(nctx):
                calls[0]()
        torch.cuda.synchronize()
        seq = (time.time() - t0) / reps * 1e6
        t0 = time.time()
        for _ in range(reps):
            for c in calls:
                c()
        torch.cuda.synchronize()
        par = (time.time() - t0) / reps * 1e6
        print("%dx%d: %d filter gradients on one stream %.1f us, on %d streams %.1f us" % (hw, hw, nctx, seq, nctx, par))


if __name__ == "__main__":
    main()
