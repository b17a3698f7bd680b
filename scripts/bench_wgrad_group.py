#!/usr/bin/env python3
"""Micro-benchmark of rcgan_conv2d_bwd_weight_group on the discriminator's layer set of one critic step (HIP events).

usage: python scripts/bench_wgrad_group.py [B]      (B = per-GPU critic batch, default 64; the step sees 2B images)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import rcgan_amd  # noqa: E402,F401
from rcgan_amd import _lib as L  # noqa: E402
from rcgan_amd.runtime import Context  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = 2 * B
    ctx = Context(0, "bf16", arena_bytes=4 << 30, ws_bytes=1 << 30)
    lib, h = ctx.lib, ctx.h
    R = L.CONV_IN_RELU
    sets = {
        "three-tap only": [(n, 32, 32, 128, 128, 3, R), (n, 16, 16, 128, 128, 3, R), (n, 16, 16, 128, 128, 3, R)] + [(n, 8, 8, 128, 128, 3, R)] * 8,
        "+ 1x1 shortcut": [(n, 32, 32, 128, 128, 3, R), (n, 16, 16, 128, 128, 3, R), (n, 16, 16, 128, 128, 3, R)] + [(n, 8, 8, 128, 128, 3, R)] * 8 +
                          [(n, 16, 16, 128, 128, 1, 0)],
        "critic step (+ image-end layers)": [(n, 32, 32, 3, 128, 3, 0), (n, 16, 16, 3, 128, 1, 0), (n, 32, 32, 128, 128, 3, R), (n, 16, 16, 128, 128, 3, R),
                                             (n, 16, 16, 128, 128, 3, R), (n, 16, 16, 128, 128, 1, 0)] + [(n, 8, 8, 128, 128, 3, R)] * 8,
        "image-end layers alone": [(n, 32, 32, 3, 128, 3, 0), (n, 16, 16, 3, 128, 1, 0)],
        # the generator step's 256-channel layers as plain 3x3 layers (the up blocks' Conv1 at their output resolution)
        "generator 256-ch: 32x32 alone": [(n, 32, 32, 256, 256, 3, R)],
        "generator 256-ch: 32x32 x2, 16x16 x2, 8x8 x2": [(n, 32, 32, 256, 256, 3, R)] * 2 + [(n, 16, 16, 256, 256, 3, R)] * 2 + [(n, 8, 8, 256, 256, 3, R)] * 2,
        "critic 128-ch: 32x32 alone": [(n, 32, 32, 128, 128, 3, R)],
    }
    only = os.environ.get("WGRAD_SETS")
    if only:
        sets = {k: v for k, v in sets.items() if any(o in k for o in only.split(","))}
    reps = 20
    for name, shapes in sets.items():
        ctx.new_step()
        items = []
        for i, (nn, hh, ww, cin, cout, k, fl) in enumerate(shapes):
            x, dy = ctx.empty((nn, hh, ww, cin)), ctx.empty((nn, hh, ww, cout))
            ctx.check(lib.rcgan_rng_fill(h, x.size, x.dtype, 1, 0.0, 1.0, 100 + i, None, C.c_void_p(x.ptr)))
            ctx.check(lib.rcgan_rng_fill(h, dy.size, dy.dtype, 1, 0.0, 1.0, 200 + i, None, C.c_void_p(dy.ptr)))
            items.append((L.ConvDesc(nn, hh, ww, cin, cout, k, k, 1, L.BF16, fl), x, dy, ctx.zeros((k, k, cin, cout), L.F32), ctx.zeros((cout,), L.F32)))
        m = len(items)
        descs = (L.ConvDesc * m)(*[it[0] for it in items])
        arr = lambda f: (C.c_void_p * m)(*[f(it) for it in items])
        xs, dys, dws, dbs = arr(lambda it: it[1].ptr), arr(lambda it: it[2].ptr), arr(lambda it: it[3].ptr), arr(lambda it: it[4].ptr)

        def call():
            ctx.check(lib.rcgan_conv2d_bwd_weight_group(h, m, descs, xs, dys, dws, dbs, 0, C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
        call(); call()
        ctx.event_record(0)
        for _ in range(reps):
            call()
        ctx.event_record(1)
        us = ctx.event_elapsed_ms(0, 1) * 1e3 / reps
        gf = sum(2.0 * nn * hh * ww * cin * cout * k * k for (nn, hh, ww, cin, cout, k, fl) in shapes) / 1e9
        print("%-48s %8.1f us  %7.1f GFLOP  %6.1f TFLOP/s" % (name, us, gf, gf / us * 1e3))
    ctx.close()


if __name__ == "__main__":
    main()
