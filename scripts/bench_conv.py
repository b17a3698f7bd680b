#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels on the CIFAR RCGAN layer shapes (HIP events on the launch stream).

usage: python scripts/bench_conv.py [B]      (B = per-GPU critic batch, default 64)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import rcgan_amd  # noqa: E402,F401
from rcgan_amd import _lib as L  # noqa: E402
from rcgan_amd.runtime import Context  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    ctx = Context(0, "bf16", arena_bytes=8 << 30, ws_bytes=2 << 30)
    lib, h = ctx.lib, ctx.h
    shapes = [
        # name, n, h, w, cin, cout, k, flags
        ("D1.Conv2   32x32 128>128", 2 * B, 32, 32, 128, 128, 3, L.CONV_IN_RELU),
        ("D2.Conv1   16x16 128>128", 2 * B, 16, 16, 128, 128, 3, L.CONV_IN_RELU),
        ("D2.Short   16x16 128>128 1x1", 2 * B, 16, 16, 128, 128, 1, 0),
        ("D3.Conv     8x8  128>128", 2 * B, 8, 8, 128, 128, 3, L.CONV_IN_RELU),
        ("G1.Conv1    8x8 1024>256 up", 2 * B, 8, 8, 1024, 256, 3, L.CONV_IN_UPSAMPLE2X),
        ("G1.Conv2    8x8  256>256", 2 * B, 8, 8, 256, 256, 3, 0),
        ("G1.Short    8x8 1024>256 1x1 up", 2 * B, 8, 8, 1024, 256, 1, L.CONV_IN_UPSAMPLE2X),
        ("G2.Conv1   16x16 256>256 up", 2 * B, 16, 16, 256, 256, 3, L.CONV_IN_UPSAMPLE2X),
        ("G2.Conv2   16x16 256>256", 2 * B, 16, 16, 256, 256, 3, 0),
        ("G3.Conv1   32x32 256>256 up", 2 * B, 32, 32, 256, 256, 3, L.CONV_IN_UPSAMPLE2X),
        ("G3.Conv2   32x32 256>256", 2 * B, 32, 32, 256, 256, 3, 0),
        ("G3.Short   32x32 256>256 1x1 up", 2 * B, 32, 32, 256, 256, 1, L.CONV_IN_UPSAMPLE2X),
        ("G3.Conv2   32x32 256>256 (n=B)", B, 32, 32, 256, 256, 3, 0),
        # image-end layers (conv_image.hip): bound by streaming the big-channel tensor, the TFLOP/s columns mean nothing
        ("D1.Conv1   32x32   3>128", 2 * B, 32, 32, 3, 128, 3, 0),
        ("D1.Short   16x16   3>128 1x1", 2 * B, 16, 16, 3, 128, 1, 0),
        ("G.Output   32x32 256>3", 2 * B, 32, 32, 256, 3, 3, L.CONV_IN_RELU),
    ]
    only = os.environ.get("BENCH_CONV_ONLY")
    if only:
        shapes = [s for s in shapes if only in s[0]]
    reps = 20
    print("%-36s %10s %10s %10s   (TFLOP/s: fwd dgrad wgrad)" % ("layer", "fwd us", "dgrad us", "wgrad us"))
    tot = [0.0, 0.0, 0.0]
    for name, n, hh, ww, cin, cout, k, flags in shapes:
        ctx.new_step()
        up = bool(flags & L.CONV_IN_UPSAMPLE2X)
        hs, ws = (hh // 2, ww // 2) if up else (hh, ww)
        x = ctx.empty((n, hs, ws, cin))
        y = ctx.empty((n, hh, ww, cout))
        dx = ctx.empty((n, hs, ws, cin))
        w = ctx.empty((k, k, cin, cout), L.F32)
        dw = ctx.empty((k, k, cin, cout), L.F32)
        for t in (x, y):
            ctx.check(lib.rcgan_rng_fill(h, t.size, t.dtype, 1, 0.0, 1.0, 7, None, C.c_void_p(t.ptr)))
        ctx.check(lib.rcgan_rng_fill(h, w.size, L.F32, 1, 0.0, 0.05, 9, None, C.c_void_p(w.ptr)))
        desc = L.ConvDesc(n, hh, ww, cin, cout, k, k, 1, L.BF16, flags)
        prep = ctx.arena.alloc(lib.rcgan_conv_prepared_bytes(C.byref(desc)))
        ctx.check(lib.rcgan_conv_prepare(h, C.byref(desc), C.c_void_p(w.ptr), None, C.c_void_p(prep)))
        flops = 2.0 * n * hh * ww * k * k * cin * cout
        res = []
        for which in range(3):
            def call():
                if which == 0:
                    ctx.check(lib.rcgan_conv2d_fwd(h, C.byref(desc), C.c_void_p(x.ptr), C.c_void_p(prep), None, C.c_void_p(y.ptr)))
                elif which == 1:
                    ctx.check(lib.rcgan_conv2d_bwd_data(h, C.byref(desc), C.c_void_p(y.ptr), C.c_void_p(prep), C.c_void_p(x.ptr),
                                                        C.c_void_p(dx.ptr), C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
                else:
                    ctx.check(lib.rcgan_conv2d_bwd_weight(h, C.byref(desc), C.c_void_p(x.ptr), C.c_void_p(y.ptr), C.c_void_p(dw.ptr),
                                                          None, 0, C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
            call()
            call()
            ctx.event_record(0)
            for _ in range(reps):
                call()
            ctx.event_record(1)
            us = ctx.event_elapsed_ms(0, 1) * 1e3 / reps
            res.append(us)
            tot[which] += us
        print("%-36s %10.1f %10.1f %10.1f   %7.0f %7.0f %7.0f" % (name, res[0], res[1], res[2],
                                                               flops / res[0] / 1e6, flops / res[1] / 1e6, flops / res[2] / 1e6))
    print("total us: fwd %.0f dgrad %.0f wgrad %.0f" % tuple(tot))
    ctx.close()


if __name__ == "__main__":
    main()
