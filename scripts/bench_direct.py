#!/usr/bin/env python3
"""Micro-benchmark of the fp32 gather-GEMM conv path on the MNIST RCGAN layer shapes (HIP events on the launch stream).

usage: python scripts/bench_direct.py [B] [only-substring] [reps]
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    ctx = Context(0, "f32", arena_bytes=4 << 30, ws_bytes=1 << 30)
    lib, h = ctx.lib, ctx.h
    shapes = [
        # name, n, h, w, cin, cout, k, stride   (conv geometry; a transposed conv runs the data gradient as its forward)
        ("g_h2 deconv 7>14 138>128 (as conv 14>7 128>138)", B, 14, 14, 128, 138, 5, 2),
        ("g_h3 deconv 14>28 138>1 (as conv 28>14 1>138)", B, 28, 28, 1, 138, 5, 2),
        ("d_h0 conv 28>14 1>64", B, 28, 28, 1, 64, 5, 2),
        ("d_h1 conv 14>7 64>64", B, 14, 14, 64, 64, 5, 2),
        ("d_h2 conv 7>4 64>64", B, 7, 7, 64, 64, 5, 2),
        ("d_h3 conv 4>2 64>64", B, 4, 4, 64, 64, 5, 2),
    ]
    print("%-52s %10s %10s %10s   (TFLOP/s: fwd dgrad wgrad)" % ("layer", "fwd us", "dgrad us", "wgrad us"))
    for name, n, hh, ww, cin, cout, k, s in shapes:
        if only and only not in name:
            continue
        ctx.new_step()
        oh, ow = (hh + s - 1) // s, (ww + s - 1) // s
        x = ctx.empty((n, hh, ww, cin))
        y = ctx.empty((n, oh, ow, cout))
        dx = ctx.empty((n, hh, ww, cin))
        w = ctx.empty((k, k, cin, cout), L.F32)
        dw = ctx.empty((k, k, cin, cout), L.F32)
        for t in (x, y):
            ctx.check(lib.rcgan_rng_fill(h, t.size, t.dtype, 1, 0.0, 1.0, 7, None, C.c_void_p(t.ptr)))
        ctx.check(lib.rcgan_rng_fill(h, w.size, L.F32, 1, 0.0, 0.05, 9, None, C.c_void_p(w.ptr)))
        desc = L.ConvDesc(n, hh, ww, cin, cout, k, k, s, L.F32, 0)
        prep = ctx.arena.alloc(lib.rcgan_conv_prepared_bytes(C.byref(desc)))
        ctx.check(lib.rcgan_conv_prepare(h, C.byref(desc), C.c_void_p(w.ptr), None, C.c_void_p(prep)))
        flops = 2.0 * n * oh * ow * k * k * cin * cout
        res = []
        for which in range(3):
            def call():
                if which == 0:
                    ctx.check(lib.rcgan_conv2d_fwd(h, C.byref(desc), C.c_void_p(x.ptr), C.c_void_p(prep), None, C.c_void_p(y.ptr)))
                elif which == 1:
                    ctx.check(lib.rcgan_conv2d_bwd_data(h, C.byref(desc), C.c_void_p(y.ptr), C.c_void_p(prep), None,
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
        print("%-52s %10.1f %10.1f %10.1f   %7.1f %7.1f %7.1f" % (name, res[0], res[1], res[2],
                                                               flops / res[0] / 1e6, flops / res[1] / 1e6, flops / res[2] / 1e6))


if __name__ == "__main__":
    main()
