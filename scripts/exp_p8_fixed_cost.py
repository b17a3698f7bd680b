#!/usr/bin/env python3
"""Fixed cost per workgroup round of the 256x256-tile convolution kernel: forward time of a 32x32 conv with Cout = 256 as a
function of the K-tile count (Cin x taps / 64) and of the number of rounds (n = 64: 256 tiles = one round of 256 CUs).
time = rounds x (fixed + per_ktile x KT): `fixed` is what a round pays outside its K loop (launch, tap table, first
K-tile's round trip, epilogue stores).  HIP events on the launch stream.
usage: python scripts/exp_p8_fixed_cost.py [quick]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import rcgan_amd  # noqa: E402,F401
from rcgan_amd import _lib as L  # noqa: E402
from rcgan_amd.runtime import Context  # noqa: E402


def main():
    ctx = Context(0, "bf16", arena_bytes=12 << 30, ws_bytes=1 << 30)
    lib, h = ctx.lib, ctx.h
    reps = 30
    rows = []
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"      # three 3x3 layers at n = 320: the per-K-tile slope only
    ns = (320,) if quick else (64, 128, 320)
    for n in ns:
        for k, cin in (((3, 128), (3, 256), (3, 512)) if quick else ((1, 256), (1, 512), (1, 1024), (3, 64), (3, 128), (3, 256), (3, 512))):
            ctx.new_step()
            x = ctx.empty((n, 32, 32, cin))
            y = ctx.empty((n, 32, 32, 256))
            w = ctx.empty((k, k, cin, 256), L.F32)
            ctx.check(lib.rcgan_rng_fill(h, x.size, x.dtype, 1, 0.0, 1.0, 7, None, C.c_void_p(x.ptr)))
            ctx.check(lib.rcgan_rng_fill(h, w.size, L.F32, 1, 0.0, 0.05, 9, None, C.c_void_p(w.ptr)))
            desc = L.ConvDesc(n, 32, 32, cin, 256, k, k, 1, L.BF16, 0)
            prep = ctx.arena.alloc(lib.rcgan_conv_prepared_bytes(C.byref(desc)))
            ctx.check(lib.rcgan_conv_prepare(h, C.byref(desc), C.c_void_p(w.ptr), None, C.c_void_p(prep)))
            call = lambda: ctx.check(lib.rcgan_conv2d_fwd(h, C.byref(desc), C.c_void_p(x.ptr), C.c_void_p(prep), None, C.c_void_p(y.ptr)))
            call(); call()
            ctx.event_record(0)
            for _ in range(reps):
                call()
            ctx.event_record(1)
            us = ctx.event_elapsed_ms(0, 1) * 1e3 / reps
            kt = k * k * cin // 64
            tiles = n * 1024 // 256
            fl = 2.0 * n * 1024 * k * k * cin * 256
            rows.append((n, k, cin, kt, tiles, us, fl / us / 1e6))
            print("n=%3d k=%d cin=%4d  KT=%3d tiles=%4d  %8.1f us  %7.0f TFLOP/s" % rows[-1], flush=True)
    for n in ns:
        r = [x for x in rows if x[0] == n]
        A = np.array([[1.0, x[3]] for x in r])
        b = np.array([x[5] for x in r])
        (fixed, per), *_ = np.linalg.lstsq(A, b, rcond=None)
        rounds = -(-r[0][4] // 256)
        print("n=%3d (%d rounds): time = %.1f us + %.2f us x KT  -> per round: fixed %.1f us, %.2f us per K-tile"
              % (n, rounds, fixed, per, fixed / rounds, per / rounds))
    ctx.close()


if __name__ == "__main__":
    main()
