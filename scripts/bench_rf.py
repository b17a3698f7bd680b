#!/usr/bin/env python3
"""The register-filter convolution (rcgan_conv2d_rf, csrc/conv_rf.hip) against the tile-per-tap kernels on the small-grid 3x3 layers:
parity (norm-relative difference of the outputs) and time, forward and data gradient, HIP events on the launch stream.
usage: python scripts/bench_rf.py [n]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import rcgan_amd  # noqa: E402,F401
from rcgan_amd import _lib as L  # noqa: E402
from rcgan_amd.runtime import Context  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    ctx = Context(0, "bf16", arena_bytes=4 << 30, ws_bytes=1 << 30)
    lib, h = ctx.lib, ctx.h
    P = C.c_void_p

    def timed(fn, reps=30):
        for _ in range(3):
            fn()
        ctx.event_record(0)
        for _ in range(reps):
            fn()
        ctx.event_record(1)
        return ctx.event_elapsed_ms(0, 1) * 1e3 / reps

    cases = [("D.Block.2.Conv1 16x16 128>128", 16, 128, 128), ("D.Block.3 8x8 128>128", 8, 128, 128),
             ("G.Block.1.Conv2 8x8 256>256", 8, 256, 256), ("G.Block.2.Conv2 16x16 256>256", 16, 256, 256)]
    print("%-34s %9s %9s %9s %9s   %s" % ("layer (n = %d)" % n, "fwd us", "rf fwd", "dgrad us", "rf dgrad", "norm-rel diff fwd / dgrad"))
    bad = 0
    for name, hw, cin, cout in cases:
        desc = L.ConvDesc(n, hw, hw, cin, cout, 3, 3, 1, L.BF16, L.CONV_IN_RELU)
        if not lib.rcgan_conv_rf_ok(C.byref(desc)):
            continue
        x, dy = ctx.empty((n, hw, hw, cin)), ctx.empty((n, hw, hw, cout))
        ctx.check(lib.rcgan_rng_fill(h, x.size, x.dtype, 1, 0.0, 1.0, 7, None, P(x.ptr)))
        ctx.check(lib.rcgan_rng_fill(h, dy.size, dy.dtype, 1, 0.0, 1.0, 8, None, P(dy.ptr)))
        w = ctx.empty((3, 3, cin, cout), L.F32)
        bias = ctx.empty((cout,), L.F32)
        ctx.check(lib.rcgan_rng_fill(h, w.size, L.F32, 1, 0.0, 0.03, 9, None, P(w.ptr)))
        ctx.check(lib.rcgan_rng_fill(h, bias.size, L.F32, 1, 0.0, 0.1, 10, None, P(bias.ptr)))
        pr = ctx.arena.alloc(lib.rcgan_conv_prepared_bytes(C.byref(desc)))
        ctx.check(lib.rcgan_conv_prepare(h, C.byref(desc), P(w.ptr), None, P(pr)))
        frag = ctx.arena.alloc(lib.rcgan_conv_rf_fragment_bytes(C.byref(desc)))
        ctx.check(lib.rcgan_conv_rf_prepare(h, 1, C.byref(desc), (C.c_void_p * 1)(pr), (C.c_void_p * 1)(frag)))
        y0, y1, dx0, dx1 = ctx.empty(dy.shape), ctx.empty(dy.shape), ctx.empty(x.shape), ctx.empty(x.shape)
        f0 = lambda: ctx.check(lib.rcgan_conv2d_fwd(h, C.byref(desc), P(x.ptr), P(pr), P(bias.ptr), P(y0.ptr)))
        f1 = lambda: ctx.check(lib.rcgan_conv2d_rf(h, C.byref(desc), 0, P(x.ptr), P(frag), P(bias.ptr), None, None, P(y1.ptr)))
        b0 = lambda: ctx.check(lib.rcgan_conv2d_bwd_data(h, C.byref(desc), P(dy.ptr), P(pr), P(x.ptr), P(dx0.ptr), P(ctx.ws_ptr), ctx.ws_bytes))
        b1 = lambda: ctx.check(lib.rcgan_conv2d_rf(h, C.byref(desc), 1, P(dy.ptr), P(frag), None, P(x.ptr), None, P(dx1.ptr)))
        t = [timed(f) for f in (f0, f1, b0, b1)]
        if os.environ.get("STAMPS"):
            import torch
            nwg = n * (hw // 8) * (cout // (128 if cin == 128 else 64))
            st = torch.zeros(nwg * 16, dtype=torch.int64, device=ctx.device)
            ctx.check(lib.rcgan_debug_stamps(h, P(st.data_ptr())))
            f1()
            ctx.sync()
            ctx.check(lib.rcgan_debug_stamps(h, None))
            tt = st.cpu().numpy().reshape(nwg, 16).astype(np.float64)
            ng = hw * 8 // 64
            d = np.diff(tt[:, :3 + 3 * ng], axis=1) / 2270.0       # s_memtime ticks at ~2.27 GHz
            print("   per-workgroup us: issue %.2f, patch wait %.2f" % (d[:, 0].mean(), d[:, 1].mean()), end="")
            for g in range(ng):
                print(" | group %d: K loop %.2f, exchange %.2f, epilogue %.2f" % (g, d[:, 2 + 3 * g].mean(), d[:, 3 + 3 * g].mean(), d[:, 4 + 3 * g].mean()), end="")
            print(" | total %.2f" % ((tt[:, 2 + 3 * ng] - tt[:, 0]).mean() / 2270.0))
        ctx.sync()
        rel = lambda a, b: float(np.linalg.norm(ctx.download(a).astype(np.float64) - ctx.download(b)) / np.linalg.norm(ctx.download(b).astype(np.float64)))
        print("%-34s %9.1f %9.1f %9.1f %9.1f   %.2e / %.2e" % (name, t[0], t[1], t[2], t[3], rel(y1, y0), rel(dx1, dx0)))
        # admission contract of rcgan_conv_rf_ok: whatever it admits runs within 1.2x of the tile kernels, both directions
        verdict = "ok" if (t[1] <= 1.2 * t[0] and t[3] <= 1.2 * t[2]) else "FAIL"
        print("   admitted shape within 1.2x of the tile kernel (fwd %.2fx, dgrad %.2fx): %s" % (t[1] / t[0], t[3] / t[2], verdict))
        bad += verdict != "ok"
    ctx.close()
    if bad:
        sys.exit("bench_rf: %d admitted shape(s) slower than 1.2x the tile kernel" % bad)


if __name__ == "__main__":
    main()
