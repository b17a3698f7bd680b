#!/usr/bin/env python3
"""The 8x8 discriminator stage (D.Block.3-6): one fused launch (rcgan_dtrunk) vs eight convolution launches, forward and
data gradient, HIP events on the launch stream.   usage: python scripts/bench_trunk.py [n]"""
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
    shape = (n, 8, 8, 128)
    x = ctx.empty(shape)
    ctx.check(lib.rcgan_rng_fill(h, x.size, x.dtype, 1, 0.0, 1.0, 7, None, P(x.ptr)))
    desc = L.ConvDesc(n, 8, 8, 128, 128, 3, 3, 1, L.BF16, L.CONV_IN_RELU)
    preps, outs, gouts = [], [ctx.empty(shape) for _ in range(8)], [ctx.empty(shape) for _ in range(8)]
    for i in range(8):
        w = ctx.empty((3, 3, 128, 128), L.F32)
        ctx.check(lib.rcgan_rng_fill(h, w.size, L.F32, 1, 0.0, 0.03, 9 + i, None, P(w.ptr)))
        pr = ctx.arena.alloc(lib.rcgan_conv_prepared_bytes(C.byref(desc)))
        ctx.check(lib.rcgan_conv_prepare(h, C.byref(desc), P(w.ptr), None, P(pr)))
        preps.append(pr)
    arr = lambda ps: (C.c_void_p * 8)(*ps)
    frag = ctx.arena.alloc(lib.rcgan_dtrunk_fragment_bytes())
    prep = lambda: ctx.check(lib.rcgan_dtrunk_prepare(h, arr(preps), P(frag)))
    prep()
    fwd = lambda: ctx.check(lib.rcgan_dtrunk(h, n, 0, P(x.ptr), P(frag), None, None, arr([o.ptr for o in outs])))
    masks = [outs[6], outs[5], outs[4], outs[3], outs[2], outs[1], outs[0], x]
    bwd = lambda: ctx.check(lib.rcgan_dtrunk(h, n, 1, P(x.ptr), P(frag), None, arr([m.ptr for m in masks]), arr([o.ptr for o in gouts])))

    def layerwise_fwd():
        t = x
        for k in range(4):
            ctx.check(lib.rcgan_conv2d_fwd(h, C.byref(desc), P(t.ptr), P(preps[2 * k]), None, P(outs[2 * k].ptr)))
            ctx.check(lib.rcgan_conv2d_fwd_residual(h, C.byref(desc), P(outs[2 * k].ptr), P(preps[2 * k + 1]), None, P(t.ptr), P(outs[2 * k + 1].ptr)))
            t = outs[2 * k + 1]

    def layerwise_bwd():
        for k in range(8):
            ctx.check(lib.rcgan_conv2d_bwd_data(h, C.byref(desc), P(x.ptr), P(preps[k]), P(outs[k].ptr), P(gouts[k].ptr), P(ctx.ws_ptr), ctx.ws_bytes))
    if os.environ.get("STAMPS"):
        import torch
        st = torch.zeros(n * 24, dtype=torch.int64, device=ctx.device)
        fwd(); fwd()
        ctx.check(lib.rcgan_debug_stamps(h, P(st.data_ptr())))
        fwd()
        ctx.sync()
        ctx.check(lib.rcgan_debug_stamps(h, None))
        t = st.cpu().numpy().reshape(n, 24).astype(np.float64)
        d = np.diff(t[:, :18], axis=1) / 2270.0
        print("per-workgroup segments, us @2.27 GHz (mean over %d workgroups): prologue %.2f" % (n, d[:, 0].mean()))
        for ly in range(8):
            print("  layer %d: K loop %.2f  epilogue+barrier %.2f" % (ly, d[:, 1 + 2 * ly].mean(), d[:, 2 + 2 * ly].mean()))
        print("  total %.2f" % ((t[:, 17] - t[:, 0]).mean() / 2270.0))
    reps = 30
    for name, fn in [("fragment-major filter copy", prep), ("fused forward", fwd), ("8 launches forward", layerwise_fwd), ("fused backward", bwd), ("8 launches data gradient", layerwise_bwd)]:
        fn(); fn()
        ctx.event_record(0)
        for _ in range(reps):
            fn()
        ctx.event_record(1)
        us = ctx.event_elapsed_ms(0, 1) * 1e3 / reps
        print("n=%d %-26s %8.1f us   (%.0f TFLOP/s)" % (n, name, us, 8 * 2.0 * n * 64 * 1152 * 128 / us / 1e6))
    ctx.close()


if __name__ == "__main__":
    main()
