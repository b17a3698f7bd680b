#!/usr/bin/env python3
"""Where a workgroup of the 256 x 256 convolution kernel spends its time (s_memtime stamps, rcgan_debug_stamps):
per workgroup  start -> table -> first K-tile landed -> K loop done -> stores issued -> stores complete,
and per CU the gap between one workgroup's last stamp and the next workgroup's first (dispatch + wave launch).
usage: python scripts/exp_p8_timeline.py [n] [cin] [k] [hw] [cout]     (defaults 320 256 3 32 256: the 256 x 256 kernel;
       e.g. 128 128 3 8 128 = a D.Block.3-6 convolution on the 64 x 64 K-split kernel, whose stamps are start, set-up done, first
       K-tile multiplied, K loop done, partial sums combined, stores complete)"""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 320
    cin = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    hw = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    cout = int(sys.argv[5]) if len(sys.argv) > 5 else 256
    ctx = Context(0, "bf16", arena_bytes=8 << 30, ws_bytes=1 << 30)
    lib, h = ctx.lib, ctx.h
    x = ctx.empty((n, hw, hw, cin))
    y = ctx.empty((n, hw, hw, cout))
    w = ctx.empty((k, k, cin, cout), L.F32)
    ctx.check(lib.rcgan_rng_fill(h, x.size, x.dtype, 1, 0.0, 1.0, 7, None, C.c_void_p(x.ptr)))
    ctx.check(lib.rcgan_rng_fill(h, w.size, L.F32, 1, 0.0, 0.05, 9, None, C.c_void_p(w.ptr)))
    desc = L.ConvDesc(n, hw, hw, cin, cout, k, k, 1, L.BF16, 0)
    prep = ctx.arena.alloc(lib.rcgan_conv_prepared_bytes(C.byref(desc)))
    ctx.check(lib.rcgan_conv_prepare(h, C.byref(desc), C.c_void_p(w.ptr), None, C.c_void_p(prep)))
    call = lambda: ctx.check(lib.rcgan_conv2d_fwd(h, C.byref(desc), C.c_void_p(x.ptr), C.c_void_p(prep), None, C.c_void_p(y.ptr)))
    tiles = int(os.environ.get("TILES", n * hw * hw // 256 * (cout // 256) if cout % 256 == 0 and n * hw * hw >= 51200 else (n * hw * hw + 63) // 64 * (cout // 64)))
    stamps = torch.zeros(tiles * 8, dtype=torch.int64, device=ctx.device)
    for _ in range(3):
        call()
    ctx.check(lib.rcgan_debug_stamps(h, C.c_void_p(stamps.data_ptr())))
    call()
    ctx.sync()
    ctx.check(lib.rcgan_debug_stamps(h, None))
    s = stamps.cpu().numpy().reshape(tiles, 8)
    if os.environ.get("RAW"):
        np.set_printoptions(linewidth=200)
        print(s[:6]); print(s[-3:]); print("zeros per column", (s == 0).sum(0))
    t = s[:, :6].astype(np.float64)
    xcc = (s[:, 7] & 0xf).astype(int)
    # s_memtime is a per-XCD counter (different bases): only differences inside one XCD mean anything
    for xc in np.unique(xcc):
        t[xcc == xc] -= t[xcc == xc, 0].min()
    seg = np.diff(t, axis=1)
    names = ["setup (tap table / patch sources)", "first K-tile (load latency)", "K loop (%d K-tiles)" % (k * k * cin // 64), "epilogue issue", "store drain"]
    ctx.event_record(0)
    for _ in range(10):
        call()
    ctx.event_record(1)
    us = ctx.event_elapsed_ms(0, 1) * 1e2
    span = t[:, 5].max()
    print("n=%d cin=%d k=%d: %d tiles; kernel %.1f us per launch (events, back to back); longest XCD span %.0f ticks" % (n, cin, k, tiles, us, span))
    tick = 1.0 / 2270.0       # us per tick if the counter runs at ~2.27 GHz (K loop of 36 K-tiles = 58 us = 132k ticks)
    for i, nm in enumerate(names):
        print("  %-34s mean %8.0f ticks (%6.2f us @2.27GHz)   min %8.0f   max %8.0f" % (nm, seg[:, i].mean(), seg[:, i].mean() * tick, seg[:, i].min(), seg[:, i].max()))
    tot = t[:, 5] - t[:, 0]
    print("  %-34s mean %8.0f ticks (%6.2f us)   min %8.0f   max %8.0f" % ("workgroup total", tot.mean(), tot.mean() * tick, tot.min(), tot.max()))
    cu = xcc * 65536 + ((s[:, 6] >> 8) & 0xffff)               # XCC + (CU, SH, SE) bits of HW_ID
    gaps, per_cu = [], []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        idx = idx[np.argsort(t[idx, 0])]
        per_cu.append(len(idx))
        for a, b in zip(idx[:-1], idx[1:]):
            gaps.append(t[b, 0] - t[a, 5])
    print("  distinct (XCC, HW_ID>>8) = %d; workgroups per CU min %d max %d" % (len(per_cu), min(per_cu), max(per_cu)))
    if gaps:
        gaps = np.array(gaps)
        print("  gap between consecutive workgroups on a CU: mean %.0f ticks (%.2f us), min %.0f, max %.0f" % (gaps.mean(), gaps.mean() * tick, gaps.min(), gaps.max()))
    for xc in np.unique(xcc):
        m = xcc == xc
        print("  XCD %d: %4d workgroups, first start spread %7.0f ticks, last end %8.0f ticks, mean K loop %8.0f" %
              (xc, m.sum(), t[m, 0].min() if m.sum() == 0 else np.sort(t[m, 0])[min(31, m.sum() - 1)], t[m, 5].max(), seg[m, 2].mean()))
    ctx.close()


if __name__ == "__main__":
    main()
