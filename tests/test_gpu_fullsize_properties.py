"""Size-independent properties of the conv kernels at BASELINE's full layer sizes (where the numpy oracle would take
minutes per layer): a convolution is bilinear in (input, filter), so for any x, W, dy

        < conv(x; W), dy >  =  < x, conv_bwd_data(dy; W) >  =  < W, conv_bwd_weight(x, dy) >

(the data gradient is the adjoint in x, the filter gradient the adjoint in W).  One identity checks the forward, the data
gradient and the filter gradient of a layer against each other through the C ABI, on exactly the launch configurations
the benchmark runs (256x256 / 256x128 eight-wave tiles, 64x64 K-split tiles, the three-tap filter gradient, the fp32
gather GEMM with its parity-class transposed convs).  dy is the layer's own output, so all three numbers equal |y|^2 and
a wrong tap, tile edge or K-split shows up at the 1e-1 level instead of drowning in a random-sign sum.  Inner products
are accumulated in float64 on the host.  bf16 / fp16: y and dx are rounded once to 16 bits (relative 2^-9 / 2^-12 per
element, independent signs), which averages down over >= 10^6 terms to ~1e-6 (measured 2e-7 .. 2e-6); the filter gradient
is fp32.  Tolerance: 1e-4 (bf16), 2e-5 (fp16), 1e-6 (fp32) of |y|^2."""
import ctypes as C

import numpy as np
import pytest

from tests.gpu_util import make_ctx

pytestmark = pytest.mark.gpu

# name, dtype, n, h, w, cin, cout, k, stride, flags("up"/"relu"/"")
CASES = [
    ("cfg3 G.Block3.Conv2 32x32 256>256 n=128 (p8)", "bf16", 128, 32, 32, 256, 256, 3, 1, ""),
    ("cfg3 G.Block3.Conv1 16>32 upsample 256>256 n=128 (p8)", "bf16", 128, 32, 32, 256, 256, 3, 1, "up"),
    ("cfg3 G.Block3.Shortcut 1x1 upsample n=128 (p8)", "bf16", 128, 32, 32, 256, 256, 1, 1, "up"),
    ("cfg3 D.Block1.Conv2 32x32 128>128 n=128 (p8n)", "bf16", 128, 32, 32, 128, 128, 3, 1, ""),
    ("cfg3 D.Block3.Conv1 8x8 128>128 n=128 (64x64 K-split)", "bf16", 128, 8, 8, 128, 128, 3, 1, ""),
    ("cfg3 G.Block1.Conv1 4>8 upsample 1024>256 n=128", "bf16", 128, 8, 8, 1024, 256, 3, 1, "up"),
    ("cfg3 D.Block1.Conv1 32x32 3>128 n=128 (image end)", "bf16", 128, 32, 32, 3, 128, 3, 1, ""),
    ("cfg3 G.Output 32x32 256>3 n=128 (image end)", "bf16", 128, 32, 32, 256, 3, 3, 1, ""),
    ("cfg5 G.Block3.Conv2 n=256 fp16", "f16", 256, 32, 32, 256, 256, 3, 1, ""),
    ("cfg2 MNIST g_h2 as conv 14>7 128>138 5x5 s2 n=256 fp32", "f32", 256, 14, 14, 128, 138, 5, 2, ""),
    ("cfg2 MNIST g_h3 as conv 28>14 1>138 5x5 s2 n=256 fp32", "f32", 256, 28, 28, 1, 138, 5, 2, ""),
    ("cfg2 MNIST d_h1 conv 14>7 64>64 5x5 s2 n=256 fp32", "f32", 256, 14, 14, 64, 64, 5, 2, ""),
]


@pytest.fixture(scope="module", params=["bf16", "f16", "f32"])
def dev(request):
    ctx = make_ctx(request.param, arena=6 << 30)
    yield ctx, request.param
    ctx.close()


def _dot(a, b):
    return float(np.dot(np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()))


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_adjoint_identity(dev, case):
    from rcgan_amd import _lib as L
    ctx, mode = dev
    name, dtype, n, h, w, cin, cout, k, s, fl = case
    if dtype != mode:
        pytest.skip("runs in the %s context" % dtype)
    lib, hd = ctx.lib, ctx.h
    ctx.new_step()
    up = fl == "up"
    flags = L.CONV_IN_UPSAMPLE2X if up else 0
    hs, ws = (h // 2, w // 2) if up else (h, w)
    oh, ow = (h + s - 1) // s, (w + s - 1) // s
    x = ctx.empty((n, hs, ws, cin))
    y = ctx.empty((n, oh, ow, cout))
    dy = ctx.empty((n, oh, ow, cout))
    dx = ctx.empty((n, hs, ws, cin))
    wt = ctx.empty((k, k, cin, cout), L.F32)
    dw = ctx.empty((k, k, cin, cout), L.F32)
    ctx.check(lib.rcgan_rng_fill(hd, x.size, x.dtype, 1, 0.0, 1.0, 11, None, C.c_void_p(x.ptr)))
    ctx.check(lib.rcgan_rng_fill(hd, wt.size, L.F32, 1, 0.0, 0.05, 13, None, C.c_void_p(wt.ptr)))
    desc = L.ConvDesc(n, h, w, cin, cout, k, k, s, ctx.act_dtype, flags)
    prep = ctx.arena.alloc(lib.rcgan_conv_prepared_bytes(C.byref(desc)))
    ctx.check(lib.rcgan_conv_prepare(hd, C.byref(desc), C.c_void_p(wt.ptr), None, C.c_void_p(prep)))
    ctx.check(lib.rcgan_conv2d_fwd(hd, C.byref(desc), C.c_void_p(x.ptr), C.c_void_p(prep), None, C.c_void_p(y.ptr)))
    import torch
    with torch.cuda.stream(ctx.stream):
        ctx.view(dy).copy_(ctx.view(y))                     # dy := y
    ctx.check(lib.rcgan_conv2d_bwd_data(hd, C.byref(desc), C.c_void_p(dy.ptr), C.c_void_p(prep), None, C.c_void_p(dx.ptr),
                                        C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
    ctx.check(lib.rcgan_conv2d_bwd_weight(hd, C.byref(desc), C.c_void_p(x.ptr), C.c_void_p(dy.ptr), C.c_void_p(dw.ptr), None, 0,
                                          C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
    W = ctx.download(wt)
    # the matrix-core path multiplies with the filter rounded to 16 bits: use the same filter on the <W, dW> side
    mfma = mode != "f32" and s == 1 and k in (1, 3)
    if mfma:
        t = torch.from_numpy(W)
        W = (t.bfloat16() if mode == "bf16" else t.half()).float().numpy()
    X, Y = ctx.download(x), ctx.download(y)
    yy = _dot(Y, Y)
    b = _dot(X, ctx.download(dx))
    c = _dot(W, ctx.download(dw))
    tol = {"f32": 1e-6, "f16": 2e-5, "bf16": 1e-4}[mode]
    assert np.isfinite([yy, b, c]).all() and yy > 0
    # |y|^2 is at its expected scale (unit-variance x, filter std 0.05): the forward did multiply something
    expect = float(n * oh * ow * cout) * 0.05 ** 2 * k * k * cin * (0.5 if s == 2 else 1.0)
    assert 0.3 * expect < yy < 2.0 * expect, (name, yy, expect)
    assert abs(yy - b) <= tol * yy, (name, "forward vs data gradient", yy, b)
    assert abs(yy - c) <= tol * yy, (name, "forward vs filter gradient", yy, c)


# ---- conditional batch norm at full size: numpy (float64) restates it in seconds, and the backward has two exact
#      invariants per channel: sum(dx) = 0 and sum(dx * xhat) = 0 (the batch statistics absorb both directions) -------
BN_CASES = [
    ("cfg3 G.Block3 condBN n=128 32x32x256 bf16", "bf16", 128, 1024, 256, 10),
    ("cfg5 G.Block3 condBN n=256 32x32x256 fp16", "f16", 256, 1024, 256, 10),
    ("cfg2 MNIST g_bn2 n=256 14x14x128 fp32", "f32", 256, 196, 128, 1),
]


@pytest.mark.parametrize("case", BN_CASES, ids=[c[0] for c in BN_CASES])
def test_batch_norm_full_size(dev, case):
    from rcgan_amd import _lib as L
    ctx, mode = dev
    name, dtype, n, rps, c, nl = case
    if dtype != mode:
        pytest.skip("runs in the %s context" % dtype)
    lib, hd = ctx.lib, ctx.h
    ctx.new_step()
    rows = n * rps
    rs = np.random.RandomState(5)
    x, y, dy, dx = (ctx.empty((rows, c)) for _ in range(4))
    ctx.check(lib.rcgan_rng_fill(hd, x.size, x.dtype, 1, 1.5, 3.0, 21, None, C.c_void_p(x.ptr)))
    ctx.check(lib.rcgan_rng_fill(hd, dy.size, dy.dtype, 1, 0.0, 1.0, 22, None, C.c_void_p(dy.ptr)))
    labels = rs.randint(nl, size=n).astype(np.int32)
    gamma = (1.0 + 0.3 * rs.randn(nl, c)).astype(np.float32)
    beta = (0.5 * rs.randn(nl, c)).astype(np.float32)
    d_lab = ctx.upload(labels) if nl > 1 else None
    d_g, d_b = ctx.upload(gamma, dtype=L.F32), ctx.upload(beta, dtype=L.F32)
    mean, rstd = ctx.empty((c,), L.F32), ctx.empty((c,), L.F32)
    dgam, dbet = ctx.zeros((nl, c), L.F32), ctx.zeros((nl, c), L.F32)
    eps = 1e-5
    lp = C.c_void_p(d_lab.ptr) if d_lab is not None else None
    ws, wsb = C.c_void_p(ctx.ws_ptr), ctx.ws_bytes
    ctx.check(lib.rcgan_bn_stats(hd, rows, c, ctx.act_dtype, C.c_void_p(x.ptr), eps, C.c_void_p(mean.ptr), C.c_void_p(rstd.ptr),
                                 None, None, 0.9, ws, wsb))
    ctx.check(lib.rcgan_bn_apply_fwd(hd, n, rps, c, nl, ctx.act_dtype, C.c_void_p(x.ptr), lp, C.c_void_p(d_g.ptr), C.c_void_p(d_b.ptr),
                                     C.c_void_p(mean.ptr), C.c_void_p(rstd.ptr), L.ACT_NONE, C.c_void_p(y.ptr), ws, wsb))
    ctx.check(lib.rcgan_bn_bwd(hd, n, rps, c, nl, ctx.act_dtype, C.c_void_p(x.ptr), C.c_void_p(y.ptr), C.c_void_p(dy.ptr), lp,
                               C.c_void_p(d_g.ptr), C.c_void_p(mean.ptr), C.c_void_p(rstd.ptr), L.ACT_NONE,
                               C.c_void_p(dx.ptr), 0, C.c_void_p(dgam.ptr), C.c_void_p(dbet.ptr), 0, ws, wsb))
    X = ctx.download(x).astype(np.float64)
    m_ref, v_ref = X.mean(0), X.var(0)
    r_ref = 1.0 / np.sqrt(v_ref + eps)
    tol = {"f32": 2e-5, "f16": 2e-3, "bf16": 1e-2}[mode]
    assert np.abs(ctx.download(mean) - m_ref).max() <= 1e-5 * np.abs(m_ref).max() + 1e-6
    assert np.abs(ctx.download(rstd) / r_ref - 1).max() <= 2e-5
    xhat = (X - m_ref) * r_ref
    lab_rows = np.repeat(labels, rps)
    y_ref = gamma[lab_rows].astype(np.float64) * xhat + beta[lab_rows]
    Y = ctx.download(y)
    assert np.abs(Y - y_ref).max() <= tol * np.abs(y_ref).max(), name
    DY = ctx.download(dy).astype(np.float64)
    DX = ctx.download(dx).astype(np.float64)
    # parameter gradients against the float64 restatement
    dg_ref = np.zeros((nl, c))
    db_ref = np.zeros((nl, c))
    np.add.at(dg_ref, lab_rows, DY * xhat)
    np.add.at(db_ref, lab_rows, DY)
    assert np.abs(ctx.download(dgam) - dg_ref).max() <= 2e-4 * np.abs(dg_ref).max(), name
    assert np.abs(ctx.download(dbet) - db_ref).max() <= 2e-4 * np.abs(db_ref).max(), name
    # data gradient: the float64 formula, and the two invariants
    g_rows = gamma[lab_rows].astype(np.float64) * DY
    dx_ref = r_ref * (g_rows - g_rows.mean(0) - xhat * (g_rows * xhat).mean(0))
    assert np.abs(DX - dx_ref).max() <= tol * np.abs(dx_ref).max(), name
    size = np.sqrt((DX ** 2).sum(0) * rows)          # |dx|_2 * sqrt(rows) bounds |sum dx| per channel
    inv_tol = {"f32": 1e-5, "f16": 2e-4, "bf16": 1e-3}[mode]
    assert (np.abs(DX.sum(0)) <= inv_tol * size).all(), name
    assert (np.abs((DX * xhat).sum(0)) <= inv_tol * size).all(), name
