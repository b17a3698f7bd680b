"""Per-op parity: every HIP kernel, called through the C ABI, against the numpy oracle on the same
seeded inputs.  fp32 activations: tolerance 2e-5 of the reference's scale (fp32 accumulation-order
noise only).  bf16 activations: inputs are rounded to bf16 on the host first, so what remains is fp32
accumulation order + one bf16 rounding of the stored result: tolerance 1e-2 (bf16 has 8 mantissa bits,
half-ulp 2^-9 = 0.2 %; sums of a few roundings stay well under 1 %)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import nn
from tests.gpu_util import HALF, FakeParam, assert_close, half_round, make_ctx

pytestmark = pytest.mark.gpu

TOL = {"f32": 2e-5, "bf16": 1e-2, "f16": 3e-3}     # fp16: 11 significand bits, half-ulp 2^-12 = 0.02 %


@pytest.fixture(scope="module", params=["f32", "bf16", "f16"])
def dev(request):
    ctx = make_ctx(request.param)
    yield ctx, request.param
    ctx.close()


def _prep(a, mode):
    return half_round(mode, a)


def test_selftest_layouts(dev):
    ctx, _ = dev
    assert ctx.uses_tr_read in (0, 1)
    print("ds_read_b64_tr_b16 path in use:", ctx.uses_tr_read)


CONV_CASES = [
    # n, h, w, cin, cout, k, stride, in_up, in_relu
    (2, 9, 7, 3, 5, 3, 1, False, False),
    (2, 28, 28, 1, 8, 5, 2, False, False),
    (3, 7, 7, 6, 4, 5, 2, False, True),
    (2, 4, 4, 5, 3, 5, 2, False, False),
    (2, 8, 8, 16, 8, 1, 1, False, False),
    (2, 8, 8, 4, 6, 3, 1, True, True),
    (2, 8, 8, 64, 64, 3, 1, False, False),      # MFMA 64x64 tile (bf16)
    (3, 5, 5, 64, 128, 3, 1, False, True),      # MFMA, M tail (75 pixels)
    (2, 16, 16, 128, 128, 3, 1, False, True),   # MFMA + MFMA filter gradient
    (1, 8, 8, 128, 256, 3, 1, True, False),     # MFMA with folded upsample
    (32, 8, 8, 256, 128, 3, 1, True, True),     # ... in its sub-pixel form (four summed 2x2 filters), 64 x 64 tiles with K-split 2
    (200, 8, 8, 64, 256, 3, 1, True, False),    # ... 64 x 64 tiles, 800 workgroups (no K-split)
    (8, 32, 32, 128, 128, 3, 1, True, True),    # ... and the sub-pixel filter gradient over a 16 x 16 low-resolution grid, input ReLU
    (6, 16, 16, 256, 256, 3, 1, True, False),   # ... 8 x 8 grid, 4 x 2 channel tiles, a ragged last pixel chunk
    (2, 8, 8, 256, 128, 1, 1, False, False),    # MFMA 1x1
    (5, 32, 32, 128, 128, 3, 1, False, True),   # MFMA 128x128 tile (M = 5120 -> 40 blocks < 384 -> 64 tile) ...
    (48, 32, 32, 128, 128, 3, 1, False, False), # ... and M = 49152 -> 384 blocks of 128x128
    (4, 32, 32, 3, 128, 3, 1, False, False),    # image-end kernels: D.Block.1.Conv1 (small reduction / small-side wgrad)
    (4, 32, 32, 3, 128, 1, 1, False, False),    # D.Block.1.Shortcut
    (3, 16, 16, 256, 3, 3, 1, False, False),    # G.Output (small output)
    (3, 10, 6, 128, 3, 3, 1, False, True),      # small output with folded input ReLU (generic dgrad because of the mask)
    (2, 8, 8, 3, 256, 3, 1, False, False),
    (2, 32, 32, 256, 3, 3, 1, False, False),    # G.Output at the image resolution (MFMA small-output kernel, 4 channel chunks)
    (3, 16, 16, 128, 3, 1, 1, False, False),    # 1x1 small output
    (2, 32, 8, 3, 256, 3, 1, False, False),     # W = 8 bands, 256-channel small reduction
    (5, 16, 32, 128, 2, 3, 1, False, True),     # 2 output channels, folded input ReLU
    (52, 32, 32, 64, 256, 3, 1, False, True),   # 208 workgroups of 256x256: the 8-wavefront four-phase kernel (fwd; dgrad stays 64->...)
    (50, 32, 32, 256, 256, 1, 1, False, False), # four-phase kernel forward AND data gradient (Cin = Cout = 256), M tail (51200 = 200 tiles)
    (52, 32, 32, 256, 256, 3, 1, False, True),  # 208 tiles: the halo-patch 256x256 kernel (conv_mfma8h.hip), forward AND data gradient, four 64-channel chunks
    (200, 16, 16, 128, 256, 3, 1, False, False),# ... its 16-pixel-wide form (a tile = one whole image), two chunks; dgrad on the 128-channel path
    (52, 32, 32, 256, 256, 3, 1, True, True),   # ... the sub-pixel form of an upsample-3x3 layer on it (G.Block.3.Conv1): 16 x 16 low-resolution images, four taps per chunk
    (13, 64, 64, 64, 256, 3, 1, True, False),   # ... and on 32-wide low-resolution images (one chunk: no patch hand-over)
    # the 256 x 128-tile sibling of the halo-patch kernel (conv_mfma_h8n_kernel, round 5; the dgrad of the 16-wide case above takes it too):
    (100, 16, 16, 256, 256, 3, 1, False, False),# ... 100 pixel tiles x 2 channel halves (G.Block.2.Conv2's shape), four chunks, forward AND data gradient
    (50, 32, 32, 128, 128, 3, 1, False, True),  # ... 32-wide, Cout = 128, input ReLU, two chunks, both directions
    (48, 32, 32, 128, 128, 3, 1, True, True),   # ... sub-pixel form over 16 x 16 low-resolution images (the shape of D.Block.1.Conv2's pooled data gradient)
    (13, 64, 64, 64, 128, 3, 1, True, False),   # ... over 32-wide low-resolution images, one chunk
    (64, 32, 32, 256, 256, 3, 1, True, False),  # ... its parity-plane (stride-2 16-tap) form for the DATA GRADIENT of an upsample-3x3 layer: G.Block.3.Conv1's shape, 16 plane chunks
    (30, 64, 64, 128, 128, 3, 1, True, True),   # ... the same over 32-wide planes (data gradient), and the two-phases-per-workgroup sub-pixel forward on 32-wide images
    # the nine-tap filter gradient (conv_wgrad9.hip; the 16- and 32-wide cases above take it too): 8-pixel-wide images (four image rows per K-step),
    (6, 8, 8, 64, 128, 3, 1, False, True),
    (3, 16, 8, 128, 128, 3, 1, False, False),   # ... H != W
    (8, 4, 8, 64, 128, 3, 1, False, False),     # ... an image = one K-step (its first and its last row in the same step)
    (4, 2, 16, 64, 256, 3, 1, False, True),     # ... two-row images, 16 wide
    (40, 8, 8, 128, 128, 3, 1, False, True),    # ... several ring revolutions per chunk (D.Block.3/4)
]


def _random_conv_cases(count=36, seed=2024):
    """Seeded sweep over shapes the fixed list does not pin down: odd / non-square sizes, channel counts that are not
    multiples of 2, 4 or 8 (scalar / 8-byte / 16-byte operand loads, runs of 8 that straddle two filter taps), 1x1 / 3x3 /
    5x5 filters at stride 1 and 2 (parity-class data gradient, narrow outputs), folded ReLU."""
    rs = np.random.RandomState(seed)
    chans = [1, 2, 3, 5, 8, 9, 10, 12, 16, 20, 33, 64, 70, 138]
    cases = []
    while len(cases) < count:
        k = int(rs.choice([1, 3, 5]))
        s = int(rs.choice([1, 2]))
        h, w = int(rs.randint(2, 15)), int(rs.randint(2, 15))
        cin, cout = int(rs.choice(chans)), int(rs.choice(chans))
        n = int(rs.randint(1, 5))
        relu = bool(rs.randint(2))
        if n * h * w * cin * cout * k * k > 3e7:
            continue
        cases.append((n, h, w, cin, cout, k, s, False, relu))
    return cases


# shapes of CONV_CASES the nine-tap filter-gradient kernel (conv_wgrad9.hip) can take: by default only big layers go to it (a size rule in
# the one-layer entry point and in the grouped one), so these run a second time with the rule switched off -- plain 8 / 16 / 32-wide layers,
# the upsample form over 8x8 and 16x16 low-resolution grids, with and without input ReLU, bias gradients included
WGRAD9_CASES = [c for c in CONV_CASES if c[5] == 3 and c[6] == 1 and c[3] % 64 == 0 and c[4] % 128 == 0 and
                (c[1] // (2 if c[7] else 1)) * (c[2] // (2 if c[7] else 1)) * c[0] % 128 == 0 and (c[2] // (2 if c[7] else 1)) in (8, 16, 32)]
assert len(WGRAD9_CASES) >= 10


@pytest.mark.parametrize("grouped", [False, True])
@pytest.mark.parametrize("case", WGRAD9_CASES)
def test_conv2d_fwd_bwd_nine_tap_forced(dev, case, grouped, monkeypatch):
    ctx, mode = dev
    if mode == "f32":
        pytest.skip("the nine-tap kernel is a 16-bit matrix-core kernel")
    monkeypatch.setenv("RCGAN_WGRAD9_MINWORK", "0")
    monkeypatch.setenv("RCGAN_WGRAD9_GROUP_MINWORK", "0")
    keep, ctx.group_wgrads = ctx.group_wgrads, grouped
    try:
        test_conv2d_fwd_bwd(dev, case)
    finally:
        ctx.group_wgrads = keep


# the parity-plane (stride-2 16-tap) form of the 256 x 128 halo kernel takes a layer only when its tiles fill the chip (190 workgroups): forced
# here on the data gradients of the upsample-3x3 cases -- 16- and 32-wide planes, one / two / four channel chunks, with and without input ReLU
GATHER_CASES = [c for c in CONV_CASES if c[7] and c[5] == 3 and c[3] % 128 == 0 and c[4] % 64 == 0 and (c[0] * c[1] * c[2] // 4) % 256 == 0 and c[2] in (32, 64)]
assert len(GATHER_CASES) >= 4


@pytest.mark.parametrize("case", GATHER_CASES)
def test_conv2d_fwd_bwd_plane_gather_forced(dev, case, monkeypatch):
    ctx, mode = dev
    if mode == "f32":
        pytest.skip("a 16-bit matrix-core kernel")
    monkeypatch.setenv("RCGAN_H8N_GATHER_MINBLK", "1")
    test_conv2d_fwd_bwd(dev, case)


@pytest.mark.parametrize("case", CONV_CASES + _random_conv_cases())
def test_conv2d_fwd_bwd(dev, case):
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    n, h, w, cin, cout, k, s, up, relu = case
    if mode == "f32" and n * h * w * cin > 2 ** 21:
        pytest.skip("large case only exercises the MFMA tiles (bf16)")
    rs = np.random.RandomState(hash(case) % 2 ** 31)
    hs, ws = (h // 2, w // 2) if up else (h, w)
    x = _prep(rs.randn(n, hs, ws, cin), mode)
    wgt = (rs.randn(k, k, cin, cout) / np.sqrt(k * k * cin)).astype(np.float32)
    b = rs.randn(cout).astype(np.float32)
    sigma = np.float32(1.7)
    ctx.new_step()
    xd = ctx.upload(x)
    xd.req = True
    wp, bp = FakeParam(ctx, wgt), FakeParam(ctx, b)
    sg = ctx.upload(np.array([sigma], np.float32), L.F32)
    W = O.Weight(ctx, wp.t, sg)
    y = O.conv2d(ctx, xd, W, bp.t, k, s, in_up=up, in_relu=relu)
    # oracle: same bf16-rounded filter the MFMA path uses (direct path keeps the filter in fp32)
    mf = mode in HALF and s == 1 and k in (1, 3) and cin % 64 == 0 and cout % 64 == 0
    w_eff = (wgt / sigma)
    w_eff = half_round(mode, w_eff) if mf else w_eff
    xin = np.maximum(x, 0) if relu else x
    xin = nn.upsample2(xin) if up else xin
    ref = nn.conv2d_fwd(xin.astype(np.float64), w_eff.astype(np.float64), s) + b
    assert_close(ctx.download(y), ref, TOL[mode], "conv fwd %s" % (case,))
    dy = _prep(rs.randn(*ref.shape), mode)
    y.grad = ctx.upload(dy)
    ctx.backward()
    dxin = nn.conv2d_bwd_input(dy.astype(np.float64), w_eff.astype(np.float64), xin.shape, s)
    if up:
        dxin = nn.upsample2_bwd(dxin)
    if relu:
        dxin = dxin * (x > 0)
    assert_close(ctx.download(xd.grad), dxin, TOL[mode], "conv dgrad %s" % (case,))
    dw_bar = nn.conv2d_bwd_filter(xin.astype(np.float64), dy.astype(np.float64), wgt.shape, s)
    assert_close(ctx.download(W.dwbar), dw_bar, 2e-4 if mode in HALF else TOL[mode], "conv wgrad %s" % (case,))
    assert_close(bp.grad(ctx), dy.astype(np.float64).sum(axis=(0, 1, 2)), 2e-4, "conv dbias %s" % (case,))


@pytest.mark.parametrize("case", [(4, 16, 16, 64, 128),        # 64 x 64 kernel
                                  (64, 32, 32, 64, 256),       # 256 x 256 kernel
                                  (64, 32, 32, 64, 128),       # 256 x 128 kernel
                                  (3, 8, 8, 6, 5)])            # no matrix-core path: explicit upsample + residual
def test_conv_residual_from_half_resolution(dev, case):
    """y = conv3x3(x) + upsample2(r) with r on the half-resolution grid read in place by the epilogue
    (RCGAN_CONV_RESID_UPSAMPLE2X: the up blocks' 1x1 shortcut evaluated before the upsample), and d r = the 2x2 sum of dy."""
    from rcgan_amd import ops as O
    ctx, mode = dev
    n, h, w, cin, cout = case
    if mode == "f32" and n * h * w * cin > 2 ** 21:
        pytest.skip("large case only exercises the MFMA tiles")
    rs = np.random.RandomState(n + cout)
    x = _prep(rs.randn(n, h, w, cin).astype(np.float32), mode)
    r = _prep(rs.randn(n, h // 2, w // 2, cout).astype(np.float32), mode)
    wgt = (rs.randn(3, 3, cin, cout) / np.sqrt(9 * cin)).astype(np.float32)
    b = rs.randn(cout).astype(np.float32)
    mf = mode in HALF and cin % 64 == 0 and cout % 64 == 0
    ctx.new_step()
    xd, rd = ctx.upload(x), ctx.upload(r)
    xd.req = rd.req = True
    wp, bp = FakeParam(ctx, wgt), FakeParam(ctx, b)
    y = O.conv2d(ctx, xd, O.Weight(ctx, wp.t), bp.t, 3, residual=rd, residual_up=True)
    w_eff = half_round(mode, wgt) if mf else wgt
    ref = nn.conv2d_fwd(x.astype(np.float64), w_eff.astype(np.float64), 1) + b + nn.upsample2(r.astype(np.float64))
    assert_close(ctx.download(y), ref, TOL[mode], "conv + upsampled residual %s" % (case,))
    dy = _prep(rs.randn(*ref.shape).astype(np.float32), mode)
    y.grad = ctx.upload(dy)
    ctx.backward()
    assert_close(ctx.download(rd.grad), nn.upsample2_bwd(dy.astype(np.float64)), TOL[mode], "d residual %s" % (case,))
    dx = nn.conv2d_bwd_input(dy.astype(np.float64), w_eff.astype(np.float64), x.shape, 1)
    assert_close(ctx.download(xd.grad), dx, TOL[mode], "dgrad beside the residual %s" % (case,))


@pytest.mark.parametrize("cin", [64, 6])
def test_conv_residual_block(dev, cin):
    """Identity-shortcut block x + conv2(relu(conv1(relu(x)))) with the sum folded into conv2's epilogue
    (gan_resnet.py:295-328): values and the gradient reaching x (shortcut + masked conv path, same buffer)."""
    from rcgan_amd import ops as O
    ctx, mode = dev
    rs = np.random.RandomState(40 + cin)
    x = _prep(rs.randn(3, 8, 8, cin), mode)
    w1 = (rs.randn(3, 3, cin, cin) / np.sqrt(9 * cin)).astype(np.float32)
    w2 = (rs.randn(3, 3, cin, cin) / np.sqrt(9 * cin)).astype(np.float32)
    ctx.new_step()
    xd = ctx.upload(x)
    xd.req = True
    p1, p2 = FakeParam(ctx, w1), FakeParam(ctx, w2)
    h = O.conv2d(ctx, xd, O.Weight(ctx, p1.t), None, 3, in_relu=True)
    y = O.conv2d(ctx, h, O.Weight(ctx, p2.t), None, 3, in_relu=True, residual=xd)
    mf = mode in HALF and cin % 64 == 0
    q = (lambda a: half_round(mode, a)) if mf else (lambda a: a)
    x64 = x.astype(np.float64)
    h_ref = nn.conv2d_fwd(np.maximum(x64, 0), q(w1).astype(np.float64))
    hq = half_round(mode, h_ref).astype(np.float64) if mode in HALF else h_ref
    ref = x64 + nn.conv2d_fwd(np.maximum(hq, 0), q(w2).astype(np.float64))
    assert_close(ctx.download(y), ref, TOL[mode] * 2, "residual conv fwd")
    dy = _prep(rs.randn(*ref.shape), mode)
    y.grad = ctx.upload(dy)
    ctx.backward()
    dy64 = dy.astype(np.float64)
    dh = nn.conv2d_bwd_input(dy64, q(w2).astype(np.float64), hq.shape) * (hq > 0)
    dhq = half_round(mode, dh).astype(np.float64) if mode in HALF else dh
    dx = dy64 + nn.conv2d_bwd_input(dhq, q(w1).astype(np.float64), x.shape) * (x64 > 0)
    assert_close(ctx.download(xd.grad), dx, TOL[mode] * 3, "residual conv dx")


def test_conv_accumulate_and_force_direct(dev):
    from rcgan_amd import ops as O
    ctx, mode = dev
    rs = np.random.RandomState(5)
    x = _prep(rs.randn(2, 8, 8, 64), mode)
    w1 = (rs.randn(1, 1, 64, 64) / 8).astype(np.float32)
    w3 = (rs.randn(3, 3, 64, 64) / 24).astype(np.float32)
    ctx.new_step()
    xd = ctx.upload(x)
    p1, p3 = FakeParam(ctx, w1), FakeParam(ctx, w3)
    t = O.conv2d(ctx, xd, O.Weight(ctx, p1.t), None, 1)
    t = O.conv2d(ctx, xd, O.Weight(ctx, p3.t), None, 3, accumulate_into=t)
    t2 = O.conv2d(ctx, xd, O.Weight(ctx, p1.t), None, 1, force_direct=True)
    t2 = O.conv2d(ctx, xd, O.Weight(ctx, p3.t), None, 3, accumulate_into=t2, force_direct=True)
    ref = nn.conv2d_fwd(x.astype(np.float64), w1.astype(np.float64)) + nn.conv2d_fwd(x.astype(np.float64), w3.astype(np.float64))
    assert_close(ctx.download(t2), ref, TOL[mode], "direct accumulate")
    assert_close(ctx.download(t), ref, 2e-2 if mode in HALF else TOL[mode], "mfma accumulate")


def _torch_conv_ref(x, w, up=False, relu=False):
    """fp32 CPU convolution (torch, SAME, stride 1) of NHWC x with HWIO w: the reference for shapes where the float64
    numpy oracle would take minutes.  Both operands are already rounded to the 16-bit activation format."""
    import torch.nn.functional as F
    xt = torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2)
    if relu:
        xt = xt.clamp_min(0)
    if up:
        xt = xt.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    wt = torch.from_numpy(np.ascontiguousarray(w)).permute(3, 2, 0, 1).contiguous()
    return F.conv2d(xt, wt, padding=w.shape[0] // 2).permute(0, 2, 3, 1).contiguous().numpy()


PERSISTENT_CASES = [
    # n, h, w, cin, cout, k, up, relu, extra           tiles of 256 pixels -> persistent workgroups (256 CUs)
    (400, 16, 16, 256, 256, 3, False, True, ""),        # 400 (XCD-swizzled): 144 workgroups walk 2 tiles, 112 walk 1; fwd + dgrad (mask)
    (393, 16, 16, 64, 256, 3, True, False, ""),         # 393 (not a multiple of 8: plain order), folded upsample
    (6201, 8, 8, 128, 256, 1, False, False, ""),        # 1551, the last one partial (64 of 256 pixels); 2 K-tiles per tile (minimum)
    (400, 16, 16, 256, 256, 1, False, False, "acc"),    # accumulate into an existing tensor (shortcut + conv, gan_resnet.py:328)
    (400, 16, 16, 256, 256, 1, False, True, "res"),     # residual added in the epilogue
    (1280, 8, 8, 256, 256, 3, False, False, ""),        # 320 -> the 256 x 128 kernel (1.25 rounds of 256 x 256 tiles)
    # sub-pixel form of the upsample-3x3 convolution (four 2x2 convolutions with summed filters over the low-resolution grid):
    # taken when a phase holds whole 256-pixel tiles (the case with n = 393 above does not and keeps the folded upsample)
    (400, 16, 16, 64, 256, 3, True, False, ""),         # 256 x 256 kernel, 100 tiles per phase
    (52, 32, 32, 128, 256, 3, True, True, "res"),       # ... with input ReLU and a residual operand (rows remapped to 2i+ph, 2j+pw)
    (400, 16, 16, 64, 128, 3, True, False, "acc"),      # 256 x 128 kernel, accumulating
    # ... and its data gradient: dx over the low-resolution grid from the 4 x 4 neighbourhood of dy (16 taps, stride 2)
    (100, 32, 32, 256, 64, 3, True, True, "dgrad"),     # 256 x 128 kernel (200 workgroups), ReLU mask in the epilogue
    (100, 32, 32, 512, 64, 3, True, False, "dgrad"),    # 256 x 256 kernel (200 workgroups)
    (40, 16, 16, 128, 64, 3, True, True, "dgrad"),      # 64 x 64 kernel with K-split 2
]


@pytest.mark.parametrize("case", PERSISTENT_CASES)
def test_conv_persistent_tiles(dev, case):
    """The persistent 256 x 256 kernel (one workgroup per CU walking several tiles in one K-tile stream: tile hand-over of
    the LDS-DMA cursor, tap tables of two tiles alive at once, stores draining under the next tile) against an fp32 CPU
    convolution, forward and -- where Cin is a multiple of 256 too -- data gradient."""
    from rcgan_amd import ops as O
    ctx, mode = dev
    if mode == "f32":
        pytest.skip("MFMA tile kernels run on 16-bit activations")
    n, h, w, cin, cout, k, up, relu, extra = case
    rs = np.random.RandomState(abs(hash(case)) % 2 ** 31)
    hs, ws = (h // 2, w // 2) if up else (h, w)
    x = _prep(rs.randn(n, hs, ws, cin).astype(np.float32), mode)
    wgt = half_round(mode, (rs.randn(k, k, cin, cout) / np.sqrt(k * k * cin)).astype(np.float32))
    b = rs.randn(cout).astype(np.float32)
    ctx.new_step()
    xd = ctx.upload(x)
    xd.req = True
    wp, bp = FakeParam(ctx, wgt), FakeParam(ctx, b)
    W = O.Weight(ctx, wp.t)
    ref = _torch_conv_ref(x, wgt, up, relu) + b
    if extra == "acc":
        t0 = _prep(rs.randn(n, h, w, cout).astype(np.float32), mode)
        y = O.conv2d(ctx, xd, W, bp.t, k, in_up=up, in_relu=relu, accumulate_into=ctx.upload(t0))
        ref = ref + t0
    elif extra == "res":
        r0 = _prep(rs.randn(n, h, w, cout).astype(np.float32), mode)
        y = O.conv2d(ctx, xd, W, bp.t, k, in_up=up, in_relu=relu, residual=ctx.upload(r0))
        ref = ref + r0
    else:
        y = O.conv2d(ctx, xd, W, bp.t, k, in_up=up, in_relu=relu)
    got = ctx.download(y)
    assert_close(got, ref, TOL[mode], "persistent conv fwd %s" % (case,))
    # every tile individually (a misplaced tile would hide in a max over the whole tensor only if it were zero)
    gt, rt = got.reshape(-1, cout), ref.reshape(-1, cout)
    ntile = -(-gt.shape[0] // 256)
    pad = ntile * 256 - gt.shape[0]
    e = np.abs(np.pad(gt - rt, ((0, pad), (0, 0)))).reshape(ntile, -1).max(1)
    assert e.max() <= TOL[mode] * np.abs(rt).max(), "tile %d off by %.3e" % (int(e.argmax()), float(e.max()))
    if (cin % 256 == 0 and not up) or (up and extra == "dgrad"):
        dy = _prep(rs.randn(*ref.shape).astype(np.float32), mode)
        y.grad = ctx.upload(dy)
        ctx.group_wgrads, keep = True, ctx.group_wgrads
        ctx.backward()
        ctx.group_wgrads = keep
        wf = np.ascontiguousarray(wgt[::-1, ::-1].transpose(0, 1, 3, 2))            # adjoint: 180-degree rotation, in/out swapped
        dx = _torch_conv_ref(dy, wf)
        if up:                              # adjoint of the nearest upsample: 2x2 sum
            dx = dx.reshape(n, hs, 2, ws, 2, cin).sum(axis=(2, 4))
        if relu:
            dx = dx * (x > 0)
        tol = TOL[mode] * (2 if up else 1)  # the sub-pixel form rounds SUMS of up to four filter taps to 16 bits
        assert_close(ctx.download(xd.grad), dx, tol, "persistent conv dgrad %s" % (case,))


@pytest.mark.parametrize("case", [(4, 32, 32, 128, 128, True, True),      # 64 x 64 kernel both ways, accumulating into a shortcut
                                  (8, 16, 16, 64, 64, False, False),
                                  (128, 32, 32, 128, 128, True, False),    # D.Block.1.Conv2 at the bench batch: 256 x 128 kernel for dx
                                  (128, 16, 16, 128, 128, True, True),     # D.Block.2.Conv2
                                  (32, 64, 64, 64, 128, True, False)])     # 32-wide parity planes, one channel chunk (the gather form of the 256 x 128 halo kernel)
@pytest.mark.parametrize("force9", [False, True])      # True: the filter gradient on the nine-tap kernel's pooled form whatever the size (and the forward on the 256 x 128 halo kernel's parity-plane form)
def test_conv2d_meanpool(dev, case, force9, monkeypatch):
    """ConvMeanPool with the pool folded into the convolution (one 4x4 stride-2 convolution with summed filters forward, the
    sub-pixel form for the data gradient, the ordinary grouped filter gradient on the spread dy) against conv -> mean pool."""
    import torch.nn.functional as F
    from rcgan_amd import ops as O
    ctx, mode = dev
    if mode == "f32":
        pytest.skip("the folded pool runs on the 16-bit matrix-core path")
    if force9:
        monkeypatch.setenv("RCGAN_WGRAD9_MINWORK", "0")
        monkeypatch.setenv("RCGAN_WGRAD9_GROUP_MINWORK", "0")
        monkeypatch.setenv("RCGAN_H8N_GATHER_MINBLK", "1")
    n, h, w, cin, cout, relu, acc = case
    rs = np.random.RandomState(n * 7 + cin)
    x = _prep(rs.randn(n, h, w, cin).astype(np.float32), mode)
    wgt = half_round(mode, (rs.randn(3, 3, cin, cout) / np.sqrt(9 * cin)).astype(np.float32))
    b = rs.randn(cout).astype(np.float32)
    ctx.new_step()
    xd = ctx.upload(x)
    xd.req = True
    wp, bp = FakeParam(ctx, wgt), FakeParam(ctx, b)
    W = O.Weight(ctx, wp.t)
    assert O.conv_meanpool_ok(ctx, xd, W)
    full = _torch_conv_ref(x, wgt, False, relu) + b
    ref = full.reshape(n, h // 2, 2, w // 2, 2, cout).mean(axis=(2, 4))
    t0 = None
    if acc:
        t0 = _prep(rs.randn(n, h // 2, w // 2, cout).astype(np.float32), mode)
        ref = ref + t0
    y = O.conv2d_meanpool(ctx, xd, W, bp.t, in_relu=relu, accumulate_into=ctx.upload(t0) if acc else None)
    assert_close(ctx.download(y), ref, 2 * TOL[mode], "conv+meanpool fwd %s" % (case,))       # sums of up to four taps rounded once
    dy = _prep(rs.randn(*ref.shape).astype(np.float32), mode)
    y.grad = ctx.upload(dy)
    # (the small case takes the one-layer entry point, the others the grouped launch)
    ctx.group_wgrads, keep = n > 4, ctx.group_wgrads
    ctx.backward()
    ctx.group_wgrads = keep
    dyf = np.repeat(np.repeat(dy, 2, axis=1), 2, axis=2) * 0.25
    wf = np.ascontiguousarray(wgt[::-1, ::-1].transpose(0, 1, 3, 2))
    dx = _torch_conv_ref(dyf, wf)
    if relu:
        dx = dx * (x > 0)
    assert_close(ctx.download(xd.grad), dx, 2 * TOL[mode], "conv+meanpool dgrad %s" % (case,))
    xin = np.maximum(x, 0) if relu else x
    xt = torch.from_numpy(np.ascontiguousarray(xin)).permute(0, 3, 1, 2).double()
    gt = torch.from_numpy(np.ascontiguousarray(half_round(mode, dyf))).permute(0, 3, 1, 2).double()
    dw = torch.nn.grad.conv2d_weight(xt, (cout, cin, 3, 3), gt, padding=1).permute(2, 3, 1, 0).numpy()
    assert_close(wp.grad(ctx), dw, 2e-2 if mode in HALF else 2e-4, "conv+meanpool wgrad %s" % (case,))
    assert_close(bp.grad(ctx), dyf.astype(np.float64).sum((0, 1, 2)), 2e-2, "conv+meanpool bias grad %s" % (case,))


def test_batched_phase_filters_equal_single_calls(dev):
    """The summed sub-pixel filters written by the batched preparation (tile x tap-class units riding in the grid of
    conv_prepare_batch_kernel, LDS transpose) against the one-filter entry point (one element per thread): every byte of the
    prepared buffers -- both ordinary layouts and both summed layouts, upsample and mean-pool families, with and without sigma."""
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    if mode == "f32":
        pytest.skip("the sub-pixel forms run on the 16-bit matrix-core path")
    rs = np.random.RandomState(5)
    ctx.new_step()
    shapes = [(1024, 256, L.CONV_IN_UPSAMPLE2X, False), (256, 256, L.CONV_IN_UPSAMPLE2X, False), (128, 128, L.CONV_OUT_MEANPOOL2, True),
              (128, 128, L.CONV_OUT_MEANPOOL2, True), (64, 192, L.CONV_IN_UPSAMPLE2X, True), (192, 64, L.CONV_OUT_MEANPOOL2, False),
              (128, 128, 0, True)]
    ws, items = [], []
    for cin, cout, flags, sn in shapes:
        wp = FakeParam(ctx, (rs.randn(3, 3, cin, cout) / np.sqrt(9 * cin)).astype(np.float32))
        sigma = ctx.upload(np.array([1.7], np.float32)) if sn else None
        ws.append((O.Weight(ctx, wp.t, sigma), flags))
        items.append((ws[-1][0], 3, 1, 8, flags))
    O.prepare_batch(ctx, items, ctx.act_dtype)
    for (w, flags), (cin, cout, _, _) in zip(ws, shapes):
        desc = L.ConvDesc(1, 8, 8, cin, cout, 3, 3, 1, ctx.act_dtype, flags)
        nbytes = ctx.lib.rcgan_conv_prepared_bytes(C.byref(desc))
        (batched,) = w._prepared.values()
        assert batched.shape == (nbytes,)
        w._prepared.clear()
        single = w.prepared(desc)
        raw = lambda t: t.base.view(torch.uint8).reshape(-1)[t.ptr - t.base.data_ptr():][:nbytes - 256].cpu().numpy()
        a, b = raw(batched), raw(single)
        assert a.size == (18 + (32 if flags else 0)) * cin * cout * 2
        assert np.array_equal(a, b), "prepared bytes differ for %s: first at %d" % ((cin, cout, flags), int(np.flatnonzero(a != b)[0]))


@pytest.mark.parametrize("n", [3, 128])
def test_d_trunk_equals_layerwise_blocks(dev, n):
    """The fused 8x8 discriminator stage (four residual blocks, eight 3x3 convolutions in one launch each way, activations
    in LDS) against the same blocks built from eight conv2d calls: stage output, every saved activation's effect through the
    gradient of the input, and all filter / bias gradients.  Same rounding points, so only the fp32 summation order inside
    a convolution differs; both sides are also checked against the float64 oracle for the forward values."""
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    if mode == "f32":
        pytest.skip("the fused stage runs on 16-bit activations")
    rs = np.random.RandomState(n)
    x = _prep(rs.randn(n, 8, 8, 128), mode)
    ws = [(rs.randn(3, 3, 128, 128) / np.sqrt(9 * 128) * 1.2).astype(np.float32) for _ in range(8)]
    bs = [(0.1 * rs.randn(128)).astype(np.float32) for _ in range(8)]
    dy = _prep(rs.randn(n, 8, 8, 128), mode)
    res = {}
    for fused in (True, False):
        ctx.new_step()
        xd = ctx.upload(x)
        xd.req = True
        pw, pb = [FakeParam(ctx, w) for w in ws], [FakeParam(ctx, b) for b in bs]
        W = [O.Weight(ctx, p.t, ctx.upload(np.array([1.3], np.float32), L.F32)) for p in pw]
        if fused:
            y = O.d_trunk(ctx, xd, [(W[2 * k], pb[2 * k].t, W[2 * k + 1], pb[2 * k + 1].t) for k in range(4)])
        else:
            t = xd
            for k in range(4):
                h = O.conv2d(ctx, t, W[2 * k], pb[2 * k].t, 3, in_relu=True)
                t = O.conv2d(ctx, h, W[2 * k + 1], pb[2 * k + 1].t, 3, in_relu=True, residual=t)
            y = t
        out = ctx.download(y)
        y.grad = ctx.upload(dy)
        ctx.backward()
        res[fused] = dict(y=out, dx=ctx.download(xd.grad), dw=[ctx.download(w.dwbar) for w in W], db=[p.grad(ctx) for p in pb])
    a, b = res[True], res[False]
    assert_close(a["y"], b["y"], TOL[mode], "trunk output vs layerwise")
    # gradients: the two forwards differ by summation-order noise, so a pre-activation within that noise of zero gets a
    # different ReLU mask -- isolated elements differ by their whole value; compare in the norm
    from tests.gpu_util import rel_err
    tol = 6e-2 if n < 16 else 2e-2            # a few hundred pixels: one flipped mask is a visible fraction of a filter gradient
    assert rel_err(a["dx"], b["dx"]) < tol, rel_err(a["dx"], b["dx"])
    for i in range(8):
        assert rel_err(a["dw"][i], b["dw"][i]) < tol, (i, rel_err(a["dw"][i], b["dw"][i]))
        assert rel_err(a["db"][i], b["db"][i]) < tol, (i, rel_err(a["db"][i], b["db"][i]))
    if n <= 8:       # float64 oracle of the forward values (16-bit rounding between layers restated)
        t = x.astype(np.float64)
        q = lambda v: half_round(mode, v).astype(np.float64)
        for k in range(4):
            h = q(nn.conv2d_fwd(np.maximum(t, 0), q(ws[2 * k] / np.float32(1.3))) + bs[2 * k])
            t = q(t + nn.conv2d_fwd(np.maximum(h, 0), q(ws[2 * k + 1] / np.float32(1.3))) + bs[2 * k + 1])
        assert_close(a["y"], t, TOL[mode] * 3, "trunk output vs oracle")


@pytest.mark.parametrize("n,kind_a,kind_b", [(16, "HINGE_REAL", "HINGE_FAKE"), (128, "HINGE_REAL", "HINGE_FAKE"), (6, "NEG_MEAN", None)])
def test_d_trunk_pooled_boundary(dev, n, kind_a, kind_b):
    """rcgan_dtrunk_pooled: the stage's launch leaves mean_hw(relu(y)) beside y and the backward launch forms its incoming gradient
    from the features' gradient (ops.d_trunk(pool=ACT_RELU) -> act_meanhw_later -> proj_head) -- against the same stage followed by
    the head that pools by itself (ops.PooledLater).  The features differ by fp32 summation order only; the hinge / mean losses'
    logit gradients are piecewise constant, so everything behind them agrees to rounding."""
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    from tests.gpu_util import rel_err
    ctx, mode = dev
    if mode == "f32":
        pytest.skip("the fused stage runs on 16-bit activations")
    rs = np.random.RandomState(n + 40)
    d, v, ed = 128, 10, 128
    x = _prep(rs.randn(n, 8, 8, 128), mode)
    ws = [(rs.randn(3, 3, 128, 128) / np.sqrt(9 * 128) * 1.2).astype(np.float32) for _ in range(8)]
    bs = [(0.1 * rs.randn(128)).astype(np.float32) for _ in range(8)]
    w_out = (rs.randn(d, 1) * 0.3).astype(np.float32); b_out = rs.randn(1).astype(np.float32)
    table = (rs.randn(v, ed) * 0.1).astype(np.float32)
    w_e = (rs.randn(ed, d) * 0.2).astype(np.float32); b_e = (rs.randn(d) * 0.1).astype(np.float32)
    kinds = {"HINGE_REAL": L.LOSS_HINGE_REAL, "HINGE_FAKE": L.LOSS_HINGE_FAKE, "NEG_MEAN": L.LOSS_NEG_MEAN}
    rows_a = n // 2 if kind_b else n
    labs = [rs.randint(v, size=r).astype(np.int32) for r in (rows_a, n - rows_a) if r]
    res = {}
    for pooled in (True, False):
        ctx.new_step()
        xd = ctx.upload(x); xd.req = True
        pw, pb = [FakeParam(ctx, w) for w in ws], [FakeParam(ctx, b) for b in bs]
        W = [O.Weight(ctx, p.t, ctx.upload(np.array([1.3], np.float32), L.F32)) for p in pw]
        pw_out, pb_out, ptab, pw_e, pb_e = (FakeParam(ctx, a) for a in (w_out, b_out, table, w_e, b_e))
        W_out = O.Weight(ctx, pw_out.t, ctx.upload(np.array([1.1], np.float32), L.F32))
        W_e = O.Weight(ctx, pw_e.t, ctx.upload(np.array([0.9], np.float32), L.F32))
        y = O.d_trunk(ctx, xd, [(W[2 * k], pb[2 * k].t, W[2 * k + 1], pb[2 * k + 1].t) for k in range(4)],
                      pool=(L.ACT_RELU if pooled else None))
        feat = O.act_meanhw_later(ctx, y, L.ACT_RELU)
        assert isinstance(feat, O.PooledByProducer if pooled else O.PooledLater)
        if pooled:
            want = np.maximum(ctx.download(y).astype(np.float64), 0).mean(axis=(1, 2))
            assert_close(ctx.download(feat.feat), want, 2e-6, "features out of the stage's launch")
        parts = [(len(l), kinds[k], ctx.upload(l), None) for l, k in zip(labs, (kind_a, kind_b))]
        loss = ctx.persistent((1,), L.F32, fill=0.0)
        O.proj_head(ctx, feat, W_out, pb_out.t, ptab.t, W_e, pb_e.t, parts, 2.0, loss)
        ctx.backward()
        res[pooled] = dict(loss=ctx.download(loss), dx=ctx.download(xd.grad), dw=[ctx.download(w.dwbar) for w in W],
                           db=[p.grad(ctx) for p in pb], dwo=ctx.download(W_out.dwbar), dwe=ctx.download(W_e.dwbar), dt=ptab.grad(ctx))
    a, b = res[True], res[False]
    assert_close(a["loss"], b["loss"], 1e-5, "loss")
    assert rel_err(a["dx"], b["dx"]) < 1e-5, rel_err(a["dx"], b["dx"])
    for i in range(8):
        assert rel_err(a["dw"][i], b["dw"][i]) < 1e-5, (i, rel_err(a["dw"][i], b["dw"][i]))
        assert rel_err(a["db"][i], b["db"][i]) < 1e-5, (i, rel_err(a["db"][i], b["db"][i]))
    for k in ("dwo", "dwe", "dt"):
        assert rel_err(a[k], b[k]) < 1e-5, (k, rel_err(a[k], b[k]))


def test_fragment_copies_written_by_the_preparation_launch(dev):
    """(round 6) rcgan_conv_prepare_batch_frags: the fragment-major filter copies of the fused 8x8 stage (eight 3x3 128 -> 128 filters,
    both directions) and of the register-filter layer, written by the filter-preparation launch straight from the fp32 weights, against
    rcgan_conv_prepare_batch + rcgan_fragments_prepare (the launch of their own they used to be): bit for bit, the row-major copies too."""
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    if mode == "f32":
        pytest.skip("fragment-major copies exist for 16-bit filters only")
    rs = np.random.RandomState(91)
    ws = [(rs.randn(3, 3, 128, 128) / np.sqrt(9 * 128)).astype(np.float32) for _ in range(9)]
    sig = [np.array([0.7 + 0.1 * i], np.float32) for i in range(9)]
    tdesc = L.ConvDesc(1, 8, 8, 128, 128, 3, 3, 1, ctx.act_dtype, L.CONV_IN_RELU)
    rdesc = L.ConvDesc(1, 16, 16, 128, 128, 3, 3, 1, ctx.act_dtype, L.CONV_IN_RELU)
    out = {}
    for fused in (True, False):
        ctx.new_step()
        W = [O.Weight(ctx, FakeParam(ctx, w).t, ctx.upload(s_, L.F32)) for w, s_ in zip(ws, sig)]
        trunk, rf = W[:8], [(W[8], rdesc)]
        shapes = [(w, 3, 1, 8) for w in W]
        if fused:
            tf, reqs = O.fragment_requests(ctx, trunk, rf)
            rode, written = O.prepare_batch(ctx, shapes, ctx.act_dtype, frags=reqs)
            assert not rode and len(written) == 9
        else:
            assert O.prepare_batch(ctx, shapes, ctx.act_dtype) is False
            tf = O.fragments_batch(ctx, trunk, rf)
        ctx.sync()

        def raw(t):       # the buffer's bytes (the DTs are untyped "u8" blocks)
            off = t.ptr - t.base.data_ptr()
            return t.base.view(torch.uint8).reshape(-1)[off:off + t.nbytes].cpu().numpy().copy()
        out[fused] = (raw(tf), raw(W[8].rf_frag), [raw(w.prepared(tdesc)) for w in W])
    a, b = out[True], out[False]
    assert np.array_equal(a[0], b[0]), "8x8 stage fragments: %d bytes differ" % int((a[0] != b[0]).sum())
    assert np.array_equal(a[1], b[1]), "register-filter fragments: %d bytes differ" % int((a[1] != b[1]).sum())
    nrow = 2 * 9 * 128 * 128 * 2          # forward rows + rotated data-gradient rows, 16-bit (the buffer is padded behind them)
    for k in range(9):
        assert np.array_equal(a[2][k][:nrow], b[2][k][:nrow]), "row-major copy %d" % k


def test_register_filter_conv_admits_only_shapes_it_runs_well(dev):
    """rcgan_conv_rf_ok is the admission test of rcgan_conv2d_rf: 3x3 stride-1 128 -> 128 on 16x16 images, 16-bit.  8x8 images are
    refused (their data gradient ran 250x slower than the tile kernel in round 3; the fused stage serves those layers)."""
    from rcgan_amd import _lib as L
    ctx, mode = dev
    if mode == "f32":
        pytest.skip("16-bit kernel")
    ok = lambda *a: bool(ctx.lib.rcgan_conv_rf_ok(C.byref(L.ConvDesc(*a))))
    dt = ctx.act_dtype
    assert ok(128, 16, 16, 128, 128, 3, 3, 1, dt, L.CONV_IN_RELU)
    assert not ok(128, 8, 8, 128, 128, 3, 3, 1, dt, L.CONV_IN_RELU)
    assert not ok(128, 16, 16, 256, 256, 3, 3, 1, dt, 0)
    assert not ok(128, 16, 16, 128, 128, 1, 1, 1, dt, 0)
    assert not ok(128, 16, 16, 128, 128, 3, 3, 1, L.F32, 0)
    d8 = L.ConvDesc(4, 8, 8, 128, 128, 3, 3, 1, dt, 0)
    assert ctx.lib.rcgan_conv2d_rf(ctx.h, C.byref(d8), 0, None, None, None, None, None, None) == -1      # RCGAN_EINVALID_ARG


@pytest.mark.parametrize("n,hw,in_relu,second", [(3, 16, True, False), (128, 16, True, True), (5, 16, True, False), (4, 16, False, True)])
def test_register_filter_conv_equals_tile_kernels(dev, n, hw, in_relu, second):
    """rcgan_conv2d_rf (csrc/conv_rf.hip: filter slices in registers, input patch resident in LDS, K split over the wavefronts) through
    ops.conv2d against the tile-per-tap kernels: forward (+bias, input ReLU), data gradient (ReLU mask; `second`: accumulated onto an
    existing gradient) and the float64 oracle of the forward values.  Same 16-bit inputs and rounding points: fp32 summation order only."""
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    from tests.gpu_util import rel_err
    ctx, mode = dev
    if mode == "f32":
        pytest.skip("the register-filter kernel runs on 16-bit activations")
    rs = np.random.RandomState(n + hw)
    x = _prep(rs.randn(n, hw, hw, 128), mode)
    wv = (rs.randn(3, 3, 128, 128) / np.sqrt(9 * 128) * 1.2).astype(np.float32)
    bv = (0.1 * rs.randn(128)).astype(np.float32)
    dy = _prep(rs.randn(n, hw, hw, 128), mode)
    g0 = _prep(rs.randn(n, hw, hw, 128), mode)
    res = {}
    for rf in (True, False):
        ctx.new_step()
        xd = ctx.upload(x); xd.req = True
        if second:
            xd.grad = ctx.upload(g0)
        pw, pb = FakeParam(ctx, wv), FakeParam(ctx, bv)
        W = O.Weight(ctx, pw.t, ctx.upload(np.array([1.3], np.float32), L.F32))
        desc = L.ConvDesc(n, hw, hw, 128, 128, 3, 3, 1, xd.dtype, L.CONV_IN_RELU if in_relu else 0)
        W.prepared(desc)
        if rf:
            assert ctx.lib.rcgan_conv_rf_ok(C.byref(desc))
            O.fragments_batch(ctx, None, [(W, desc)])
            assert W.rf_frag is not None
        y = O.conv2d(ctx, xd, W, pb.t, 3, in_relu=in_relu)
        out = ctx.download(y)
        y.grad = ctx.upload(dy)
        ctx.backward()
        res[rf] = dict(y=out, dx=ctx.download(xd.grad), dw=ctx.download(W.dwbar), db=pb.grad(ctx))
    a, b = res[True], res[False]
    assert_close(a["y"], b["y"], TOL[mode], "register-filter forward vs tile kernel")
    assert rel_err(a["y"], b["y"]) < 2e-4, rel_err(a["y"], b["y"])
    assert rel_err(a["dx"], b["dx"]) < 2e-4, rel_err(a["dx"], b["dx"])
    assert np.array_equal(a["dw"], b["dw"]) and np.array_equal(a["db"], b["db"])          # same inputs, same filter-gradient kernel
    if n <= 8:
        q = lambda v: half_round(mode, v).astype(np.float64)
        xin = np.maximum(x.astype(np.float64), 0) if in_relu else x.astype(np.float64)
        ref = q(nn.conv2d_fwd(xin, q(wv / np.float32(1.3))) + bv)
        assert_close(a["y"], ref, TOL[mode] * 3, "register-filter forward vs oracle")


@pytest.mark.parametrize("n,segments,hw", [(10, 5, 32), (6, 1, 32), (4, 2, 16)])
def test_batch_norm_inside_the_image_end_convolution(dev, n, segments, hw):
    """rcgan_conv2d_fwd_bn (forward-only passes: ops.BnPending): conditional batch norm + ReLU applied to G.Output's staged input inside
    its launch, against the written-out batch_norm_act followed by conv2d.  The affine is the same fp32 sequence rounded to 16 bits at
    the same point and the convolution is the same kernel on the same values: bit-identical outputs."""
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    if mode == "f32":
        pytest.skip("the image-end kernels run on 16-bit activations")
    rs = np.random.RandomState(n + segments)
    c, nl = 256, 10
    x = _prep(rs.randn(n, hw, hw, c) * 1.5 + 0.3, mode)
    gamma = (1.0 + 0.2 * rs.randn(nl, c)).astype(np.float32); beta = (0.2 * rs.randn(nl, c)).astype(np.float32)
    lab = rs.randint(nl, size=n).astype(np.int32)
    wv = (rs.randn(3, 3, c, 3) * 0.05).astype(np.float32); bv = (0.1 * rs.randn(3)).astype(np.float32)
    outs = []
    for defer in (True, False):
        ctx.new_step()
        rec, ctx.recording = ctx.recording, False
        try:
            xd = ctx.upload(x)
            pg, pb2, pw, pbias = FakeParam(ctx, gamma), FakeParam(ctx, beta), FakeParam(ctx, wv), FakeParam(ctx, bv)
            W = O.Weight(ctx, pw.t, None)
            h = O.batch_norm_act(ctx, xd, pg.t, pb2.t, act=L.ACT_RELU, labels=ctx.upload(lab), n_labels=nl, segments=segments, defer_apply=defer)
            assert isinstance(h, O.BnPending) == defer
            y = O.conv2d(ctx, h, W, pbias.t, 3)
            outs.append(ctx.download(y).copy())
            if defer:                         # what materialize() writes is what the other branch convolves
                hm = ctx.download(h.materialize()).copy()
            else:
                assert np.array_equal(ctx.download(h), hm)
        finally:
            ctx.recording = rec
    assert outs[0].shape == (n, hw, hw, 3) and np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 0.1
    assert np.array_equal(outs[0], outs[1])


# (n, h, w of the convolution's OUTPUT, cin, cout, upsample-folded, residual: 0 none / 1 same grid / 2 half resolution, segments, act)
BN_PATCH_CASES = [
    (60, 32, 32, 256, 256, False, 2, 5, "relu"),     # G.Block.3.Conv2's form: plain 3x3, 240 tiles of the 256 x 256 kernel, four chunks, half-resolution residual, 5 segments
    (52, 32, 32, 128, 256, True, 0, 1, "relu"),      # G.Block.3.Conv1's: sub-pixel form over 16 x 16 low-resolution images, two chunks, one image per tile
    (100, 16, 16, 256, 256, False, 1, 5, "relu"),    # G.Block.2.Conv2's: the 256 x 128 kernel, four chunks, full-resolution residual
    (50, 32, 32, 64, 128, False, 0, 2, "none"),      # one chunk (the whole patch transformed in the prologue), no activation, Cout = 128
    (13, 64, 64, 128, 128, True, 0, 1, "relu"),      # the 256 x 128 kernel's sub-pixel form over 32-wide low-resolution images
]


@pytest.mark.parametrize("case", BN_PATCH_CASES)
def test_batch_norm_inside_the_patch_convolution(dev, case):
    """conv(act(cond_batch_norm(x))) with the norm's affine + activation applied to the halo-patch kernels' staged input
    (rcgan_conv2d_fwd_bn_residual; forward-only passes: the critic steps' generator forwards, normalization.py:47-57 +
    gan_resnet.py:304-326): bit-identical to the written-out norm followed by the plain convolution."""
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    if mode == "f32":
        pytest.skip("the halo-patch kernels run on 16-bit activations")
    n, h, w, cin, cout, up, resid, segments, actname = case
    act = L.ACT_RELU if actname == "relu" else L.ACT_NONE
    rs = np.random.RandomState(n + cin + segments)
    nl = 10
    hs, ws = (h // 2, w // 2) if up else (h, w)
    x = _prep(rs.randn(n, hs, ws, cin) * 1.5 + 0.3, mode)
    gamma = (1.0 + 0.2 * rs.randn(nl, cin)).astype(np.float32); beta = (0.2 * rs.randn(nl, cin)).astype(np.float32)
    lab = rs.randint(nl, size=n).astype(np.int32)
    wv = (rs.randn(3, 3, cin, cout) / np.sqrt(9 * cin)).astype(np.float32); bv = (0.1 * rs.randn(cout)).astype(np.float32)
    r = None
    if resid:
        r = _prep(rs.randn(n, h // 2, w // 2, cout) if resid == 2 else rs.randn(n, h, w, cout), mode)
    outs = []
    for defer in (True, False):
        ctx.new_step()
        rec, ctx.recording = ctx.recording, False
        try:
            xd = ctx.upload(x)
            pg, pb2, pw, pbias = FakeParam(ctx, gamma), FakeParam(ctx, beta), FakeParam(ctx, wv), FakeParam(ctx, bv)
            W = O.Weight(ctx, pw.t, None)
            hh = O.batch_norm_act(ctx, xd, pg.t, pb2.t, act=act, labels=ctx.upload(lab), n_labels=nl, segments=segments, defer_apply=defer)
            assert isinstance(hh, O.BnPending) == defer
            if defer:       # the fused route is the one under test
                bflags = (L.CONV_IN_UPSAMPLE2X if up else 0) | (L.CONV_RESID_UPSAMPLE2X if resid == 2 else 0)
                assert ctx.lib.rcgan_conv_bn_in_ok(C.byref(L.ConvDesc(n, h, w, cin, cout, 3, 3, 1, xd.dtype, bflags)))
            y = O.conv2d(ctx, hh, W, pbias.t, 3, in_up=up, residual=ctx.upload(r) if resid else None, residual_up=resid == 2)
            outs.append(ctx.download(y).copy())
        finally:
            ctx.recording = rec
    assert outs[0].shape == (n, h, w, cout) and np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 0.1
    assert np.array_equal(outs[0], outs[1])


def test_batch_norm_on_the_staged_input_refuses_what_no_kernel_takes(dev):
    """rcgan_conv_bn_in_ok / rcgan_conv2d_fwd_bn_residual: a convolution whose routing does not end on a halo-patch (or image-end) kernel is
    refused with RCGAN_EUNSUPPORTED_SHAPE and a message -- never run on a kernel that would ignore the norm."""
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    if mode == "f32":
        pytest.skip("16-bit kernels")
    n, h, w, cin, cout = 4, 8, 8, 64, 64                   # 64 x 64-tile kernel territory
    d = L.ConvDesc(n, h, w, cin, cout, 3, 3, 1, ctx.act_dtype, 0)
    assert not ctx.lib.rcgan_conv_bn_in_ok(C.byref(d))
    assert not ctx.lib.rcgan_conv_bn_in_ok(C.byref(L.ConvDesc(60, 32, 32, 256, 256, 3, 3, 1, ctx.act_dtype, L.CONV_IN_RELU)))      # the norm's own activation only
    assert ctx.lib.rcgan_conv_bn_in_ok(C.byref(L.ConvDesc(60, 32, 32, 256, 256, 3, 3, 1, ctx.act_dtype, 0)))
    rs = np.random.RandomState(0)
    ctx.new_step()
    x = ctx.upload(_prep(rs.randn(n, h, w, cin), mode))
    pw = FakeParam(ctx, (rs.randn(3, 3, cin, cout) * 0.05).astype(np.float32))
    W = O.Weight(ctx, pw.t, None)
    y = ctx.empty((n, h, w, cout), x.dtype)
    one = ctx.upload(np.ones((1, cin), np.float32), L.F32)
    rc = ctx.lib.rcgan_conv2d_fwd_bn_residual(ctx.h, C.byref(d), C.c_void_p(x.ptr), C.c_void_p(W.prepared(d).ptr), None, None, C.c_void_p(y.ptr), 1, None,
                                             C.c_void_p(one.ptr), C.c_void_p(one.ptr), C.c_void_p(one.ptr), C.c_void_p(one.ptr), L.ACT_RELU)
    assert rc == -2      # RCGAN_EUNSUPPORTED_SHAPE
    assert b"rcgan_conv_bn_in_ok" in ctx.lib.rcgan_last_error(ctx.h)


@pytest.mark.parametrize("m,k,n", [(128, 128, 16384), (40, 64, 1024)])
def test_wide_dense_layer_on_the_matrix_cores(dev, m, k, n, monkeypatch):
    """ops.linear routes a wide dense layer on 16-bit activations (G.Input: 128 -> 16384) through the 1x1-convolution kernels with its
    prepared 16-bit filter (RCGAN_LINEAR_MFMA): against the fp32 gather GEMM on the same inputs (which keeps the weights in fp32:
    the difference is the 16-bit rounding of W) and the float64 product of the rounded operands."""
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    from tests.gpu_util import rel_err
    ctx, mode = dev
    if mode == "f32":
        pytest.skip("fp32 activations keep the fp32 gather GEMM")
    rs = np.random.RandomState(m + n)
    x = _prep(rs.randn(m, k), mode)
    wv = (rs.randn(k, n) / np.sqrt(k)).astype(np.float32); bv = (0.1 * rs.randn(n)).astype(np.float32)
    dy = _prep(rs.randn(m, n), mode)
    res = {}
    for mfma in (True, False):
        monkeypatch.setattr(O, "LINEAR_MFMA", mfma)
        ctx.new_step()
        xd = ctx.upload(x)
        pw, pb = FakeParam(ctx, wv), FakeParam(ctx, bv)
        W = O.Weight(ctx, pw.t, None)
        y = O.linear(ctx, xd, W, pb.t)
        assert y.shape == (m, n)
        out = ctx.download(y)
        if y.grad is not None:            # (a reshaped view of the convolution's output: its gradient buffer already exists)
            ctx.upload(dy, out=y.grad)
        else:
            y.grad = ctx.upload(dy)
        ctx.backward()
        res[mfma] = dict(y=out, dw=pw.grad(ctx), db=pb.grad(ctx))
    a, b = res[True], res[False]
    q = lambda v: half_round(mode, v).astype(np.float64)
    ref = q(x.astype(np.float64) @ q(wv) + bv)
    assert_close(a["y"], ref, TOL[mode], "dense layer on the matrix cores vs float64 of the rounded operands")
    assert rel_err(a["y"], b["y"]) < 6e-3, rel_err(a["y"], b["y"])                       # W rounded to 16 bits vs W in fp32
    assert rel_err(a["dw"], b["dw"]) < 2e-3 and rel_err(a["db"], b["db"]) < 2e-3, (rel_err(a["dw"], b["dw"]), rel_err(a["db"], b["db"]))
    dw_ref = x.astype(np.float64).T @ dy.astype(np.float64)
    assert rel_err(a["dw"], dw_ref) < 2e-3


HEAD_CASES = [
    # n, rows_a, kind_a, mode_a, kind_b, mode_b          mode: "lab" one-hot labels, "wts" explicit weight matrix (with gradient)
    (16, 8, "HINGE_REAL", "lab", "HINGE_FAKE", "lab"),        # rcgan / biased critic step (gan_resnet.py:585-606)
    (16, 16, "NEG_MEAN", "lab", None, None),                   # generator step (:763-773)
    (24, 12, "HINGE_REAL", "lab", "HINGE_FAKE", "wts"),       # rcgan-u critic step: fake logits of every label x confusion rows (:654-684)
    (12, 12, "NEG_MEAN", "wts", None, None),                   # rcgan-u generator step (:751-760)
    (20, 10, "HINGE_REAL", "wts", "HINGE_FAKE", "lab"),       # unbiased: real logits of every label x C^-1 rows (:613-648)
    (1024, 512, "HINGE_REAL", "lab", "HINGE_FAKE", "lab"),    # the largest supported batch (B = 512 per GPU)
]


@pytest.mark.parametrize("case", HEAD_CASES)
def test_proj_head(dev, case):
    """The fused projection head (D.Output, label embedding + D.Embedding_y, projection logits, loss terms and every
    gradient in 2-4 launches) against a float64 autograd restatement of the same formulas."""
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    if mode != "f32":
        pytest.skip("the head is fp32 regardless of the activation dtype")
    n, rows_a, kind_a, mode_a, kind_b, mode_b = case
    d, v, ed = 128, 10, 300
    rs = np.random.RandomState(n + rows_a)
    feat = rs.rand(n, d).astype(np.float32)
    w_out = (rs.randn(d, 1) * 0.3).astype(np.float32); b_out = rs.randn(1).astype(np.float32)
    table = (rs.randn(v, ed) * 0.1).astype(np.float32)
    w_e = (rs.randn(ed, d) * 0.2).astype(np.float32); b_e = (rs.randn(d) * 0.1).astype(np.float32)
    s_out, s_e, weight = np.float32(1.3), np.float32(0.7), 3.0
    kinds = {"HINGE_REAL": L.LOSS_HINGE_REAL, "HINGE_FAKE": L.LOSS_HINGE_FAKE, "NEG_MEAN": L.LOSS_NEG_MEAN}
    ctx.new_step()
    fd = ctx.upload(feat, L.F32); fd.req = True
    pw_out, pb_out, ptab, pw_e, pb_e = (FakeParam(ctx, a) for a in (w_out, b_out, table, w_e, b_e))
    W_out = O.Weight(ctx, pw_out.t, ctx.upload(np.array([s_out]), L.F32))
    W_e = O.Weight(ctx, pw_e.t, ctx.upload(np.array([s_e]), L.F32))
    parts, host = [], []
    for rows, kind, md in ((rows_a, kind_a, mode_a), (n - rows_a, kind_b, mode_b)):
        if rows == 0:
            continue
        if md == "lab":
            lab = rs.randint(v, size=rows).astype(np.int32)
            parts.append((rows, kinds[kind], ctx.upload(lab), None)); host.append((rows, kind, lab, None, None))
        else:
            w = rs.rand(rows, v).astype(np.float32)
            wd = ctx.upload(w, L.F32); wd.req = True
            parts.append((rows, kinds[kind], None, wd)); host.append((rows, kind, None, w, wd))
    loss = ctx.persistent((1,), L.F32, fill=0.0)
    logits = ctx.empty((n, v), L.F32)
    O.proj_head(ctx, fd, W_out, pb_out.t, ptab.t, W_e, pb_e.t, parts, weight, loss, logits=logits)
    ctx.flush_wgrads()          # the head's parameter gradients are deferred (they ride in later launches of a step): here, on their own
    # float64 autograd restatement
    T = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    tf, two, tbo, tt, twe, tbe = T(feat), T(w_out), T(b_out), T(table), T(w_e), T(b_e)
    E = tt @ (twe / float(s_e)) + tbe
    psi = (tf @ (two / float(s_out))).reshape(-1) + tbo
    lg = psi[:, None] + tf @ E.t()
    total, r0, twts = 0.0, 0, []
    for rows, kind, lab, w, _ in host:
        x = lg[r0:r0 + rows]
        term = {"HINGE_REAL": torch.relu(1 - x), "HINGE_FAKE": torch.relu(1 + x), "NEG_MEAN": -x}[kind]
        if lab is not None:
            wt = torch.nn.functional.one_hot(torch.as_tensor(lab, dtype=torch.long), v).double()
        else:
            wt = T(w); twts.append(wt)
        total = total + (term * wt).sum(1).mean()
        r0 += rows
    total = weight * total
    total.backward()
    assert_close(ctx.download(loss), np.array([float(total.detach())]), 1e-5, "head loss")
    got_l, ref_l = ctx.download(logits), lg.detach().numpy()
    r0 = 0
    for rows, kind, lab, w, _ in host:          # logits are reported where they enter the loss
        m = np.ones((rows, v), bool) if lab is None else (np.arange(v)[None] == lab[:, None])
        assert_close(got_l[r0:r0 + rows][m], ref_l[r0:r0 + rows][m], 2e-5, "head logits")
        r0 += rows
    assert_close(ctx.download(fd.grad), tf.grad.numpy(), 2e-5, "head dfeat")
    assert_close(ctx.download(W_out.dwbar), two.grad.numpy() * float(s_out), 2e-5, "head d(w_out / sigma)")
    assert_close(pb_out.grad(ctx), tbo.grad.numpy(), 2e-5, "head db_out")
    assert_close(ptab.grad(ctx), tt.grad.numpy(), 2e-5, "head dtable")
    assert_close(ctx.download(W_e.dwbar), twe.grad.numpy() * float(s_e), 2e-5, "head d(W_e / sigma)")
    assert_close(pb_e.grad(ctx), tbe.grad.numpy(), 2e-5, "head db_e")
    for (_, _, _, _, wd), wt in zip([h for h in host if h[3] is not None], twts):
        assert_close(ctx.download(wd.grad), wt.grad.numpy(), 2e-5, "head dwts")


@pytest.mark.parametrize("case", [(16, 8, 64, 128, "HINGE_REAL", "HINGE_FAKE"),      # the critic step's shape: [2B, 8, 8, 128]
                                  (8, 8, 64, 128, "NEG_MEAN", None),                    # the generator step's
                                  (6, 2, 9, 256, "HINGE_REAL", "HINGE_FAKE")])          # two 128-channel halves, odd pixel count
def test_proj_head_pools_features(dev, case):
    """proj_head fed with the trunk's output (ops.act_meanhw_later): relu + spatial mean inside the head's launch and the
    gradient written straight to dx, against float64 autograd of relu -> mean -> head."""
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    n, rows_a, hw, d, kind_a, kind_b = case
    v, ed = 10, 300
    rs = np.random.RandomState(n * 3 + hw)
    x = _prep(rs.randn(n, hw, 1, d).astype(np.float32), mode)
    w_out = (rs.randn(d, 1) * 0.3).astype(np.float32); b_out = rs.randn(1).astype(np.float32)
    table = (rs.randn(v, ed) * 0.1).astype(np.float32)
    w_e = (rs.randn(ed, d) * 0.2).astype(np.float32); b_e = (rs.randn(d) * 0.1).astype(np.float32)
    s_out, s_e, weight = np.float32(1.3), np.float32(0.7), 3.0
    kinds = {"HINGE_REAL": L.LOSS_HINGE_REAL, "HINGE_FAKE": L.LOSS_HINGE_FAKE, "NEG_MEAN": L.LOSS_NEG_MEAN}
    ctx.new_step()
    xd = ctx.upload(x); xd.req = True
    pw_out, pb_out, ptab, pw_e, pb_e = (FakeParam(ctx, a) for a in (w_out, b_out, table, w_e, b_e))
    W_out = O.Weight(ctx, pw_out.t, ctx.upload(np.array([s_out]), L.F32))
    W_e = O.Weight(ctx, pw_e.t, ctx.upload(np.array([s_e]), L.F32))
    parts, host = [], []
    for rows, kind in ((rows_a, kind_a), (n - rows_a, kind_b)):
        if rows:
            lab = rs.randint(v, size=rows).astype(np.int32)
            parts.append((rows, kinds[kind], ctx.upload(lab), None)); host.append((rows, kind, lab))
    loss = ctx.persistent((1,), L.F32, fill=0.0)
    feat = O.act_meanhw_later(ctx, xd, L.ACT_RELU)
    assert isinstance(feat, O.PooledLater)
    O.proj_head(ctx, feat, W_out, pb_out.t, ptab.t, W_e, pb_e.t, parts, weight, loss)
    ctx.flush_wgrads()
    T = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    tx, two, tbo, tt, twe, tbe = T(x), T(w_out), T(b_out), T(table), T(w_e), T(b_e)
    tf = torch.relu(tx).mean(dim=(1, 2))
    E = tt @ (twe / float(s_e)) + tbe
    lg = ((tf @ (two / float(s_out))).reshape(-1) + tbo)[:, None] + tf @ E.t()
    total, r0 = 0.0, 0
    for rows, kind, lab in host:
        xx = lg[r0:r0 + rows]
        term = {"HINGE_REAL": torch.relu(1 - xx), "HINGE_FAKE": torch.relu(1 + xx), "NEG_MEAN": -xx}[kind]
        total = total + (term * torch.nn.functional.one_hot(torch.as_tensor(lab, dtype=torch.long), v).double()).sum(1).mean()
        r0 += rows
    total = weight * total
    total.backward()
    tol = 2e-5 if mode == "f32" else TOL[mode]             # dx is stored in the activation dtype
    assert_close(ctx.download(loss), np.array([float(total.detach())]), 1e-5, "pooled head loss")
    assert_close(ctx.download(xd.grad), tx.grad.numpy(), tol, "pooled head dx")
    assert_close(ctx.download(W_out.dwbar), two.grad.numpy() * float(s_out), 2e-5, "pooled head d(w_out / sigma)")
    assert_close(ptab.grad(ctx), tt.grad.numpy(), 2e-5, "pooled head dtable")
    assert_close(ctx.download(W_e.dwbar), twe.grad.numpy() * float(s_e), 2e-5, "pooled head d(W_e / sigma)")
    assert_close(pb_e.grad(ctx), tbe.grad.numpy(), 2e-5, "pooled head db_e")


def test_label_embeddings_ride_in_filter_preparation(dev):
    """rcgan_conv_prepare_batch_embed: the projection head's E = table @ W_e / sigma + b_e computed as extra workgroups of the batched
    filter preparation -- against numpy, and the head fed with it (rcgan_head_desc::E_pre) against the head computing it itself."""
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    d, v, ed, n = 128, 10, 300, 12
    rs = np.random.RandomState(11)
    feat = rs.rand(n, d).astype(np.float32)
    w_out = (rs.randn(d, 1) * 0.3).astype(np.float32); b_out = rs.randn(1).astype(np.float32)
    table = (rs.randn(v, ed) * 0.1).astype(np.float32)
    w_e = (rs.randn(ed, d) * 0.2).astype(np.float32); b_e = (rs.randn(d) * 0.1).astype(np.float32)
    lab = rs.randint(v, size=n).astype(np.int32)
    out = []
    for ride in (True, False):
        ctx.new_step()
        fd = ctx.upload(feat, L.F32); fd.req = True
        pw_out, pb_out, ptab, pw_e, pb_e = (FakeParam(ctx, a) for a in (w_out, b_out, table, w_e, b_e))
        W_out = O.Weight(ctx, pw_out.t, ctx.upload(np.array([1.3], np.float32), L.F32))
        W_e = O.Weight(ctx, pw_e.t, ctx.upload(np.array([0.7], np.float32), L.F32))
        E = None
        if ride:
            conv_w = FakeParam(ctx, (rs.randn(3, 3, 64, 64) * 0.05).astype(np.float32))
            E = ctx.empty((v, d), L.F32)
            assert O.prepare_batch(ctx, [(O.Weight(ctx, conv_w.t), 3, 1, 8)], ctx.act_dtype, embed=(ptab.t, W_e, pb_e.t, E))
            ref = table.astype(np.float64) @ (w_e.astype(np.float64) / 0.7) + b_e
            assert_close(ctx.download(E), ref, 2e-5, "riding label embeddings")
        loss = ctx.persistent((1,), L.F32, fill=0.0)
        O.proj_head(ctx, fd, W_out, pb_out.t, ptab.t, W_e, pb_e.t, [(n, L.LOSS_HINGE_REAL, ctx.upload(lab), None)], 2.0, loss, E_pre=E)
        ctx.flush_wgrads()
        out.append((ctx.download(loss).copy(), ctx.download(fd.grad).copy(), ctx.download(W_e.dwbar).copy(), ptab.grad(ctx).copy()))
    for a, b in zip(*out):
        assert np.array_equal(a, b), "head with riding embeddings differs from the head computing them itself"


def _random_cases(kind, count, seed):
    """Seeded shape sweeps for the dense / transposed-conv / batch-norm tests (sizes the fixed lists do not pin down)."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(count):
        if kind == "deconv":          # (input size, cin, cout): 5x5 stride-2 transposed conv, odd sizes and channel counts
            out.append((int(rs.randint(2, 9)), int(rs.choice([1, 3, 8, 10, 33, 64, 138])), int(rs.choice([1, 2, 4, 5, 16, 70, 128]))))
        elif kind == "linear":        # (m, k, n): tiny / skinny / tiled paths, unaligned sizes
            out.append((int(rs.randint(1, 70)), int(rs.choice([1, 7, 64, 100, 110, 513, 1030, 4100])), int(rs.choice([1, 3, 10, 17, 64, 130, 1024]))))
        else:                         # (shape, conditional): fused (power-of-two channels) and generic paths
            c = int(rs.choice([8, 24, 64, 96, 128, 200, 256, 512]))
            if rs.randint(2):
                out.append(((int(rs.randint(2, 9)), int(rs.randint(1, 7)), int(rs.randint(1, 7)), c), bool(rs.randint(2))))
            else:
                out.append(((int(rs.randint(2, 40)), c), False))
    return out


@pytest.mark.parametrize("hin,cin,cout", [(7, 10, 6), (14, 138, 1), (7, 138, 128)] + _random_cases("deconv", 8, 11))
def test_deconv(dev, hin, cin, cout):
    from rcgan_amd import ops as O
    ctx, mode = dev
    rs = np.random.RandomState(hin + cin)
    n = 3
    x = _prep(rs.randn(n, hin, hin, cin), mode)
    w = (rs.randn(5, 5, cout, cin) * 0.05).astype(np.float32)
    b = rs.randn(cout).astype(np.float32)
    ctx.new_step()
    xd = ctx.upload(x)
    xd.req = True
    wp, bp = FakeParam(ctx, w), FakeParam(ctx, b)
    oshape = (n, 2 * hin, 2 * hin, cout)
    y = O.deconv2d(ctx, xd, wp.t, bp.t, oshape)
    ref = nn.conv2d_transpose_fwd(x.astype(np.float64), w.astype(np.float64), oshape, 2) + b
    assert_close(ctx.download(y), ref, TOL[mode], "deconv fwd")
    dy = _prep(rs.randn(*oshape), mode)
    y.grad = ctx.upload(dy)
    ctx.backward()
    assert_close(ctx.download(xd.grad), nn.conv2d_transpose_bwd_input(dy.astype(np.float64), w.astype(np.float64), 2), TOL[mode], "deconv dx")
    assert_close(wp.grad(ctx), nn.conv2d_transpose_bwd_filter(x.astype(np.float64), dy.astype(np.float64), w.shape, 2), 2e-4, "deconv dw")
    assert_close(bp.grad(ctx), dy.astype(np.float64).sum(axis=(0, 1, 2)), 2e-4, "deconv db")


@pytest.mark.parametrize("n,hin,c1,c2,cout,k", [(5, 7, 128, 10, 128, 5), (3, 7, 64, 10, 64, 5), (9, 4, 128, 3, 64, 3), (2, 14, 64, 16, 128, 5), (11, 7, 192, 10, 128, 5)])
def test_deconv_filter_gradient_with_label_columns(dev, n, hin, c1, c2, cout, k):
    """(round 6) ops.deconv2d on x = conv_cond_concat(t, yb) (mnist/ops.py:46-51, model.py:722-731): rcgan_deconv2d_bwd_weight_concat --
    the gather GEMM over the c1 real channels, the c2 label columns of dW from per-sample sub-grid sums of dy -- against the oracle's
    filter gradient of the full (c1 + c2)-channel input, with DENSE label rows (not only one-hot), odd sample counts, 3x3 and 5x5
    filters; the data gradient (first c1 channels) and the bias gradient ride along.  Tolerance of the plain filter-gradient path."""
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    rs = np.random.RandomState(n * 100 + hin + c1 + c2)
    t = _prep(rs.randn(n, hin, hin, c1), mode)
    yb = rs.rand(n, c2).astype(np.float32)
    if c2 == 10:
        yb = np.eye(10, dtype=np.float32)[rs.randint(10, size=n)]          # (the trainer's case: one-hot rows)
    w = (rs.randn(k, k, cout, c1 + c2) * 0.05).astype(np.float32)
    b = rs.randn(cout).astype(np.float32)
    ctx.new_step()
    td = ctx.upload(t)
    td.req = True
    ybd = ctx.upload(yb.astype(np.float32), dtype=L.F32)
    wp, bp = FakeParam(ctx, w), FakeParam(ctx, b)
    x = O.concat_channels(ctx, td, ybd)
    assert x.concat_labels is ybd and x.concat_src[1] == c1
    oshape = (n, 2 * hin, 2 * hin, cout)
    y = O.deconv2d(ctx, x, wp.t, bp.t, oshape, k=k)
    xfull = np.concatenate([t.astype(np.float64), np.broadcast_to(yb[:, None, None, :].astype(np.float64), (n, hin, hin, c2))], axis=3)
    ref = nn.conv2d_transpose_fwd(xfull, w.astype(np.float64), oshape, 2) + b
    assert_close(ctx.download(y), ref, TOL[mode], "deconv fwd on the concatenated input")
    dy = _prep(rs.randn(*oshape), mode)
    y.grad = ctx.upload(dy)
    ctx.backward()
    dxf = nn.conv2d_transpose_bwd_input(dy.astype(np.float64), w.astype(np.float64), 2)
    assert_close(ctx.download(td.grad), dxf[..., :c1], TOL[mode], "deconv dx (real channels)")
    dwr = nn.conv2d_transpose_bwd_filter(xfull, dy.astype(np.float64), w.shape, 2)
    got = wp.grad(ctx)
    assert_close(got[..., :c1], dwr[..., :c1], 2e-4, "deconv dw, real columns")
    assert_close(got[..., c1:], dwr[..., c1:], 2e-4, "deconv dw, label columns")
    assert_close(bp.grad(ctx), dy.astype(np.float64).sum(axis=(0, 1, 2)), 2e-4, "deconv db")


@pytest.mark.parametrize("m,k,n", [(5, 110, 1024), (64, 128, 16384), (7, 128, 1), (9, 300, 128), (4, 3072, 10)] + _random_cases("linear", 12, 12))
def test_linear(dev, m, k, n):
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    rs = np.random.RandomState(m * 7 + n)
    x = _prep(rs.randn(m, k), mode)
    w = (rs.randn(k, n) / np.sqrt(k)).astype(np.float32)
    b = rs.randn(n).astype(np.float32)
    ctx.new_step()
    xd = ctx.upload(x)
    xd.req = True
    wp, bp = FakeParam(ctx, w), FakeParam(ctx, b)
    sg = ctx.upload(np.array([0.8], np.float32), L.F32)
    W = O.Weight(ctx, wp.t, sg)
    y = O.linear(ctx, xd, W, bp.t)
    we = w.astype(np.float64) / np.float32(0.8)
    ref = x.astype(np.float64) @ we + b
    assert_close(ctx.download(y), ref, TOL[mode], "linear fwd")
    dy = _prep(rs.randn(m, n), mode)
    y.grad = ctx.upload(dy)
    ctx.backward()
    assert_close(ctx.download(xd.grad), dy.astype(np.float64) @ we.T, TOL[mode], "linear dx")
    assert_close(ctx.download(W.dwbar), x.astype(np.float64).T @ dy.astype(np.float64), 2e-4, "linear dw")
    assert_close(bp.grad(ctx), dy.astype(np.float64).sum(0), 2e-4, "linear db")


@pytest.mark.parametrize("shape,cond", [((6, 4, 4, 64), True), ((5, 8, 8, 256), True), ((16, 1024), False), ((7, 14, 14, 128), False), ((4, 2, 2, 64), False),
                                        ((32, 16, 16, 256), True), ((48, 32, 32, 64), False), ((3, 5, 5, 24), True),
                                        # conditional backward finisher: 4x4 samples two to a workgroup (even n) / one each (odd n), more than
                                        # one 64-sample round with a ragged tail, split samples (groups per sample > 1)
                                        ((5, 4, 4, 128), True), ((130, 4, 4, 64), True), ((200, 8, 8, 64), True), ((3, 32, 32, 64), True),
                                        # >= 4M elements: the tree reduction (whole rows per workgroup, two-level arrival tree)
                                        ((32, 32, 32, 128), True), ((128, 8, 8, 512), True), ((64, 16, 16, 256), False),
                                        ((40, 16, 16, 1024), True), ((70, 32, 32, 64), False), ((33, 16, 16, 512), True)]
                         + _random_cases("bn", 10, 13))
def test_batch_norm(dev, shape, cond):
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    rs = np.random.RandomState(len(shape) * 100 + shape[-1])
    c = shape[-1]
    x = _prep(rs.randn(*shape) * 1.5 + 0.3, mode)
    nl = 10 if cond else 1
    gamma = (1 + 0.3 * rs.randn(nl, c)).astype(np.float32)
    beta = (0.2 * rs.randn(nl, c)).astype(np.float32)
    labels = rs.randint(nl, size=shape[0]).astype(np.int32)
    ctx.new_step()
    xd = ctx.upload(x)
    xd.req = True
    gp, bp = FakeParam(ctx, gamma), FakeParam(ctx, beta)
    act = L.ACT_RELU if cond else L.ACT_LRELU
    mm = ctx.upload(np.zeros(c, np.float32), L.F32)
    mv = ctx.upload(np.ones(c, np.float32), L.F32)
    lab = ctx.upload(labels) if cond else None
    y = O.batch_norm_act(ctx, xd, gp.t, bp.t, act=act, labels=lab, n_labels=nl, moving=None if cond else (mm, mv))
    x64 = x.astype(np.float64)
    x4 = x64 if x64.ndim == 4 else x64.reshape(shape[0], 1, 1, c)
    if cond:
        pre, st = nn.cond_batchnorm_fwd(x4, labels, gamma.astype(np.float64), beta.astype(np.float64))
        ref = np.maximum(pre, 0)
    else:
        pre, st, mm2, mv2 = nn.batch_norm_train_fwd(x4, gamma[0].astype(np.float64), beta[0].astype(np.float64), np.zeros(c), np.ones(c))
        ref = np.maximum(pre, 0.2 * pre)
        assert_close(ctx.download(mm), mm2, 1e-5, "moving mean")
        assert_close(ctx.download(mv), mv2, 1e-5, "moving var")
    yk = ctx.download(y).reshape(x4.shape)
    assert_close(yk, ref, TOL[mode], "bn fwd")
    dy = _prep(rs.randn(*shape), mode)
    y.grad = ctx.upload(dy)
    ctx.backward()
    # the activation mask is the sign of the kernel's OWN pre-activation (recomputed from x with the forward's arithmetic, or
    # read from the stored y): the reference backward takes it from the forward output the kernel produced, so an element
    # whose pre-activation is within fp32 rounding of zero does not decide the test
    yq = yk
    dpre = dy.astype(np.float64).reshape(x4.shape) * (np.where(yq > 0, 1.0, 0.0) if cond else np.where(yq > 0, 1.0, 0.2))
    if cond:
        dx, dg, db = nn.cond_batchnorm_bwd(dpre, x4, labels, gamma.astype(np.float64), st)
    else:
        dx, dg, db = nn.batch_norm_train_bwd(dpre, x4, gamma[0].astype(np.float64), st)
        dg, db = dg[None], db[None]
    assert_close(ctx.download(xd.grad).reshape(x4.shape), dx, TOL[mode] * 2, "bn dx")
    assert_close(gp.grad(ctx), dg, 2e-4, "bn dgamma")
    assert_close(bp.grad(ctx), db, 2e-4, "bn dbeta")


def test_batch_norm_segments_large(dev):
    """Segmented forward (several Generator() batches in one pass, statistics per segment) on the tree reduction path:
    3 segments of [32, 32, 32, 128] against the oracle applied to each segment on its own."""
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    rs = np.random.RandomState(77)
    K, n, c = 3, 32, 128
    x = _prep(rs.randn(K * n, 32, 32, c) * 1.3 + rs.randn(K, 1, 1, 1, 1).repeat(n, 1).reshape(K * n, 1, 1, 1), mode)
    gamma = (1 + 0.3 * rs.randn(10, c)).astype(np.float32)
    beta = (0.2 * rs.randn(10, c)).astype(np.float32)
    labels = rs.randint(10, size=K * n).astype(np.int32)
    ctx.new_step()
    rec, ctx.recording = ctx.recording, False
    try:
        y = O.batch_norm_act(ctx, ctx.upload(x), ctx.upload(gamma, L.F32), ctx.upload(beta, L.F32), act=L.ACT_RELU, labels=ctx.upload(labels),
                             n_labels=10, segments=K)
    finally:
        ctx.recording = rec
    got = ctx.download(y)
    for k in range(K):
        sl = slice(k * n, (k + 1) * n)
        pre, _ = nn.cond_batchnorm_fwd(x[sl].astype(np.float64), labels[sl], gamma.astype(np.float64), beta.astype(np.float64))
        assert_close(got[sl], np.maximum(pre, 0), TOL[mode], "segment %d" % k)


def test_batch_norm_infer(dev):
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    rs = np.random.RandomState(3)
    x = _prep(rs.randn(5, 7, 7, 128), mode)
    g, b, mm, mv = (rs.rand(128).astype(np.float32) + 0.5 for _ in range(4))
    ctx.new_step()
    up = lambda a: ctx.upload(a, L.F32)
    y = O.batch_norm_infer(ctx, ctx.upload(x), up(g), up(b), up(mm), up(mv), act=L.ACT_RELU)
    ref = np.maximum(nn.batch_norm_infer(x.astype(np.float64), g, b, mm, mv), 0)
    assert_close(ctx.download(y), ref, TOL[mode], "bn infer")


@pytest.mark.parametrize("shape", [(3, 3, 3, 128), (3, 3, 128, 128), (1, 1, 128, 128), (128, 1), (300, 128), (3072, 10), (5, 5, 64, 64)])
def test_spectral_norm(dev, shape):
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    if mode in HALF:
        pytest.skip("spectral norm is fp32 regardless of the activation dtype")
    rs = np.random.RandomState(shape[0] * 13 + shape[-1])
    w = (rs.randn(*shape) * 0.1).astype(np.float32)
    u = rs.randn(1, shape[-1]).astype(np.float32)
    ctx.new_step()
    wp = FakeParam(ctx, w)
    ud = ctx.persistent((shape[-1],), L.F32)
    ctx.view(ud).copy_(torch.from_numpy(u.reshape(-1)))
    torch.cuda.synchronize()
    (W,) = O.spectral_norm_batch(ctx, [(wp.t, ud, True)])
    wbar, sigma, u2, cache = nn.spectral_norm_fwd(w.astype(np.float64), u.astype(np.float64))
    assert_close(ctx.download(W.sigma), np.array([sigma]), 1e-5, "sigma")
    assert_close(ctx.download(ud), u2.reshape(-1), 1e-5, "u'")
    g = rs.randn(*shape).astype(np.float32)
    dwb = W.grad_target()
    ctx.upload(g, L.F32, out=dwb)
    ctx.backward()
    ref = nn.spectral_norm_bwd(g.astype(np.float64), w.astype(np.float64), u.astype(np.float64), cache)
    assert_close(wp.grad(ctx), ref, 5e-5, "sn backward")
    # NO_OPS: u untouched
    ctx.new_step()
    before = ctx.download(ud).copy()
    O.spectral_norm_batch(ctx, [(wp.t, ud, False)])
    assert np.array_equal(ctx.download(ud), before)


def test_elementwise_and_resampling(dev):
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    rs = np.random.RandomState(9)
    x = _prep(rs.randn(3, 8, 6, 10), mode)
    dy_full = _prep(rs.randn(3, 8, 6, 10), mode)
    for kind, f, df in ((L.ACT_RELU, lambda v: np.maximum(v, 0), lambda v, y: (v > 0) * 1.0),
                        (L.ACT_LRELU, lambda v: np.maximum(v, 0.2 * v), lambda v, y: np.where(v > 0, 1.0, 0.2)),
                        (L.ACT_TANH, np.tanh, lambda v, y: 1 - y * y),
                        (L.ACT_SIGMOID, lambda v: 1 / (1 + np.exp(-v)), lambda v, y: y * (1 - y))):
        ctx.new_step()
        xd = ctx.upload(x)
        xd.req = True
        y = O.act(ctx, xd, kind)
        ref = f(x.astype(np.float64))
        assert_close(ctx.download(y), ref, TOL[mode], "act %d" % kind)
        y.grad = ctx.upload(dy_full)
        ctx.backward()
        yq = half_round(mode, ref) if mode in HALF else ref
        assert_close(ctx.download(xd.grad), dy_full * df(x.astype(np.float64), yq), TOL[mode], "act bwd %d" % kind)
    ctx.new_step()
    xd = ctx.upload(x)
    xd.req = True
    y = O.meanpool2(ctx, xd)
    assert_close(ctx.download(y), nn.meanpool2(x.astype(np.float64)), TOL[mode], "meanpool")
    dy = _prep(rs.randn(3, 4, 3, 10), mode)
    y.grad = ctx.upload(dy)
    ctx.backward()
    assert_close(ctx.download(xd.grad), nn.meanpool2_bwd(dy.astype(np.float64)), TOL[mode], "meanpool bwd")
    ctx.new_step()
    xd = ctx.upload(x)
    xd.req = True
    y = O.upsample2(ctx, xd)
    assert np.array_equal(ctx.download(y), nn.upsample2(x))
    dy = _prep(rs.randn(3, 16, 12, 10), mode)
    y.grad = ctx.upload(dy)
    ctx.backward()
    assert_close(ctx.download(xd.grad), nn.upsample2_bwd(dy.astype(np.float64)), TOL[mode], "upsample bwd")
    ctx.new_step()
    a, b = ctx.upload(x), ctx.upload(dy_full)
    a.req = b.req = True
    y = O.add(ctx, a, b)
    assert_close(ctx.download(y), x.astype(np.float64) + dy_full, TOL[mode], "add")
    y.grad = ctx.upload(x)
    ctx.backward()
    assert np.array_equal(ctx.download(a.grad), x) and np.array_equal(ctx.download(b.grad), x)
    # conv_cond_concat
    ctx.new_step()
    yb = np.eye(10, dtype=np.float32)[rs.randint(10, size=3)]
    xd = ctx.upload(x)
    xd.req = True
    y = O.concat_channels(ctx, xd, ctx.upload(yb, L.F32))
    ref = np.concatenate([x, np.broadcast_to(yb[:, None, None, :], (3, 8, 6, 10))], axis=3)
    assert np.array_equal(ctx.download(y), ref)
    dy = _prep(rs.randn(3, 8, 6, 20), mode)
    y.grad = ctx.upload(dy)
    ctx.backward()
    assert np.array_equal(ctx.download(xd.grad), dy[..., :10])


def test_preprocess_cifar(dev):
    from oracle import cifar as oc
    ctx, mode = dev
    rs = np.random.RandomState(1)
    img = rs.randint(0, 256, size=(4, 3072))
    noise = rs.uniform(0, 1 / 128., size=(4, 3072)).astype(np.float32)
    ctx.new_step()
    out = ctx.empty((4, 3072))
    ctx.check(ctx.lib.rcgan_preprocess_cifar(ctx.h, 4, ctx.upload(img).ptr, ctx.upload(noise, 0).ptr, out.dtype, out.ptr))
    assert_close(ctx.download(out), oc.preprocess_real(img, noise), 4e-3 if mode in HALF else 1e-6, "preprocess")


def test_head_and_losses(dev):
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    ctx, mode = dev
    rs = np.random.RandomState(17)
    n, d, v = 6, 128, 10
    x = _prep(rs.randn(n, 8, 8, d), mode)
    ctx.new_step()
    xd = ctx.upload(x)
    xd.req = True
    feat = O.act_meanhw(ctx, xd, L.ACT_RELU)
    fref = np.maximum(x.astype(np.float64), 0).mean(axis=(1, 2))
    assert_close(ctx.download(feat), fref, 1e-5, "act_meanhw")
    table = FakeParam(ctx, rs.randn(v, 300) * 0.08)
    labels = rs.randint(v, size=n).astype(np.int32)
    lab = ctx.upload(labels)
    e = O.gather_rows(ctx, table.t, lab, n)
    assert np.array_equal(ctx.download(e), ctx.download(table.t)[labels])
    wemb = FakeParam(ctx, rs.randn(300, d) * 0.05)
    bemb = FakeParam(ctx, rs.randn(d) * 0.05)
    emb = O.linear(ctx, e, O.Weight(ctx, wemb.t), bemb.t)
    wpsi = FakeParam(ctx, rs.randn(d, 1) * 0.1)
    bpsi = FakeParam(ctx, rs.randn(1))
    psi = O.reshape(ctx, O.linear(ctx, feat, O.Weight(ctx, wpsi.t), bpsi.t), (-1,))
    logit = O.proj_logit(ctx, feat, psi, emb)
    T = ctx.download(table.t).astype(np.float64)
    We, be, Wp, bpv = (ctx.download(p.t).astype(np.float64) for p in (wemb, bemb, wpsi, bpsi))
    emb_ref = T[labels] @ We + be
    psi_ref = (fref @ Wp + bpv).reshape(-1)
    lref = psi_ref + (fref * emb_ref).sum(1)
    assert_close(ctx.download(logit), lref, 2e-5, "proj logit")
    loss = ctx.persistent((1,), L.F32, fill=0.0)
    real, fake = O.rows(ctx, logit, 0, 3), O.rows(ctx, logit, 3, 6)
    O.loss_term(ctx, L.LOSS_HINGE_REAL, real, 1.0, loss)
    O.loss_term(ctx, L.LOSS_HINGE_FAKE, fake, 0.5, loss)
    lossref = np.maximum(1 - lref[:3], 0).mean() + 0.5 * np.maximum(1 + lref[3:], 0).mean()
    assert_close(ctx.download(loss), np.array([lossref]), 1e-5, "hinge loss")
    ctx.backward()
    dl = np.concatenate([-1.0 * (1 - lref[:3] > 0) / 3.0, 0.5 * (1 + lref[3:] > 0) / 3.0])
    dfeat = dl[:, None] * emb_ref + dl[:, None] * Wp.reshape(1, -1)
    dx = np.broadcast_to(dfeat[:, None, None, :] / 64.0, x.shape) * (x > 0)
    assert_close(ctx.download(xd.grad), dx, TOL[mode], "head dx")
    demb = dl[:, None] * fref
    assert_close(wemb.grad(ctx), T[labels].T @ demb, 1e-4, "dW_emb")
    dT = np.zeros_like(T)
    np.add.at(dT, labels, demb @ We.T)
    assert_close(table.grad(ctx), dT, 1e-4, "d embedding_map")
    assert_close(wpsi.grad(ctx), fref.T @ dl[:, None], 1e-4, "dW_psi")

    # rcgan-u pieces: all-label logits, learned confusion softmax, weighted loss, perm BCE
    ctx.new_step()
    feat = FakeParam(ctx, rs.randn(n, d))
    psi = FakeParam(ctx, rs.randn(n))
    E = FakeParam(ctx, rs.randn(v, d) * 0.1)
    cl = FakeParam(ctx, rs.randn(v, v))
    E.t.grad = None          # in the model E is an activation (output of the projection Linear)
    lg = O.proj_logit_all(ctx, feat.t, psi.t, E.t)
    F_, P_, E_, CL = (ctx.download(p.t).astype(np.float64) for p in (feat, psi, E, cl))
    lgref = P_[:, None] + F_ @ E_.T
    assert_close(ctx.download(lg), lgref, 2e-5, "all-label logits")
    C = O.softmax_rows(ctx, cl.t)
    Cref = nn.softmax_rows(CL)
    yc = O.gather_rows(ctx, C, lab, n)
    loss = ctx.persistent((1,), L.F32, fill=0.0)
    O.loss_term(ctx, L.LOSS_HINGE_FAKE, lg, 1.0, loss, wts=yc)
    ycref = Cref[labels]
    term = np.maximum(1 + lgref, 0)
    assert_close(ctx.download(loss), np.array([(term * ycref).sum(1).mean()]), 1e-5, "weighted hinge")
    ctx.backward()
    dlg = (1 + lgref > 0) * ycref / n
    assert_close(feat.grad(ctx), dlg @ E_, 1e-4, "dfeat all")
    assert_close(E.grad(ctx), dlg.T @ F_, 1e-4, "dE")
    assert_close(psi.grad(ctx), dlg.sum(1), 1e-4, "dpsi all")
    dC = np.zeros_like(Cref)
    np.add.at(dC, labels, term / n)
    assert_close(cl.grad(ctx), nn.softmax_rows_bwd(dC, Cref), 1e-4, "d confusion_logits")
    ctx.new_step()
    xl = FakeParam(ctx, rs.randn(n, v) * 3)
    loss = ctx.persistent((1,), L.F32, fill=0.0)
    O.bce_onehot_term(ctx, xl.t, lab, 2.0, loss)
    X = ctx.download(xl.t).astype(np.float64)
    Z = np.eye(v)[labels]
    assert_close(ctx.download(loss), np.array([2.0 * nn.sigmoid_ce_logits(X, Z).mean()]), 1e-5, "bce")
    assert_close(ctx.download(xl.t.grad), 2.0 * nn.sigmoid_ce_logits_bwd(X, Z) / X.size, 1e-5, "bce grad")
    for kind, z in ((L.LOSS_CE_ONES, 1.0), (L.LOSS_CE_ZEROS, 0.0), (L.LOSS_NEG_MEAN, None)):
        ctx.new_step()
        xl = FakeParam(ctx, rs.randn(n) * 3)
        xl.t.grad = None
        loss = ctx.persistent((1,), L.F32, fill=0.0)
        O.loss_term(ctx, kind, xl.t, 1.0, loss)
        X = ctx.download(xl.t).astype(np.float64)
        if z is None:
            lr_, gr_ = -X.mean(), -np.ones_like(X) / n
        else:
            lr_, gr_ = nn.sigmoid_ce_logits(X, z).mean(), nn.sigmoid_ce_logits_bwd(X, z) / n
        assert_close(ctx.download(loss), np.array([lr_]), 1e-5, "loss kind %d" % kind)
        assert_close(ctx.download(xl.t.grad), gr_, 1e-5, "loss grad kind %d" % kind)


def test_adam_tf(dev):
    from rcgan_amd import _lib as L
    from rcgan_amd.runtime import ParamGroup
    ctx, mode = dev
    if mode in HALF:
        pytest.skip("optimiser state is fp32 regardless of the activation dtype")
    rs = np.random.RandomState(2)
    w0 = rs.randn(1000).astype(np.float32)
    pg = ParamGroup(ctx, [("w", (1000,), w0)])
    w, m, v = w0.copy(), np.zeros(1000, np.float32), np.zeros(1000, np.float32)
    for t in range(1, 4):
        g = rs.randn(1000).astype(np.float32)
        pg.set("w", g, "grad")
        pg.set_hyper(2e-4, t)
        pg.adam(0.5, 0.999, clip=1.0, grad_scale=0.5)
        w, m, v = nn.adam_tf(w, g * np.float32(0.5), m, v, t, 2e-4, 0.5, 0.999, clip=1.0)
    # one more step through the entry point that reads {lr, t} from device memory (a captured launch replayed with new values)
    from rcgan_amd import _lib as L
    g = rs.randn(1000).astype(np.float32)
    pg.set("w", g, "grad")
    hyper = ctx.upload(np.array([2e-4, 4.0], np.float32), L.F32)
    ctx.check(ctx.lib.rcgan_adam_tf(ctx.h, 1000, pg.value.data_ptr(), pg.grad.data_ptr(), pg.m.data_ptr(), pg.v.data_ptr(), hyper.ptr,
                                    0.5, 0.999, 1e-8, 1.0, 0.5))
    w, m, v = nn.adam_tf(w, g * np.float32(0.5), m, v, 4, 2e-4, 0.5, 0.999, clip=1.0)
    ctx.sync()
    assert_close(pg.get("w"), w, 2e-6, "adam w")
    assert_close(pg.get("w", "m"), m, 2e-6, "adam m")
    assert_close(pg.get("w", "v"), v, 2e-6, "adam v")


def test_rng_and_graph_replay(dev):
    import ctypes as C
    from rcgan_amd import _lib as L
    ctx, mode = dev
    if mode in HALF:
        pytest.skip("one dtype is enough")
    n = 1 << 16
    buf = ctx.persistent((n,), L.F32)
    state = torch.zeros(2, dtype=torch.int64, device=ctx.device)
    torch.cuda.synchronize()
    draw = lambda kind, lo, hi: ctx.check(ctx.lib.rcgan_rng_fill(ctx.h, n, L.F32, kind, lo, hi, 1234, C.c_void_p(state.data_ptr()), C.c_void_p(buf.ptr)))
    draw(0, 0.0, 1.0 / 128)
    u = ctx.download(buf).copy()
    assert 0 <= u.min() and u.max() < 1.0 / 128 and abs(u.mean() * 256 - 1) < 0.02
    draw(1, 0.0, 1.0)
    z = ctx.download(buf).copy()
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    ctx.sync()
    ctx.graph_begin()
    draw(1, 0.0, 1.0)
    gid = ctx.graph_end()
    ctx.graph_launch(gid)
    a = ctx.download(buf).copy()
    ctx.graph_launch(gid)
    b = ctx.download(buf).copy()
    assert not np.array_equal(a, b) and not np.array_equal(a, z)


def test_grouped_filter_gradients_equal_single_calls():
    """rcgan_conv2d_bwd_weight_group == one rcgan_conv2d_bwd_weight per layer: the grouped launch runs every layer's grid
    inside one kernel (plus one grouped slab reduction), with the pixels per workgroup chosen for the group as a whole, so
    the fp32 partial sums are cut differently (1e-6 relative) -- also for the 3 -> 128 image-end layers, whose workgroups ride
    in the grouped launch and whose slabs go through the grouped reduction; the 1x1 layer keeps its own pixel chunking and the
    256 -> 3 layers their own launches: bit-identical."""
    import ctypes as C
    from rcgan_amd import _lib as L
    ctx = make_ctx("bf16")
    try:
        lib, h = ctx.lib, ctx.h
        ctx.new_step()
        shapes = [(128, 8, 8, 128, 128, 3, L.CONV_IN_RELU), (16, 16, 16, 128, 128, 3, L.CONV_IN_RELU), (6, 8, 8, 256, 128, 3, 0),
                  (4, 32, 32, 3, 128, 3, 0), (8, 8, 8, 128, 128, 1, 0), (3, 32, 32, 128, 256, 3, L.CONV_IN_RELU),
                  (5, 32, 32, 3, 128, 3, 0), (16, 16, 16, 3, 128, 1, 0), (7, 16, 16, 3, 128, 1, 0),      # image-end layers: ride in the group
                  (2, 32, 32, 256, 3, 3, L.CONV_IN_RELU), (3, 32, 32, 256, 3, 3, 0)]                       # 256 -> 3: own launches
        items = []
        for i, (n, hh, ww, cin, cout, k, fl) in enumerate(shapes):
            x, dy = ctx.empty((n, hh, ww, cin)), ctx.empty((n, hh, ww, cout))
            ctx.check(lib.rcgan_rng_fill(h, x.size, x.dtype, 1, 0.0, 1.0, 100 + i, None, C.c_void_p(x.ptr)))
            ctx.check(lib.rcgan_rng_fill(h, dy.size, dy.dtype, 1, 0.0, 1.0, 200 + i, None, C.c_void_p(dy.ptr)))
            dws = [ctx.zeros((k, k, cin, cout), L.F32) for _ in range(2)]
            dbs = [ctx.zeros((cout,), L.F32) for _ in range(2)] if i % 2 == 0 else [None, None]
            items.append((L.ConvDesc(n, hh, ww, cin, cout, k, k, 1, L.BF16, fl), x, dy, dws, dbs))
        ws, wsb = C.c_void_p(ctx.ws_ptr), ctx.ws_bytes
        for d, x, dy, dws, dbs in items:
            ctx.check(lib.rcgan_conv2d_bwd_weight(h, C.byref(d), C.c_void_p(x.ptr), C.c_void_p(dy.ptr), C.c_void_p(dws[0].ptr),
                                                  C.c_void_p(dbs[0].ptr) if dbs[0] else None, 1, ws, wsb))
        n = len(items)
        descs = (L.ConvDesc * n)(*[it[0] for it in items])
        arr = lambda f: (C.c_void_p * n)(*[f(it) for it in items])
        ctx.check(lib.rcgan_conv2d_bwd_weight_group(h, n, descs, arr(lambda it: it[1].ptr), arr(lambda it: it[2].ptr),
                                                    arr(lambda it: it[3][1].ptr), arr(lambda it: it[4][1].ptr if it[4][1] else None),
                                                    1, ws, wsb))
        for i, (d, x, dy, dws, dbs) in enumerate(items):
            a, b = ctx.download(dws[0]), ctx.download(dws[1])
            assert np.abs(a).max() > 0
            own = (d.kh != 3 and d.cin % 128 == 0) or d.cout == 3
            if own:
                assert np.array_equal(a, b), "layer %d filter gradient" % i
            else:
                assert_close(b, a, 2e-6, "layer %d filter gradient" % i)
            if dbs[0] is not None:
                assert_close(ctx.download(dbs[1]), ctx.download(dbs[0]), 2e-6, "layer %d bias gradient" % i)
    finally:
        ctx.close()


@pytest.mark.parametrize("case", [
    # n, low-resolution h = w (the convolution's stored input), upsampled?, residual on the half-resolution grid?, segments
    (64, 32, False, False, 1),      # G.Block.3.Conv2-like without shortcut: 256 tiles
    (128, 16, True, False, 1),      # G.Block.3.Conv1 (sub-pixel form: phase-major tiles), generator step
    (128, 32, False, True, 1),      # G.Block.3.Conv2 + upsampled shortcut
    (160, 16, True, False, 5),      # five critic-step segments (phase form)
    (160, 32, False, True, 5),
])
def test_conv_tile_statistics_equal_a_statistics_pass(case, monkeypatch):
    """rcgan_conv2d_fwd_stats + rcgan_bn_stats_from_tiles (batch statistics out of the 256 x 256 kernel's epilogue) against the plain
    convolution followed by rcgan_bn_stats on its output: the stored tensor is bit-identical, mean / rstd agree to fp32 summation
    order, per segment; and batch_norm_act picks the tile statistics up (same normalised tensor to one 16-bit rounding)."""
    import rcgan_amd  # noqa: F401
    from rcgan_amd import _lib as L
    from rcgan_amd import ops as O
    n, hs, up, resid, nseg = case
    monkeypatch.setattr(O, "FUSE_BN_STATS", True)          # (opt-in in production: RCGAN_FUSE_BN_STATS=1, see ops.py)
    ctx = make_ctx("bf16", arena=3 << 30)
    try:
        rs = np.random.RandomState(n + hs)
        cin = cout = 256
        h = hs * 2 if up else hs
        x = ctx.upload(rs.randn(n, hs, hs, cin).astype(np.float32))
        w = FakeParam(ctx, (rs.randn(3, 3, cin, cout) * 0.03).astype(np.float32))
        b = FakeParam(ctx, (rs.randn(cout) * 0.5).astype(np.float32))
        r = ctx.upload(rs.randn(n, h // 2, h // 2, cout).astype(np.float32)) if resid else None
        ctx.recording = False
        W = O.Weight(ctx, w.t, None)
        desc = L.ConvDesc(n, h, h, cin, cout, 3, 3, 1, x.dtype, (L.CONV_IN_UPSAMPLE2X if up else 0) | (L.CONV_RESID_UPSAMPLE2X if resid else 0))
        assert ctx.lib.rcgan_conv_stats_ok(C.byref(desc)) == 1
        y0 = O.conv2d(ctx, x, W, b.t, 3, in_up=up, residual=r, residual_up=resid, want_stats=False)
        y1 = O.conv2d(ctx, x, W, b.t, 3, in_up=up, residual=r, residual_up=resid, want_stats=True)
        assert y0.tile_stats is None and y1.tile_stats is not None
        a0, a1 = ctx.download(y0), ctx.download(y1)
        assert np.array_equal(a0, a1)
        mean = ctx.empty((nseg, cout), L.F32)
        rstd = ctx.empty((nseg, cout), L.F32)
        ctx.check(ctx.lib.rcgan_bn_stats_from_tiles(ctx.h, C.byref(y1.tile_stats[0]), nseg, 1e-5, C.c_void_p(y1.tile_stats[1].ptr),
                                                    C.c_void_p(mean.ptr), C.c_void_p(rstd.ptr)))
        m1, r1 = ctx.download(mean), ctx.download(rstd)
        seg = a0.reshape(nseg, -1, cout).astype(np.float64)
        m_ref = seg.mean(axis=1)
        r_ref = 1.0 / np.sqrt(seg.var(axis=1) + 1e-5)
        assert np.abs(m1 - m_ref).max() <= 2e-6 * max(1.0, np.abs(m_ref).max()), np.abs(m1 - m_ref).max()
        assert np.abs(r1 / r_ref - 1).max() <= 2e-6, np.abs(r1 / r_ref - 1).max()
        # the batch norm behind it: statistics from the tiles vs from its own pass over the tensor
        lab = ctx.upload(rs.randint(10, size=n))
        gam = FakeParam(ctx, (1 + 0.1 * rs.randn(10, cout)).astype(np.float32))
        bet = FakeParam(ctx, (0.1 * rs.randn(10, cout)).astype(np.float32))
        z0 = ctx.download(O.batch_norm_act(ctx, y0, gam.t, bet.t, act=L.ACT_RELU, labels=lab, n_labels=10, segments=nseg))
        z1 = ctx.download(O.batch_norm_act(ctx, y1, gam.t, bet.t, act=L.ACT_RELU, labels=lab, n_labels=10, segments=nseg))
        assert np.abs(z1 - z0).max() <= 2.0 ** -7 * max(1.0, np.abs(z0).max()) and np.mean(z1 != z0) < 1e-3, (np.abs(z1 - z0).max(), np.mean(z1 != z0))
    finally:
        ctx.recording = True
        ctx.close()
