"""Cross-check of the numpy oracle against an independent PyTorch-CPU autograd implementation.

The reference's floating-point path is unpinned (no TF, no reference tests); this is the strongest
check available: two implementations written separately from the same reference lines must agree.
Run in float64 so disagreement means a semantic difference, not round-off.
"""
import numpy as np
import pytest
import torch

from oracle import nn, cifar
from oracle import torch_port as TR


def _rs(seed=0):
    return np.random.RandomState(seed)


@pytest.mark.parametrize("h,k,s", [(28, 5, 2), (14, 5, 2), (7, 5, 2), (4, 5, 2), (32, 3, 1), (8, 1, 1), (5, 3, 2)])
def test_conv_fwd_bwd(h, k, s):
    rs = _rs(1)
    x = rs.randn(2, h, h, 3)
    w = rs.randn(k, k, 3, 4)
    y = nn.conv2d_fwd(x, w, s)
    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(w, requires_grad=True)
    yt = TR.conv2d_same(xt, wt, s)
    np.testing.assert_allclose(y, yt.detach().numpy(), atol=1e-10)
    dy = rs.randn(*y.shape)
    yt.backward(torch.tensor(dy))
    np.testing.assert_allclose(nn.conv2d_bwd_input(dy, w, x.shape, s), xt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(nn.conv2d_bwd_filter(x, dy, w.shape, s), wt.grad.numpy(), atol=1e-10)


def test_same_pad_table():
    # SAME pads (before, after) for the MNIST 5x5 s2 stack
    assert nn.same_pad(28, 5, 2) == (14, 1, 2)
    assert nn.same_pad(14, 5, 2) == (7, 1, 2)
    assert nn.same_pad(7, 5, 2) == (4, 2, 2)
    assert nn.same_pad(4, 5, 2) == (2, 1, 2)


@pytest.mark.parametrize("hin,hout", [(7, 14), (14, 28)])
def test_conv_transpose(hin, hout):
    rs = _rs(2)
    x = rs.randn(2, hin, hin, 5)
    w = rs.randn(5, 5, 3, 5)        # [kh,kw,Cout,Cin]
    out_shape = (2, hout, hout, 3)
    y = nn.conv2d_transpose_fwd(x, w, out_shape, 2)
    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(w, requires_grad=True)
    yt = TR.conv2d_transpose_same(xt, wt, out_shape, 2)
    np.testing.assert_allclose(y, yt.detach().numpy(), atol=1e-10)
    # independent check against torch's own transposed conv: padding (1,1) + output_padding 1 for k5 s2
    yt2 = torch.nn.functional.conv_transpose2d(xt.permute(0, 3, 1, 2), wt.permute(3, 2, 0, 1), stride=2,
                                               padding=1, output_padding=0)
    # TF SAME for 14->7 has pad_before=1: full transposed output has size (hin-1)*2+5 = 2hin+3; crop [1:1+hout]
    full = torch.nn.functional.conv_transpose2d(xt.permute(0, 3, 1, 2), wt.permute(3, 2, 0, 1), stride=2)
    np.testing.assert_allclose(y, full[:, :, 1:1 + hout, 1:1 + hout].permute(0, 2, 3, 1).detach().numpy(), atol=1e-10)
    dy = rs.randn(*y.shape)
    yt.backward(torch.tensor(dy))
    np.testing.assert_allclose(nn.conv2d_transpose_bwd_input(dy, w, 2), xt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(nn.conv2d_transpose_bwd_filter(x, dy, w.shape, 2), wt.grad.numpy(), atol=1e-10)


def test_spectral_norm_grad():
    rs = _rs(3)
    w = rs.randn(3, 3, 4, 6)
    u = rs.randn(1, 6)
    wbar, sigma, u2, cache = nn.spectral_norm_fwd(w, u)
    wt = torch.tensor(w, requires_grad=True)
    wb_t, u2_t = TR.spectral_norm(wt, torch.tensor(u))
    np.testing.assert_allclose(wbar, wb_t.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(u2, u2_t.numpy(), atol=1e-12)
    g = rs.randn(*w.shape)
    wb_t.backward(torch.tensor(g))
    np.testing.assert_allclose(nn.spectral_norm_bwd(g, w, u, cache), wt.grad.numpy(), atol=1e-10)
    # sigma approximates the top singular value after many iterations
    uu = u
    for _ in range(200):
        _, s, uu, _ = nn.spectral_norm_fwd(w, uu)
    assert abs(s - np.linalg.svd(w.reshape(-1, 6), compute_uv=False)[0]) < 1e-8


def test_cond_batchnorm():
    rs = _rs(4)
    x = rs.randn(6, 4, 4, 5)
    lab = np.array([0, 3, 3, 9, 1, 0])
    sc = rs.randn(10, 5)
    of = rs.randn(10, 5)
    y, st = nn.cond_batchnorm_fwd(x, lab, sc, of)
    xt, sct, oft = (torch.tensor(a, requires_grad=True) for a in (x, sc, of))
    yt = TR.cond_bn(xt, torch.tensor(lab), sct, oft)
    np.testing.assert_allclose(y, yt.detach().numpy(), atol=1e-10)
    dy = rs.randn(*y.shape)
    yt.backward(torch.tensor(dy))
    dx, ds, do = nn.cond_batchnorm_bwd(dy, x, lab, sc, st)
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(ds, sct.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(do, oft.grad.numpy(), atol=1e-10)


def test_batch_norm_train_matches_torch():
    rs = _rs(5)
    x = rs.randn(8, 3, 3, 4)
    g, b = rs.randn(4), rs.randn(4)
    mm, mv = np.zeros(4), np.ones(4)
    y, st, mm2, mv2 = nn.batch_norm_train_fwd(x, g, b, mm, mv, 0.9, 1e-5)
    bn = torch.nn.BatchNorm2d(4, eps=1e-5, momentum=0.1).double()
    bn.weight.data[:] = torch.tensor(g)
    bn.bias.data[:] = torch.tensor(b)
    xt = torch.tensor(x, requires_grad=True)
    yt = bn(xt.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    np.testing.assert_allclose(y, yt.detach().numpy(), atol=1e-10)
    # torch also updates running_var with the unbiased variance: same rule as TF's fused batch norm
    np.testing.assert_allclose(mm2, bn.running_mean.numpy(), atol=1e-12)
    np.testing.assert_allclose(mv2, bn.running_var.numpy(), atol=1e-12)
    dy = rs.randn(*y.shape)
    yt.backward(torch.tensor(dy))
    dx, dg, db = nn.batch_norm_train_bwd(dy, x, g, st)
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(dg, bn.weight.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(db, bn.bias.grad.numpy(), atol=1e-10)


def test_adam_tf_form():
    rs = _rs(6)
    w, g = rs.randn(7), rs.randn(7)
    m, v = np.zeros(7), np.zeros(7)
    wt, mt, vt = torch.tensor(w), torch.tensor(m), torch.tensor(v)
    for t in range(1, 4):
        w, m, v = nn.adam_tf(w, g, m, v, t, 2e-4, 0.0, 0.9)
        wt, mt, vt = TR.adam_tf_torch(wt, torch.tensor(g), mt, vt, t, 2e-4, 0.0, 0.9)
    np.testing.assert_allclose(w, wt.numpy(), atol=1e-14)
    # eps sits OUTSIDE the bias correction: differs from torch.optim.Adam after step 1 for tiny grads
    w1, _, _ = nn.adam_tf(np.array([1.0]), np.array([1e-9]), np.zeros(1), np.zeros(1), 1, 1e-3, 0.5, 0.999)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.5)
    assert abs(w1[0] - (1.0 - lr_t * 0.5e-9 / (np.sqrt(0.001 * 1e-18) + 1e-8))) < 1e-15


def test_sigmoid_ce():
    rs = _rs(7)
    x, z = rs.randn(5, 10) * 4, (rs.rand(5, 10) > 0.5).astype(float)
    ref = torch.nn.functional.binary_cross_entropy_with_logits(torch.tensor(x), torch.tensor(z), reduction="none")
    np.testing.assert_allclose(nn.sigmoid_ce_logits(x, z), ref.numpy(), atol=1e-12)


def _batch(rs, B, alpha=0.6):
    C = cifar.c_alpha(alpha)
    Cinv = np.linalg.inv(C)
    lab = rs.randint(10, size=B)
    d = dict(real=cifar.preprocess_real(rs.randint(0, 256, size=(B, 3072)), rs.uniform(0, 1 / 128., size=(B, 3072))),
             labels=lab, labels_random=rs.randint(10, size=B), labels_biased=rs.randint(10, size=B),
             inv_weights=Cinv[lab], z=rs.randn(B, 128))
    g = dict(labels_random_G=rs.randint(10, size=2 * B), labels_biased_G=rs.randint(10, size=2 * B),
             z=rs.randn(2 * B, 128))
    return C, d, g


@pytest.mark.parametrize("alg,perm", [("rcgan", False), ("rcgan-u", True), ("biased", False), ("unbiased", False)])
def test_cifar_step_grads(alg, perm):
    rs = _rs(11)
    B = 2
    P, U = cifar.init_params(0, alg, perm_classifier=perm, confuse_init=True)
    # de-trivialise zero-initialised tensors so their gradients are exercised
    for k in P:
        if k.endswith("/Biases") or k.endswith("/b") or "CondBatchNorm" in k:
            P[k] = P[k] + 0.1 * rs.randn(*P[k].shape).astype("float32")
    C, db, gb = _batch(rs, B)
    cfg = dict(algorithm=alg, C=C, perm_classifier=perm, perm_multiplier=1.0)

    U1 = {k: v.copy() for k, v in U.items()}
    cost, grads = cifar.d_grads(P, U1, cfg, db, dtype=np.float64)
    tm = TR.CifarTorch(P, U)
    c = tm.disc_cost(cfg, db)
    c.backward()
    assert abs(cost - c.item()) < 1e-9
    for k, p in tm.P.items():
        if k.startswith("Discriminator"):
            assert k in grads, k
            np.testing.assert_allclose(grads[k], p.grad.numpy(), atol=1e-8, rtol=1e-7, err_msg=k)
    for k, v in tm.U_new.items():
        np.testing.assert_allclose(U1[k], v.numpy(), atol=1e-12, err_msg=k)

    U2 = {k: v.copy() for k, v in U.items()}
    cost, grads = cifar.g_grads(P, U2, cfg, gb, dtype=np.float64)
    tm = TR.CifarTorch(P, U)
    c = tm.gen_cost(cfg, gb)
    c.backward()
    assert abs(cost - c.item()) < 1e-9
    for k, p in tm.P.items():
        if k.startswith("Generator") or k == "confusion_logits":
            np.testing.assert_allclose(grads[k], p.grad.numpy(), atol=1e-8, rtol=1e-7, err_msg=k)
    # G step: conv u's untouched (NO_OPS), projection u updated
    assert np.array_equal(U2["Discriminator/D.Block.3.Conv1/filters/spectral_norm/u"],
                          U["Discriminator/D.Block.3.Conv1/filters/spectral_norm/u"])
    assert not np.array_equal(U2["Discriminator/D.Embedding_y/spectral_norm/u"],
                              U["Discriminator/D.Embedding_y/spectral_norm/u"])


def test_cifar_param_counts():
    # G 7 875 587 params; D 1 685 689; perm classifier +30 730
    P, U = cifar.init_params(0, "rcgan", perm_classifier=True)
    g = sum(v.size for k, v in P.items() if k.startswith("Generator"))
    d = sum(v.size for k, v in P.items() if k.startswith("Discriminator") and "perm" not in k)
    pc = sum(v.size for k, v in P.items() if "perm" in k)
    assert (g, d, pc) == (7875587, 1685689, 30730)


def test_towers_equal_mean_of_shards():
    rs = _rs(12)
    P, U = cifar.init_params(0, "rcgan")
    C, db, gb = _batch(rs, 4)
    cfg = dict(algorithm="rcgan", C=C)
    c2, g2 = cifar.d_grads(P, dict(U), cfg, db, ntowers=2, dtype=np.float64)
    parts = []
    for i in range(2):
        sh = {k: np.split(np.asarray(v), 2)[i] for k, v in db.items()}
        parts.append(cifar.d_grads(P, dict(U), cfg, sh, ntowers=1, dtype=np.float64))
    assert abs(c2 - 0.5 * (parts[0][0] + parts[1][0])) < 1e-12
    for k in g2:
        np.testing.assert_allclose(g2[k], 0.5 * (parts[0][1][k] + parts[1][1][k]), atol=1e-12)


# ----------------------------------------------------------------------------------------------------
# MNIST
# ----------------------------------------------------------------------------------------------------
def _mnist_batch(rs, B):
    from oracle import labels as LB
    C = LB.one_coin(0.3)
    eye = np.eye(10)
    yr = rs.randint(10, size=B)
    return C, dict(images=rs.rand(B, 28, 28, 1), z=rs.uniform(-1, 1, size=(B, 100)), y_real=eye[yr], y_gen=eye[rs.randint(10, size=B)],
                   y_fake=eye[rs.randint(10, size=B)], y_real_weights=np.linalg.inv(C)[yr])


@pytest.mark.parametrize("alg,disc,est,loss", [("rcgan", "projection", False, "hinge"), ("rcgan", "projection", True, "hinge"),
                                               ("unbiased", "projection", False, "hinge"), ("biased", "vanilla", False, "ce"),
                                               ("rcgan", "projection+y", False, "hinge")])
def test_mnist_step_grads(alg, disc, est, loss):
    from oracle import mnist as om
    rs = _rs(31)
    B = 3
    concat = disc.endswith("+y")
    disc = disc.split("+")[0]
    layers = (1, 3) if concat else ()
    P, S, U = om.init_params(0, disc, est, True, disc == "projection", layers)
    for k in P:
        if k.endswith("/bias") or k.endswith("/biases") or k.endswith("/beta") or k.endswith("/gamma"):
            P[k] = P[k] + 0.1 * rs.randn(*P[k].shape).astype("float32")
    C, b = _mnist_batch(rs, B)
    cfg = dict(algorithm=alg, disc_type=disc, estimate_confuse=est, loss_fn=loss, perm_regularizer=True, perm_multiplier=10.0,
               spectral_norm=disc == "projection", C=C, concat_y=concat, concat_y_layers=layers)
    tm = TR.MnistTorch(P, U, cfg)
    L = tm.losses(b)
    Ld, gd = om.d_grads(P, {k: v.copy() for k, v in S.items()}, dict(U), cfg, b, dtype=np.float64)
    (L["d_loss_real"] + L["d_loss_fake"] + L["class_loss_real"]).backward(retain_graph=True)
    for k in ("d_loss_real", "d_loss_fake", "g_loss", "class_loss_real", "class_loss_fake"):
        assert abs(Ld[k] - L[k].item()) < 1e-9, k
    for k, p in tm.P.items():
        if om.is_d_var(k):
            np.testing.assert_allclose(gd[k], p.grad.numpy(), atol=1e-8, rtol=1e-6, err_msg=k)
    for p in tm.P.values():
        p.grad = None
    Lg, gg = om.g_grads(P, {k: v.copy() for k, v in S.items()}, dict(U), cfg, b, dtype=np.float64)
    (L["g_loss"] + 10.0 * L["class_loss_fake"]).backward(retain_graph=True)
    for k, p in tm.P.items():
        if om.is_g_var(k):
            np.testing.assert_allclose(gg[k], p.grad.numpy(), atol=1e-8, rtol=1e-6, err_msg=k)
    if est:
        for p in tm.P.values():
            p.grad = None
        L["g_loss"].backward()
        # c_optim minimises g_loss alone; the perm term does not depend on confusion_logits
        np.testing.assert_allclose(gg["confusion_logits"], tm.P["confusion_logits"].grad.numpy(), atol=1e-9)


def test_mnist_param_shapes_and_moving_stats():
    from oracle import mnist as om
    P, S, U = om.init_params(0, "projection", True, True, True)
    assert P["generator/g_h2/w"].shape == (5, 5, 128, 138) and P["generator/g_h3/w"].shape == (5, 5, 1, 138)
    assert P["discriminator/d_h0_conv/w"].shape == (5, 5, 1, 64) and U["discriminator/d_h3_conv/spectral_norm/u"].shape == (1, 64)
    assert P["classifier/d_classifier_h1/Matrix"].shape == (784, 10) and P["confusion_logits"].shape == (10, 10)
    rs = _rs(5)
    C, b = _mnist_batch(rs, 4)
    cfg = dict(algorithm="rcgan", C=C)
    S0 = {k: v.copy() for k, v in S.items()}
    om.d_grads(P, S, dict(U), cfg, b)
    assert not np.array_equal(S["generator/g_bn0/moving_mean"], S0["generator/g_bn0/moving_mean"])
    out = om.sampler(P, S, b["z"], b["y_gen"])
    assert out.shape == (4, 28, 28, 1) and (out > 0).all() and (out < 1).all()
