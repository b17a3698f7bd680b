"""Step-level parity of the MNIST engine (HIP, through the C ABI) against the numpy oracle evaluated in
float64: one D run and the two G runs of a reference iteration (mnist/model.py:347-372) from identical weights
and batch.  fp32 activations (BASELINE cfg1/cfg2); tolerance 1e-2 of each gradient's scale with the same
near-zero floor as the CIFAR test (batch norm over <= 8 samples)."""
import numpy as np
import pytest

from oracle import labels as LB
from oracle import mnist as om
from tests.gpu_util import assert_close, rel_err

pytestmark = pytest.mark.gpu


def _batch(rs, B, alpha=0.3):
    C = LB.one_coin(alpha)
    eye = np.eye(10, dtype=np.float32)
    yr = rs.randint(10, size=B)
    return C, dict(images=rs.rand(B, 28, 28, 1).astype(np.float32), z=rs.uniform(-1, 1, size=(B, 100)).astype(np.float32),
                   y_real=eye[yr], y_gen=eye[rs.randint(10, size=B)], y_fake=eye[rs.randint(10, size=B)],
                   y_real_weights=np.linalg.inv(C)[yr].astype(np.float32))


KINK_EPS = 4e-6      # |rectifier input| <= this x the tensor's scale: the branch is decided by fp32 rounding


def _misfits(got, grads, grads32):
    """Tensors of the device's gradient that are not the oracle's: norm-relative error > 1e-2 (or 4x the float32 oracle's own
    distance from float64), largest entry off by > 1e-1 of the tensor's scale, or -- for tensors that are not noise-floor small --
    a projection <got, ref> / <ref, ref> more than 1.5e-2 from 1 (a dropped or mis-scaled term).  -> ([(name, message)], score):
    score = the worst tensor's largest ratio measured / bound (<= 1: everything fits)."""
    bad, score = [], 0.0
    gmax = max(float(np.abs(g).max()) for g in grads.values())
    for k, gref in grads.items():
        a = got[k]
        assert np.isfinite(a).all(), k
        scale = max(float(np.abs(gref).max()), 1e-3 * gmax)
        err = float(np.abs(a - gref).max()) / scale
        nrm = float(np.linalg.norm(a - gref)) / max(float(np.linalg.norm(gref)), scale)
        own = float(np.linalg.norm(grads32[k] - gref)) / max(float(np.linalg.norm(gref)), scale)
        rr = float((gref.astype(np.float64) ** 2).sum())
        proj = float((a.astype(np.float64) * gref).sum()) / rr if rr > 0 and float(np.abs(gref).max()) >= 1e-3 * gmax else 1.0
        sc = max(nrm / max(1e-2, 4 * own), err / 1e-1, abs(proj - 1.0) / max(1.5e-2, 4 * own))
        score = max(score, sc)
        if sc > 1.0:
            bad.append((k, "%s: norm-rel %.3e max %.3e of scale %.3e, projection %.4f (fp32 oracle norm-rel %.3e)" % (k, nrm, err, scale, proj, own)))
    return bad, score


def _cmp(tag, got, oracle, grads32):
    """HIP fp32 vs the float64 oracle, per tensor (``_misfits``).  ``oracle()`` evaluates the float64 gradients under the current
    ``oracle.tape.Kinks`` policy.  A rectifier input within fp32 rounding of zero takes the device's branch, not the float64
    oracle's, and at B = 8 ONE such unit moves whole tensors by a few per cent (round 4: "unbiased", second generator run, 3.1e-2
    on g_h0_lin / g_h1_lin with the split-reduction GEMMs on AND off).  Instead of a wider bound for every tensor (round 4 accepted
    6e-2 anywhere), the oracle lists its undecidable units (|input| <= KINK_EPS of the tensor's scale) and the device's gradient has to
    be the oracle's for SOME assignment of them -- every tensor of the run at the strict bounds under ONE assignment, found greedily
    (smallest |input| first, a flip is kept when it lowers the worst measured / bound ratio; two passes)."""
    from oracle.tape import Kinks
    try:
        Kinks.reset(eps=KINK_EPS)
        bad, score = _misfits(got, oracle(), grads32)
        found = list(Kinks.found)
        flips = []
        if bad:
            assert 0 < len(found) <= 16, "%s: %s and %d undecidable rectifier inputs" % (tag, bad[0][1], len(found))
            order = sorted(range(len(found)), key=lambda i: abs(found[i][2]))
            for _ in range(2):
                for i in order:
                    if not bad:
                        break
                    trial_flips = [f for f in flips if f != i] if i in flips else flips + [i]
                    Kinks.reset(eps=KINK_EPS, flip=trial_flips)
                    trial, tscore = _misfits(got, oracle(), grads32)
                    if tscore < score:
                        flips, bad, score = trial_flips, trial, tscore
        assert not bad, "%s (undecidable units %s, flipped %s): %s" % (tag, found, flips, "; ".join(m for _, m in bad))
        return flips
    finally:
        Kinks.reset()


CASES = [("rcgan", "projection", False, "hinge", False), ("rcgan", "projection", True, "hinge", False),
         ("unbiased", "projection", False, "hinge", False), ("biased", "vanilla", False, "ce", False),
         ("rcgan", "projection", False, "hinge", True),
         # (round 6) the flag combinations outside the run_*.sh presets: per-label passes (unbiased, estimate_confuse) through a
         # discriminator whose convolutions see the label -- ONE 10 x B pass here, ten discriminator() calls in the oracle as in the
         # reference (model.py:152-163,187-197)
         ("unbiased", "vanilla", False, "ce", False), ("rcgan", "vanilla", True, "hinge", False),
         ("rcgan", "projection", True, "hinge", True), ("unbiased", "projection", False, "hinge", True)]


@pytest.mark.parametrize("alg,disc,est,loss,concat", CASES)
def test_mnist_iteration_parity_fp32(alg, disc, est, loss, concat):
    import rcgan_amd  # noqa: F401
    from rcgan_amd.mnist import MnistRCGAN, create_variables
    rs = np.random.RandomState(41)
    B = 8
    layers = (1, 3) if concat else ()
    variables = create_variables(0, disc, est, True, disc == "projection", layers)
    gs, ds, cs, S, U = variables
    jit = np.random.RandomState(3)

    def jitter(specs):
        out = []
        for n, shp, v in specs:
            if n.endswith(("/bias", "/biases", "/beta", "/gamma")):
                v = (v + 0.1 * jit.randn(*shp)).astype(np.float32)
            out.append((n, shp, v))
        return out
    variables = (jitter(gs), jitter(ds), cs, S, U)
    C, b = _batch(rs, B)
    m = MnistRCGAN(algorithm=alg, alpha=0.3, batch_size=B, dtype="f32", disc_type=disc, loss_fn=loss, estimate_confuse=est,
                   perm_regularizer=True, perm_multiplier=10.0, spectral_norm=disc == "projection", max_norm=True,
                   concat_y=concat, concat_y_layers=layers, use_graphs=False, variables=variables)
    try:
        # the product's variable creation equals the oracle's
        P, S0, U0 = om.init_params(0, disc, est, True, disc == "projection", layers)
        for n, shp, v in gs + ds + cs:
            assert np.array_equal(v, P[n]), n
        P = {n: v.copy() for n, _, v in variables[0] + variables[1] + variables[2]}
        So = {k: v.copy() for k, v in S.items()}
        Uo = {k: v.copy() for k, v in U.items()}
        cfg = dict(algorithm=alg, disc_type=disc, estimate_confuse=est, loss_fn=loss, perm_regularizer=True, perm_multiplier=10.0,
                   spectral_norm=disc == "projection", C=C, concat_y=concat, concat_y_layers=layers, max_norm=True,
                   confuse_multiplier=10.0)
        m.set_inputs(**b)
        # ---- D run
        d64 = lambda: om.d_grads(P, {k: v.copy() for k, v in So.items()}, dict(Uo), cfg, b, dtype=np.float64)
        L64, _ = d64()
        _, g32 = om.d_grads(P, {k: v.copy() for k, v in So.items()}, dict(Uo), cfg, b, dtype=np.float32)
        m.d_step()
        got = m.losses()
        for k in ("d_loss_real", "d_loss_fake", "class_loss_real"):
            assert abs(got[k] - L64[k]) <= 2e-5 * max(1.0, abs(L64[k])), (k, got[k], L64[k])
        _cmp("D grad", m.get_grads(m.PD), lambda: d64()[1], g32)
        tr = om.Trainer(P, So, Uo, cfg)
        tr.d_step(b)
        st = m.get_state()
        for k in So:
            if k.startswith("generator/"):
                assert_close(st[k], So[k], 1e-4, "moving " + k)
        for k in Uo:
            assert_close(st[k], Uo[k], 1e-4, "u " + k)
        newp = m.get_params()
        if m.clip_range is not None:
            for k in ("discriminator/d_h4_lin/Matrix", "discriminator/d_h5_y_lin/Matrix", "discriminator/d_h5_y_lin/bias"):
                assert np.abs(newp[k]).max() <= 1.0
        # ---- two G runs from the device's own post-D state
        for run in range(2):
            newp, st = m.get_params(), m.get_state()
            for k in P:
                P[k] = newp[k].copy()
            for k in So:
                So[k] = st[k].copy()
            for k in Uo:
                Uo[k] = st[k].copy()
            g64f = lambda: om.g_grads(P, {k: v.copy() for k, v in So.items()}, dict(Uo), cfg, b, dtype=np.float64)
            L64, _ = g64f()
            _, g32 = om.g_grads(P, {k: v.copy() for k, v in So.items()}, dict(Uo), cfg, b, dtype=np.float32)
            m.g_step()
            got = m.losses()
            for k in ("g_loss", "class_loss_fake"):
                assert abs(got[k] - L64[k]) <= 2e-5 * max(1.0, abs(L64[k])), (k, got[k], L64[k])
            gg = m.get_grads(m.PG)
            if m.PC is not None:
                gg.update(m.get_grads(m.PC))
            _cmp("G grad run %d" % run, gg, lambda: g64f()[1], g32)
    finally:
        m.ctx.close()


def test_mnist_sampler_and_graph_replay():
    import rcgan_amd  # noqa: F401
    from rcgan_amd.mnist import MnistRCGAN, create_variables
    rs = np.random.RandomState(43)
    B = 8
    C, b = _batch(rs, B)
    outs = []
    for graphs in (False, True):
        variables = create_variables(0, "projection", True, True, True, ())
        m = MnistRCGAN(algorithm="rcgan", batch_size=B, dtype="f32", estimate_confuse=True, use_graphs=graphs, variables=variables)
        try:
            m.set_inputs(**b)
            for _ in range(3):
                m.iteration()
            outs.append((m.get_params(), m.get_state(), m.sampler(b["z"], b["y_gen"])))
        finally:
            m.ctx.close()
    (pa, sa, xa), (pb, sb, xb) = outs
    for k in pa:
        assert np.array_equal(pa[k], pb[k]), k
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(xa, xb) and xa.shape == (B, 28, 28, 1) and (xa > 0).all() and (xa < 1).all()
    # sampler == oracle gen_sampler on the same weights and moving statistics
    S = {k: v for k, v in sa.items() if "moving" in k}
    ref = om.sampler(pa, S, b["z"], b["y_gen"])
    assert_close(xa, ref, 1e-4, "gen_sampler")


def test_recover_labels_matches_oracle():
    """DCGAN.recover_labels (mnist/model.py:494-640): SGD on the latent z of every (image, label) pair and
    on the label logits through the frozen sampler (inference-mode batch norm).  Product (HIP, fp32) vs the float64
    oracle over three updates from identical initial variables."""
    import rcgan_amd  # noqa: F401
    from rcgan_amd.mnist import MnistRCGAN, create_variables
    R = 4
    rs = np.random.RandomState(9)
    gs, ds, cs, S, U = create_variables(0, "projection", False, True, True, ())
    S = {k: (rs.uniform(0.5, 1.5, size=v.shape) if k.endswith("moving_variance") else rs.uniform(-0.2, 0.2, size=v.shape)).astype(np.float32)
         for k, v in S.items()}
    m = MnistRCGAN(algorithm="rcgan", alpha=0.3, batch_size=R * 10, dtype="f32", disc_type="projection", use_graphs=False,
                   variables=(gs, ds, cs, S, U))
    try:
        images = rs.rand(R, 28, 28, 1).astype(np.float32)
        y_actual = np.eye(10)[rs.randint(10, size=R)]
        lr, seed, steps = 50.0, 4, 3
        hist = []
        res = m.recover_labels(images, y_actual, epochs=steps, learning_rate=lr, seed=seed, log_every=1, log=hist.append)
        # the oracle from the same glorot-uniform start (creation order: logits, then z)
        g = np.random.RandomState(seed)
        glorot = lambda shape: g.uniform(-np.sqrt(6.0 / sum(shape)), np.sqrt(6.0 / sum(shape)), size=shape).astype(np.float32)
        logits, z = glorot((R, 10)).astype(np.float64), glorot((R * 10, 100)).astype(np.float64)
        P = {n: v.astype(np.float64) for n, _, v in gs + ds + cs}
        S64 = {k: v.astype(np.float64) for k, v in S.items()}
        losses = []
        for _ in range(steps):
            loss, z, logits, yrec = om.recover_step(P, S64, z, logits, images.reshape(R, -1), lr)
            losses.append(loss)
        got_losses = [h[1] for h in res["history"]]
        assert np.allclose(got_losses, losses, rtol=2e-5), (got_losses, losses)
        assert losses[-1] < losses[0]                      # the descent direction is a descent direction
        assert_close(res["z_recover"], z, 2e-4, "z after %d SGD steps" % steps)
        sm = np.exp(logits - logits.max(1, keepdims=True))
        assert_close(res["y_recover"], sm / sm.sum(1, keepdims=True), 2e-5, "softmax of the recovered logits")
        assert len(hist) == steps and hist[0].startswith("Recover Epoch: [ 0] time:")
        with pytest.raises(ValueError):
            m.recover_labels(images[:2], y_actual[:2], epochs=1)
    finally:
        m.ctx.close()


@pytest.mark.parametrize("graphs", [False, True])
def test_fused_iteration_equals_separate_steps(graphs, monkeypatch):
    """MnistRCGAN.iteration() with the first generator step reusing the D step's generator forward (same z, y_gen and
    generator weights: model.py:347-372) == d_step(); g_step(); g_step() -- parameters, Adam state through the parameters
    after three iterations, batch-norm moving averages (two identical updates == one with the decay squared) and losses."""
    import rcgan_amd  # noqa: F401
    from rcgan_amd.mnist import MnistRCGAN, create_variables
    rs = np.random.RandomState(47)
    B = 8
    batches = [_batch(rs, B)[1] for _ in range(3)]
    outs = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("RCGAN_MNIST_FUSE_G", fuse)
        variables = create_variables(0, "projection", True, True, True, ())
        m = MnistRCGAN(algorithm="rcgan", batch_size=B, dtype="f32", estimate_confuse=True, use_graphs=graphs, variables=variables)
        assert m.fuse_first_g == (fuse == "1")
        try:
            for b in batches:
                m.set_inputs(**b)
                m.iteration()
            outs.append((m.get_params(), m.get_state(), m.losses()))
        finally:
            m.ctx.close()
    (pa, sa, la), (pb, sb, lb) = outs
    for k in pa:
        assert_close(pb[k], pa[k], 2e-4, "parameter " + k)
        assert rel_err(pb[k], pa[k]) <= 2e-5, (k, rel_err(pb[k], pa[k]))
    for k in sa:
        assert_close(sb[k], sa[k], 1e-5, "state " + k)
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-5 * max(1.0, abs(la[k])), (k, la[k], lb[k])
