"""Step-level parity of the CIFAR RCGAN engine (HIP, through the C ABI) against the numpy oracle:
identical initial weights, identical fed batch -> loss, every parameter gradient, SN ``u`` state and the
Adam-updated weights after a D step and a G step.

fp32 activations: per-tensor max error <= 2e-3 of that gradient's max magnitude (deep fp32 chains:
~30 layers of accumulation-order noise; typical observed error is 1e-5..1e-4).
bf16 activations (fp32 master weights / accumulation): norm-relative error <= 8e-2 per tensor -- bf16
rounding (2^-9 per stored activation) random-walks through ~30 layers of forward and backward.
"""
import numpy as np
import pytest

from oracle import cifar as oc
from tests.gpu_util import assert_close, rel_err

pytestmark = pytest.mark.gpu


def _batches(rs, B, alpha=0.6):
    C = oc.c_alpha(alpha)
    Cinv = np.linalg.inv(C)
    lab = rs.randint(10, size=B)
    raw = dict(images=rs.randint(0, 256, size=(B, 3072)), noise=rs.uniform(0, 1 / 128., size=(B, 3072)).astype(np.float32),
               labels=lab, labels_random=rs.randint(10, size=B), labels_biased=rs.randint(10, size=B),
               inv_weights=Cinv[lab].astype(np.float32), z=rs.randn(B, 128).astype(np.float32))
    g = dict(labels_random_G=rs.randint(10, size=2 * B), labels_biased_G=rs.randint(10, size=2 * B),
             z_G=rs.randn(2 * B, 128).astype(np.float32))
    return C, raw, g


def _make(alg, perm, B, dtype, use_graphs=False):
    import rcgan_amd  # noqa: F401
    from rcgan_amd.cifar import CifarRCGAN, create_variables
    variables = create_variables(0, alg, perm, "linear", True, 0.2)
    rs = np.random.RandomState(3)
    # de-trivialise zero-initialised tensors (biases, condBN tables) so their gradients matter
    gs, ds, cs, U = variables
    def jitter(specs):
        out = []
        for n, shp, v in specs:
            if n.endswith("/Biases") or n.endswith("/b") or "CondBatchNorm" in n:
                v = (v + 0.1 * rs.randn(*shp)).astype(np.float32)
            out.append((n, shp, v))
        return out
    variables = (jitter(gs), jitter(ds), cs, U)
    m = CifarRCGAN(algorithm=alg, alpha=0.6, batch_size=B, dtype=dtype, perm_classifier=perm, perm_multiplier=1.0,
                   use_graphs=use_graphs, device_rng=False, variables=variables, arena_bytes=2 << 30)
    P = {n: v.copy() for n, _, v in variables[0] + variables[1] + variables[2]}
    Uo = {k: v.copy() for k, v in U.items()}
    return m, P, Uo


def _labels_all(alg, raw):
    second = raw["labels_random"] if alg in ("biased", "unbiased") else raw["labels_biased"]
    return np.concatenate([raw["labels"], second])


def _check(m, P, Uo, alg, perm, raw, gb, C, tol_kind):
    """Reference = the oracle evaluated in float64 on the same weights / batch.  (The fp32 oracle is itself
    2e-4..4e-3 away from the fp64 result on these gradients -- batch-norm over 4 samples is badly
    conditioned -- and the HIP fp32 path lands at the same distance, see tests/tools/debug_gstep.py.)"""
    from rcgan_amd import _lib as L
    cfg = dict(algorithm=alg, C=C, perm_classifier=perm, perm_multiplier=1.0)
    bf16 = m.ctx.act_dtype in (L.BF16, L.F16)      # 16-bit activations (the fp16 build scales its losses: losses() and
                                                    # get_grads() report unscaled values)

    def cmp_all(tag, got, grads, bf16_tol):
        gmax = max(float(np.abs(g).max()) for g in grads.values())
        for k, gref in grads.items():
            a = got[k]
            assert np.isfinite(a).all(), k
            floor = 1e-3 * gmax          # conv biases feeding a batch norm have an exactly-zero true gradient
            scale = max(float(np.abs(gref).max()), floor)
            err = float(np.abs(a - gref).max()) / scale
            if not bf16:
                # norm-relative 1e-2; the max-error bound is looser because one ReLU input within ~1e-7 of zero
                # may take the other branch under a different fp32 summation order (tests/test_gpu_mnist_step.py)
                nrm = float(np.linalg.norm(a - gref)) / max(float(np.linalg.norm(gref)), scale)
                assert nrm <= 1e-2 and err <= 1e-1, "%s %s: norm-rel %.3e max err %.3e of scale %.3e" % (tag, k, nrm, err, scale)
            elif np.size(gref) <= 1:
                # one-element gradients (D.Output/b) are sums of +-1/B hinge indicators: a logit crossing the
                # hinge threshold under bf16 rounding moves them by a whole 1/B step
                assert abs(float(np.ravel(a)[0]) - float(np.ravel(gref)[0])) <= 1.5 / B_CUR, k
            elif float(np.abs(gref).max()) > floor:
                e = rel_err(a, gref)
                cos = float((a.astype(np.float64) * gref).sum() / (np.linalg.norm(a) * np.linalg.norm(gref) + 1e-30))
                assert e <= bf16_tol and cos >= 0.95, "%s %s: norm-rel %.3e cos %.4f" % (tag, k, e, cos)
            else:
                assert err <= 0.5, "%s %s: %.3e" % (tag, k, err)

    # ---- D step
    m.set_inputs(labels_all=_labels_all(alg, raw), **raw)
    m.d_step(iteration=0)
    ob = dict(real=oc.preprocess_real(raw["images"], raw["noise"]), labels=raw["labels"], labels_random=raw["labels_random"],
              labels_biased=raw["labels_biased"], inv_weights=raw["inv_weights"], z=raw["z"])
    cost, grads = oc.d_grads(P, dict(Uo), cfg, ob, dtype=np.float64)
    tr = oc.Trainer(P, Uo, cfg, lr=2e-4)
    tr.d_step(0, ob)                       # fp32 oracle step: reference for the Adam update and the u state
    d_loss, _ = m.losses()
    assert abs(d_loss - cost) <= (5e-3 if bf16 else 2e-5) * max(1.0, abs(cost)), (d_loss, cost)
    cmp_all("D grad", m.get_grads(m.PD), grads, 8e-2)
    st = m.get_state()
    for k in Uo:
        assert_close(st[k], Uo[k], 2e-2 if bf16 else 1e-4, "u " + k)
    newp = m.get_params()
    if not bf16:
        for k in grads:
            # Adam(beta1=0) moves each weight by ~lr*sign(g); a near-zero gradient can flip sign under fp32
            # noise and move that weight by 2*lr: allow a few
            upd, uref = newp[k] - m_init[k], P[k] - m_init[k]
            sure = np.abs(grads[k]) > 1e-6 * max(float(np.abs(g).max()) for g in grads.values())
            if sure.any():
                bad = np.mean(np.abs(upd - uref)[sure] > 0.1 * 2e-4)
                assert bad < 0.02, "D update %s: %.3f of elements differ" % (k, bad)
    # ---- G step, from the device's own post-D-step weights and state
    for k in P:
        P[k] = newp[k].copy()
    for k in Uo:
        Uo[k] = st[k].copy()
    m.set_inputs(**gb)
    m.g_step(iteration=1)
    og = dict(labels_random_G=gb["labels_random_G"], labels_biased_G=gb["labels_biased_G"], z=gb["z_G"])
    cost, grads = oc.g_grads(P, dict(Uo), cfg, og, dtype=np.float64)
    tr.g_step(1, og)
    _, g_loss = m.losses()
    assert abs(g_loss - cost) <= (2e-2 if bf16 else 2e-5) * max(1.0, abs(cost)), (g_loss, cost)
    got = m.get_grads(m.PG)
    if m.PC is not None:
        got.update(m.get_grads(m.PC))
    # bf16: the generator gradient crosses ~45 bf16-stored layers (G forward, D forward, D and G backward)
    # at a batch of 8 fakes; measured norm-relative error is 0.10-0.16 (cosine >= 0.98)
    cmp_all("G grad", got, grads, 2.5e-1)
    st = m.get_state()
    for k in Uo:
        assert_close(st[k], Uo[k], 2e-2 if bf16 else 1e-4, "u(after G) " + k)


m_init = {}
B_CUR = 8


@pytest.mark.parametrize("alg,perm", [("rcgan", False), ("rcgan-u", True), ("biased", False), ("unbiased", False)])
def test_step_parity_fp32(alg, perm):
    global m_init
    rs = np.random.RandomState(21)
    B = 4
    C, raw, gb = _batches(rs, B)
    m, P, Uo = _make(alg, perm, B, "f32")
    m_init = {k: v.copy() for k, v in P.items()}
    try:
        _check(m, P, Uo, alg, perm, raw, gb, C, "f32")
    finally:
        m.ctx.close()


@pytest.mark.parametrize("alg,perm", [("rcgan", False), ("rcgan-u", True)])
def test_step_parity_bf16(alg, perm):
    global m_init
    rs = np.random.RandomState(22)
    B = B_CUR
    C, raw, gb = _batches(rs, B)
    m, P, Uo = _make(alg, perm, B, "bf16")
    m_init = {k: v.copy() for k, v in P.items()}
    try:
        _check(m, P, Uo, alg, perm, raw, gb, C, "bf16")
    finally:
        m.ctx.close()


def test_step_parity_f16():
    """The fp16 build of the library (BASELINE config 5: v_mfma_f32_16x16x32_f16, static loss scale 1024) against the
    same float64 oracle step, at the bf16 tolerances (fp16 keeps three more significand bits)."""
    global m_init
    rs = np.random.RandomState(22)
    B = B_CUR
    C, raw, gb = _batches(rs, B)
    m, P, Uo = _make("rcgan", False, B, "f16")
    assert m.loss_scale == 1024.0
    m_init = {k: v.copy() for k, v in P.items()}
    try:
        _check(m, P, Uo, "rcgan", False, raw, gb, C, "bf16")
    finally:
        m.ctx.close()


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-5), ("bf16", 2e-2)])
def test_segmented_generator_forward_equals_separate_calls(dtype, tol):
    """Generator(segments=k) on k batches stored back to back == k separate Generator() calls: the convolutions see one
    big batch, every conditional batch norm takes its statistics per segment (rcgan_bn_fwd_segments)."""
    rs = np.random.RandomState(31)
    B, K = 6, 3
    m, P, Uo = _make("rcgan", False, B, dtype)
    try:
        lab = rs.randint(10, size=K * B)
        z = rs.randn(K * B, 128).astype(np.float32)
        whole = m.sample(lab, z, segments=K)
        parts = np.concatenate([m.sample(lab[k * B:(k + 1) * B], z[k * B:(k + 1) * B]) for k in range(K)])
        pooled = m.sample(lab, z)                 # statistics over all K*B samples: must differ
        assert_close(whole, parts, tol, "segmented generator forward")
        assert np.abs(pooled - parts).max() > 10 * tol * np.abs(parts).max()
    finally:
        m.ctx.close()


@pytest.mark.parametrize("overlap", ["1", "0"])
@pytest.mark.parametrize("alg,dtype,tol", [("rcgan", "f32", 3e-3), ("rcgan", "bf16", 3e-2), ("rcgan-u", "f32", 3e-3)])
def test_batched_critic_fakes_equal_per_step_generator(alg, dtype, tol, overlap, monkeypatch):
    """prepare_critic_fakes() + N_CRITIC d_step() == N_CRITIC plain d_step() (the generator forward inside every critic
    step, as the reference runs it) on the same z / labels / real batches: discriminator weights after the five Adam
    updates and the last critic loss agree.  (Adam moves a weight whose gradient is ~0 by up to lr per step whatever the
    gradient's size, so in fp32 single elements may differ by up to 5 * 2 * lr = 2e-3 ABSOLUTE -- that is the element bound -- while
    the norm-relative error, the criterion that matters, stays <= 2e-4.)"""
    from rcgan_amd.cifar import N_CRITIC
    # overlap "1" (round 6, the default): prepare_critic_fakes() = N_CRITIC per-step passes on the generator-forward stream, every
    # d_step() waiting for its slice; "0": the one batched pass on the step stream
    monkeypatch.setenv("RCGAN_OVERLAP_GF", overlap)
    rs = np.random.RandomState(41)
    B = 4
    steps = []
    for _ in range(N_CRITIC):
        C, raw, _ = _batches(rs, B)
        steps.append(raw)
    outs = []
    for batched in (False, True):
        m, P, Uo = _make(alg, alg == "rcgan-u", B, dtype)
        assert m.overlap_gf == (overlap == "1")
        try:
            if batched:
                m.set_inputs(labels_random_all=np.concatenate([r["labels_random"] for r in steps]),
                             z_all=np.concatenate([r["z"] for r in steps]))
                m.prepare_critic_fakes()
            for it, raw in enumerate(steps):
                m.set_inputs(labels_all=_labels_all(alg, raw), **raw)
                m.d_step(iteration=0)
            assert m._fakes_left == 0
            outs.append((m.get_params(), m.losses()[0]))
        finally:
            m.ctx.close()
    (pa, la), (pb, lb) = outs
    assert abs(la - lb) <= tol * max(1.0, abs(la)), (la, lb)
    for k in pa:
        if k.startswith("Discriminator") or "D." in k:
            if dtype == "bf16":
                assert_close(pb[k], pa[k], tol, "D weights after %d critic steps: %s" % (N_CRITIC, k))
                assert rel_err(pb[k], pa[k]) <= 5e-3, (k, rel_err(pb[k], pa[k]))
            else:
                # fp32: the two runs differ in fp32 summation order only (the batched pass runs its GEMMs at 5B rows: other tile /
                # split-reduction choices).  Adam (beta1 = 0) moves a weight whose gradient is ~0 by lr * sign(noise) per step, so
                # an element may end up to 2 * lr * N_CRITIC = 2e-3 (absolute) apart; everything else agrees to fp32 rounding,
                # i.e. the tensor as a whole to ~1e-4 (measured 8.4e-5 on D.Block.3.Conv1 with the round-4 split-reduction GEMMs,
                # 2e-5 before them)
                d = np.abs(pb[k].astype(np.float64) - pa[k])
                assert float(d.max()) <= 2 * 2e-4 * N_CRITIC + 1e-6, (k, float(d.max()))
                assert rel_err(pb[k], pa[k]) <= 2e-4, (k, rel_err(pb[k], pa[k]))


@pytest.mark.parametrize("use_graphs", [False, True])
def test_step_input_rider_equals_separate_launches(use_graphs):
    """Critic steps with the device random stream: noise, preprocessing, image pool and zero-fill riding in the filter-preparation
    launch (rcgan_conv_prepare_batch_riders) against the five separate launches -- same stream, same bits: discriminator weights,
    losses and the stream offset after two iterations' worth of critic steps are identical."""
    import rcgan_amd  # noqa: F401
    from rcgan_amd.cifar import CifarRCGAN, N_CRITIC
    rs = np.random.RandomState(77)
    B = 8
    feeds = []
    for _ in range(2 * N_CRITIC):
        _, raw, _ = _batches(rs, B)
        feeds.append(raw)
    outs = []
    for ride in (True, False):
        m = CifarRCGAN(algorithm="rcgan", alpha=0.6, batch_size=B, dtype="bf16", seed=5, use_graphs=use_graphs, device_rng=True,
                       arena_bytes=2 << 30)
        m.ride_inputs = ride
        try:
            for it in range(3):
                # (iteration 1 uses only three of its five fake batches: the device's slice counter must come back to 0)
                chunk = feeds[(it % 2) * N_CRITIC:(it % 2 + 1) * N_CRITIC]
                m.set_inputs(labels_random_all=np.concatenate([r["labels_random"] for r in chunk]))
                m.prepare_critic_fakes()
                for raw in (chunk[:3] if it == 1 else chunk):
                    r = {k: v for k, v in raw.items() if k not in ("z", "noise")}
                    m.set_inputs(labels_all=_labels_all("rcgan", raw), **r)
                    m.d_step(iteration=it)
            m.ctx.sync()
            outs.append((m.get_params(), m.losses(), m.rng_state.cpu().numpy().copy()))
        finally:
            m.ctx.close()
    (pa, la, sa), (pb, lb, sb) = outs
    assert np.array_equal(sa, sb), (sa, sb)
    assert la == lb, (la, lb)
    for k in pa:
        assert np.array_equal(pa[k], pb[k]), k


def test_graph_replay_matches_eager():
    """The captured hipGraph of a D step / G step must reproduce the eager launches bit for bit."""
    rs = np.random.RandomState(23)
    B = 4
    C, raw, gb = _batches(rs, B)
    outs = []
    for graphs in (False, True):
        m, P, Uo = _make("rcgan", False, B, "bf16", use_graphs=graphs)
        try:
            for it in range(3):      # graph mode: warm-up+capture on call 1, replays on calls 2 and 3
                m.set_inputs(labels_all=_labels_all("rcgan", raw), **raw)
                m.d_step(iteration=it)
                m.set_inputs(**gb)
                m.g_step(iteration=it)
            outs.append((m.get_params(), m.losses()))
        finally:
            m.ctx.close()
    (pa, la), (pb, lb) = outs
    assert la == lb, (la, lb)
    for k in pa:
        assert np.array_equal(pa[k], pb[k]), k


def test_multi_step_trajectory_fp32():
    """3 iterations of [G step, 2 D steps] track the oracle (weights after Adam)."""
    rs = np.random.RandomState(24)
    B = 4
    m, P, Uo = _make("rcgan", False, B, "f32")
    C = oc.c_alpha(0.6)
    tr = oc.Trainer(P, Uo, dict(algorithm="rcgan", C=C), lr=2e-4)
    try:
        for it in range(2):
            _, raw, gb = _batches(rs, B)
            if it > 0:
                m.set_inputs(**gb)
                m.g_step(iteration=it)
                tr.g_step(it, dict(labels_random_G=gb["labels_random_G"], labels_biased_G=gb["labels_biased_G"], z=gb["z_G"]))
            for _ in range(2):
                m.set_inputs(labels_all=_labels_all("rcgan", raw), **raw)
                m.d_step(iteration=it)
                tr.d_step(it, dict(real=oc.preprocess_real(raw["images"], raw["noise"]), labels=raw["labels"],
                                   labels_random=raw["labels_random"], labels_biased=raw["labels_biased"],
                                   inv_weights=raw["inv_weights"], z=raw["z"]))
        newp = m.get_params()
        worst = 0.0
        for k in P:
            worst = max(worst, float(np.abs(newp[k] - P[k]).max()))
        # lr = 2e-4, <= 4 Adam updates per tensor: a sign flip of a near-zero gradient moves a weight by 2*lr
        assert worst <= 1e-3, worst
        far = np.mean([np.mean(np.abs(newp[k] - P[k]) > 1e-4) for k in P])
        assert far < 0.02, far
    finally:
        m.ctx.close()


def test_packed_feed_equals_named_inputs():
    """set_feed (one copy of a packed batch) fills exactly what set_inputs fills field by field."""
    import rcgan_amd  # noqa: F401
    from rcgan_amd.cifar import CifarRCGAN
    m = CifarRCGAN(algorithm="unbiased", alpha=0.6, batch_size=4, dtype="f32", seed=0, device=0, use_graphs=False)
    try:
        rs = np.random.RandomState(2)
        B = 4
        d = dict(images=rs.randint(0, 256, size=(B, 3072)), labels=rs.randint(10, size=B), labels_random=rs.randint(10, size=B),
                 labels_biased=rs.randint(10, size=B), inv_weights=rs.randn(B, 10).astype(np.float32), labels_all=rs.randint(10, size=2 * B))
        g = dict(labels_random_G=rs.randint(10, size=2 * B), labels_biased_G=rs.randint(10, size=2 * B))
        m.set_feed("d", m.pack_feed("d", **d))
        m.set_feed("g", m.pack_feed("g", **g))
        got = {k: m.ctx.download(m.inp[k]) for k in list(d) + list(g)}
        m.set_inputs(**{k: np.zeros_like(v) for k, v in list(d.items()) + list(g.items())})
        m.set_inputs(**d, **g)
        for k, v in list(d.items()) + list(g.items()):
            assert np.array_equal(got[k], np.asarray(v).reshape(got[k].shape).astype(got[k].dtype)), k
            assert np.array_equal(m.ctx.download(m.inp[k]), got[k]), k
    finally:
        m.ctx.close()


def test_dynamic_loss_scale_skips_overflowed_steps_and_recovers():
    """fp16 activations, dynamic loss scale (rcgan_grad_finite_check / rcgan_adam_tf_dyn / rcgan_loss_scale_update): from an
    absurd initial scale every step overflows in the backward pass -- the update is skipped (weights, Adam slots and the
    bias-correction count untouched), the scale halves -- until the gradients fit; then steps apply, and after
    growth_interval applied steps in a row the scale doubles.  No inf / nan ever reaches the weights."""
    import rcgan_amd  # noqa: F401
    from rcgan_amd.cifar import CifarRCGAN
    rs = np.random.RandomState(5)
    B = 8
    C, raw, gb = _batches(rs, B)
    m = CifarRCGAN(algorithm="rcgan", alpha=0.6, batch_size=B, dtype="f16", seed=0, use_graphs=True, device_rng=False,
                   loss_scale=2.0 ** 40, loss_scale_growth_interval=3, arena_bytes=2 << 30)
    assert m.dynamic_ls
    try:
        m.set_inputs(labels_all=_labels_all("rcgan", raw), **raw)
        p0 = m.get_params()
        m.d_step(iteration=0)
        st = m.loss_scale_state()
        assert st["skipped_steps"] == 1 and st["scale"] == 2.0 ** 39, st
        p1 = m.get_params()
        for k in p0:
            assert np.array_equal(p0[k], p1[k]), "a skipped step changed " + k
        assert m.PD.steps_applied() == 0 and float(m.PD.m.abs().max()) == 0.0 and float(m.PD.v.abs().max()) == 0.0
        applied_at = None
        for i in range(60):
            m.d_step(iteration=0)
            st = m.loss_scale_state()
            if m.PD.steps_applied() > 0:
                applied_at = st
                break
        assert applied_at is not None, st
        assert applied_at["scale"] == 2.0 ** (40 - applied_at["skipped_steps"]) and 2.0 ** 8 <= applied_at["scale"] <= 2.0 ** 30, applied_at
        d_loss, _ = m.losses()
        assert np.isfinite(d_loss) and 0.1 < d_loss < 10.0, d_loss          # the loss value is never scaled
        # generator step under the same scale: both of its groups are checked
        m.set_inputs(**gb)
        m.g_step(iteration=1)
        # growth: three applied steps in a row double the scale (an overflow in between would reset the count)
        s0, k0 = m.loss_scale_state(), m.PD.steps_applied()
        for _ in range(6):
            m.d_step(iteration=1)
        s1 = m.loss_scale_state()
        skipped = s1["skipped_steps"] - s0["skipped_steps"]
        assert m.PD.steps_applied() + skipped - k0 == 6
        if skipped == 0:       # (good_steps + 6) // 3 doublings, never beyond 2^24 and never downwards
            want = s0["scale"]
            for _ in range((s0["good_steps"] + 6) // 3):
                want = want if want >= 2.0 ** 24 else min(2 * want, 2.0 ** 24)
            assert s1["scale"] == want and s1["good_steps"] == (s0["good_steps"] + 6) % 3, (s0, s1)
        for k, v in m.get_params().items():
            assert np.isfinite(v).all(), k
    finally:
        m.ctx.close()


def test_dynamic_loss_scale_equals_static_scale_without_overflow():
    """With no overflow and no growth the dynamic path (scale read from device memory, finite check, device-side step count)
    is the static path bit for bit: weights after two critic steps and a generator step."""
    import rcgan_amd  # noqa: F401
    from rcgan_amd.cifar import CifarRCGAN
    rs = np.random.RandomState(6)
    B = 8
    C, raw, gb = _batches(rs, B)
    outs = []
    for dyn in (False, True):
        m = CifarRCGAN(algorithm="rcgan-u", alpha=0.6, batch_size=B, dtype="f16", seed=0, use_graphs=True, device_rng=False,
                       perm_classifier=True, confuse_init=True, dynamic_loss_scale=dyn, arena_bytes=2 << 30)
        try:
            for it in range(2):
                m.set_inputs(labels_all=_labels_all("rcgan-u", raw), **raw)
                m.d_step(iteration=it)
            m.set_inputs(**gb)
            m.g_step(iteration=1)
            m.set_inputs(labels_all=_labels_all("rcgan-u", raw), **raw)
            m.d_step(iteration=1)                    # replays of both graphs' captures come after this point in training
            assert m.loss_scale_state()["skipped_steps"] == 0
            outs.append((m.get_params(), m.losses()))
        finally:
            m.ctx.close()
    (pa, la), (pb, lb) = outs
    assert la == lb, (la, lb)
    for k in pa:
        assert np.array_equal(pa[k], pb[k]), k


def test_grad_finite_check_finds_any_non_finite_value():
    """rcgan_grad_finite_check through the raw C ABI: inf / nan anywhere in the slab (first element, a vector-tail element,
    the very last one) raises the flag; finite slabs -- denormals and the largest finite float included -- do not."""
    import ctypes as C
    import torch
    from tests.gpu_util import make_ctx
    ctx = make_ctx("bf16")
    try:
        n = 1000003
        base = torch.randn(n, device=ctx.device)
        base[7] = 3.4028234e38
        base[8] = 1e-45
        for pos, val, want in [(None, 0.0, 0.0), (0, float("inf"), 1.0), (n - 1, float("nan"), 1.0), (n - 2, float("-inf"), 1.0),
                               (123456, float("nan"), 1.0)]:
            g = base.clone()
            if pos is not None:
                g[pos] = val
            ls = torch.tensor([1024.0, 0, 0, 0], dtype=torch.float32, device=ctx.device)
            torch.cuda.synchronize()
            ctx.check(ctx.lib.rcgan_grad_finite_check(ctx.h, n, C.c_void_p(g.data_ptr()), C.c_void_p(ls.data_ptr())))
            ctx.sync()
            assert float(ls[2]) == want, (pos, val, ls)
    finally:
        ctx.close()


@pytest.mark.parametrize("alg,perm,dtype,use_graphs", [("rcgan", False, "bf16", True), ("rcgan-u", True, "bf16", False), ("rcgan", False, "f32", True)])
def test_optimiser_inside_the_spectral_norm_backward(alg, perm, dtype, use_graphs, monkeypatch):
    """(round 6) rcgan_sn_bwd_adam: the critic step whose last launch applies TF-Adam (the rows of dW a workgroup has just formed;
    rider workgroups for the biases / embeddings between the spectrally normalised weights; the step count on the device) against
    the separate optimiser launch (RCGAN_SN_ADAM=0) on the same weights and batches.  After the first step: gradients identical,
    parameters and both Adam slots equal to 2 ulp (same operation sequence, compiled into two kernels).  After four steps (a new
    learning rate each, graph replays from the second on): the host's step count, and the parameters within what Adam with
    beta1 = 0 allows two runs 2 ulp apart (a weight whose gradient is ~0 moves by lr * sign(noise): 2 * lr per step)."""
    rs = np.random.RandomState(5)
    B = 4
    steps = [_batches(rs, B)[1] for _ in range(4)]
    outs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("RCGAN_SN_ADAM", fused)
        m, P, Uo = _make(alg, perm, B, dtype, use_graphs=use_graphs)
        try:
            assert m.fused_tail == (fused == "1")
            first = None
            for it, raw in enumerate(steps):
                m.set_inputs(labels_all=_labels_all(alg, raw), **raw)
                m.d_step(iteration=it * 7000)            # (a different learning rate every step: lr_decay)
                if it == 0:
                    m.ctx.sync()
                    first = (m.get_grads(m.PD), {w: {n: m.PD.get(n, w) for n in m.PD.names} for w in ("value", "m", "v")})
            assert m._tail_fused.get(False, False) == (fused == "1")
            assert m.PD.t == len(steps)
            m.ctx.sync()
            outs.append((first, {n: m.PD.get(n, "value") for n in m.PD.names}))
        finally:
            m.ctx.close()
    ((ga, sa), fa), ((gb, sb), fb) = outs
    for k in ga:
        assert np.array_equal(ga[k], gb[k]), "gradient %s" % k
    for w in ("value", "m", "v"):
        for n in sa[w]:
            a, b = sa[w][n].astype(np.float64), sb[w][n].astype(np.float64)
            tol = 2 * np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)).astype(np.float64)
            assert np.all(np.abs(a - b) <= tol), "%s of %s: max diff %.3e" % (w, n, float(np.abs(a - b).max()))
    for n in fa:
        assert np.isfinite(fa[n]).all() and float(np.abs(fa[n].astype(np.float64) - fb[n]).max()) <= 2 * 2e-4 * len(steps) + 1e-6, n


@pytest.mark.parametrize("alg,perm,dtype", [("rcgan", False, "bf16"), ("rcgan-u", True, "bf16"), ("rcgan", False, "f32")])
def test_critic_steps_as_one_graph_equal_single_steps(alg, perm, dtype, monkeypatch):
    """(round 6) CifarRCGAN.critic_steps(): the N_CRITIC critic updates of an iteration as ONE captured graph (one hand-over of their
    batches, the optimiser inside each step's last launch, the step count on the device) against prepare_critic_fakes() + N_CRITIC
    d_step() calls: the same launches in the same order on the same random stream -- generator, discriminator and spectral-norm state
    after three iterations (capture + replays, a generator step between them, a new learning rate every iteration) bit for bit."""
    import rcgan_amd  # noqa: F401
    from rcgan_amd.cifar import CifarRCGAN, N_CRITIC
    rs = np.random.RandomState(23)
    B = 8
    its = []
    for _ in range(3):
        ds = []
        for _ in range(N_CRITIC):
            raw = _batches(rs, B)[1]
            d = {k: raw[k] for k in ("images", "labels", "labels_random", "labels_biased", "inv_weights")}
            d["labels_all"] = _labels_all(alg, raw)
            ds.append(d)
        its.append((ds, dict(labels_random_G=rs.randint(10, size=2 * B), labels_biased_G=rs.randint(10, size=2 * B))))
    outs = []
    for one_graph in ("1", "0"):
        monkeypatch.setenv("RCGAN_CRITIC_GRAPH", one_graph)
        m = CifarRCGAN(algorithm=alg, alpha=0.6, batch_size=B, dtype=dtype, seed=5, perm_classifier=perm, confuse_init=perm,
                       use_graphs=True, device_rng=True, arena_bytes=2 << 30)
        try:
            for it, (ds, g) in enumerate(its):
                if it > 0:
                    m.feed_host("g", **g)
                    m.g_step(iteration=it * 9000)
                m.feed_host("gf", labels_random_all=np.concatenate([d["labels_random"] for d in ds]))
                m.prepare_critic_fakes()
                assert m._critic_graph_ok() == (one_graph == "1")
                if one_graph == "1":
                    m.critic_steps(ds, iteration=it * 9000)
                else:
                    for d in ds:
                        m.feed_host("d", **d)
                        m.d_step(iteration=it * 9000)
            assert ("d5" in m._graphs) == (one_graph == "1")
            assert m.PD.t == 3 * N_CRITIC and m._fakes_left == 0
            m.ctx.sync()
            outs.append((m.get_params(), m.get_state(), m.losses()))
        finally:
            m.ctx.close()
    (pa, sa, la), (pb, sb, lb) = outs
    assert la == lb, (la, lb)
    for k in pa:
        assert np.array_equal(pa[k], pb[k]), k
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
