"""The projection head's deferred parameter gradients (csrc/head_rider.h): dE = dlogit^T feat riding in the 8x8 stage's backward
launch, the dW_e / dtable / dw_out / db_e sums riding in the grouped filter-gradient launch -- against the same step with the two
launches on their own (RCGAN_HEAD_RIDERS=0).  Same arithmetic in the same order: weights, losses and spectral-norm state after two
iterations of the production loop (capture + replays) are equal bit for bit."""
import numpy as np
import pytest

from tests.test_gpu_dp import _feeds, _model, _run_iterations

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("alg,dtype,trunk", [("rcgan", "bf16", "1"), ("rcgan-u", "bf16", "1"), ("rcgan", "f16", "1"), ("rcgan", "bf16", "0"),
                                             ("biased", "bf16", "1")])
def test_riding_parameter_gradients_equal_their_own_launches(alg, dtype, trunk, monkeypatch):
    import rcgan_amd  # noqa: F401
    from rcgan_amd import cifar as cm
    from rcgan_amd import ops as O
    rs = np.random.RandomState(21)
    B = 8
    its = _feeds(rs, B, 2, alg)
    monkeypatch.setattr(cm, "FUSED_TRUNK", trunk == "1", raising=False)
    outs = []
    for riders in (False, True):
        monkeypatch.setattr(O, "HEAD_RIDERS", riders)
        m = _model(alg, dtype, B)
        try:
            outs.append(_run_iterations(m, its))
            assert m.ctx.lib.rcgan_head_flush(m.ctx.h) == 0          # nothing may be left pending (a no-op)
        finally:
            m.ctx.close()
    (pa, la, sa), (pb, lb, sb) = outs
    assert la == lb, (la, lb)
    for k in pa:
        assert np.array_equal(pa[k], pb[k]), k
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k


def test_flush_launches_what_no_launch_carried():
    """rcgan_proj_head_fwd_bwd with defer_ws, then rcgan_head_flush straight away (stage 1: both launches on their own) --
    the parameter gradients equal the undeferred call's."""
    import ctypes as C
    import rcgan_amd  # noqa: F401
    from rcgan_amd import _lib as L
    from rcgan_amd.runtime import Context
    ctx = Context(0, "bf16", arena_bytes=1 << 26, ws_bytes=1 << 24)
    try:
        rs = np.random.RandomState(5)
        n, d, v, e = 16, 128, 10, 128
        up = lambda a: ctx.upload(np.ascontiguousarray(a, np.float32), L.F32)
        feat, w_out, b_out = up(rs.randn(n, d)), up(rs.randn(d) * 0.1), up(rs.randn(1))
        table, w_e, b_e = up(rs.randn(v, e) * 0.3), up(rs.randn(e, d) * 0.1), up(rs.randn(d) * 0.1)
        labels = ctx.upload(rs.randint(v, size=n).astype(np.int32))
        res = []
        for defer in (False, True):
            outs = {k: ctx.upload(np.zeros(s, np.float32), L.F32) for k, s in
                    dict(loss=(1,), dfeat=(n, d), dw_out=(d,), db_out=(1,), dtable=(v, e), dw_e=(e, d), db_e=(d,)).items()}
            hd = L.HeadDesc(n, d, v, e, n // 2, L.LOSS_HINGE_REAL, L.LOSS_HINGE_FAKE, 1.0)
            hd.labels_a = labels.ptr
            hd.labels_b = labels.ptr + 4 * (n // 2)
            if defer:
                nbytes = (n * (v + 1) + (v + 1) * d) * 4
                hd.defer_ws, hd.defer_ws_bytes = ctx.arena.alloc(nbytes), nbytes
            p = lambda t: C.c_void_p(t.ptr)
            ctx.check(ctx.lib.rcgan_proj_head_fwd_bwd(ctx.h, C.byref(hd), p(feat), p(w_out), None, p(b_out), p(table), p(w_e), None, p(b_e),
                                                      p(outs["loss"]), None, p(outs["dfeat"]), p(outs["dw_out"]), p(outs["db_out"]),
                                                      p(outs["dtable"]), p(outs["dw_e"]), p(outs["db_e"]), C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
            if defer:
                ctx.sync()
                assert not np.any(ctx.download(outs["dw_e"])), "deferred: nothing written yet"
                ctx.check(ctx.lib.rcgan_head_flush(ctx.h))
                ctx.check(ctx.lib.rcgan_head_flush(ctx.h))       # idempotent
            res.append({k: ctx.download(t) for k, t in outs.items()})
        for k in res[0]:
            assert np.array_equal(res[0][k], res[1][k]), k
        assert np.any(res[0]["dw_e"]) and np.any(res[0]["dtable"])
        # a defer buffer that is too small is refused
        hd.defer_ws_bytes = 16
        rc = ctx.lib.rcgan_proj_head_fwd_bwd(ctx.h, C.byref(hd), p(feat), p(w_out), None, p(b_out), p(table), p(w_e), None, p(b_e),
                                             p(outs["loss"]), None, p(outs["dfeat"]), p(outs["dw_out"]), p(outs["db_out"]),
                                             p(outs["dtable"]), p(outs["dw_e"]), p(outs["db_e"]), C.c_void_p(ctx.ws_ptr), ctx.ws_bytes)
        assert rc == -3
    finally:
        ctx.close()


@pytest.mark.parametrize("alg,dtype", [("rcgan", "bf16"), ("rcgan-u", "bf16"), ("rcgan", "f16")])
def test_pooling_inside_the_stage_matches_pooling_inside_the_head(alg, dtype, monkeypatch):
    """RCGAN_POOL_IN_TRUNK (rcgan_dtrunk_pooled: features out of the 8x8 stage's launch, their gradient into its backward launch)
    against the head pooling by itself: the features differ by fp32 summation order, so the first critic step's gradients agree to
    1e-5 per tensor and two iterations stay on one trajectory (tests/test_gpu_dp._same_trajectory)."""
    import rcgan_amd  # noqa: F401
    from rcgan_amd import cifar as cm
    from tests.gpu_util import rel_err
    from tests.test_gpu_dp import _same_trajectory
    rs = np.random.RandomState(22)
    B = 8
    its = _feeds(rs, B, 2, alg)
    outs = []
    for pool in (False, True):
        monkeypatch.setattr(cm, "POOL_IN_TRUNK", pool)
        m = _model(alg, dtype, B)
        try:
            g1 = {}
            outs.append(_run_iterations(m, its, g1) + (g1,))
        finally:
            m.ctx.close()
    (pa, la, sa, ga), (pb, lb, sb, gb) = outs
    gmax = max(float(np.abs(v).max()) for v in ga.values())
    for k in ga:
        if float(np.abs(ga[k]).max()) > 1e-3 * gmax:
            assert rel_err(gb[k], ga[k]) <= 1e-5, (k, rel_err(gb[k], ga[k]))
    _same_trajectory(pa, pb)
