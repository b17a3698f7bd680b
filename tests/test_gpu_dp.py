"""The data-parallel step of the CIFAR engine (gan_resnet.py:529-546 tower split, :697,786 mean of the tower costs).

The product's world > 1 path: per optimiser step the gradient slab(s) all-reduced inside the C ABI (rcgan_allreduce_sum*) and inside the
step's captured graph (one whole-slab bucket per optimiser group, fp32 or bf16), and the optimiser launch in the same graph.  (The
overlapped two-bucket schedule of round 3 lost under the link model of `bench.py --dp-stub` and was deleted in round 4; the ABI keeps
rcgan_allreduce_sum_async / _join.)  This file checks it three ways:

  1. on ONE GPU against the in-ABI test-double communicator (rcgan_comm_init_stub: "every rank holds what this rank holds", so the
     all-reduce sum is world * x and the mean is x again): the world-size-N schedule must reproduce the world-size-1 run bit for bit;
  2. on ONE GPU with a real one-rank RCCL communicator (comm="rccl-self"): ncclAllReduce captured into the step graphs and replayed;
  3. with TWO processes on two GPUs over RCCL (skipped below 2 GPUs): ranks fed different shards end two iterations with
     bit-identical weights (the same summed gradients reached every rank) and different tower losses.
"""
import os
import sys

import numpy as np
import pytest

from tests.gpu_util import rel_err

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _feeds(rs, B, n_iter, alg):
    from rcgan_amd.cifar import N_CRITIC
    C = ((1 - 0.6) / 9.0) * np.ones((10, 10)) + (0.6 - (1 - 0.6) / 9.0) * np.eye(10)
    Cinv = np.linalg.inv(C)
    its = []
    for _ in range(n_iter):
        lra = rs.randint(10, size=N_CRITIC * B)
        ds = []
        for k in range(N_CRITIC):
            lab = rs.randint(10, size=B)
            lb = rs.randint(10, size=B)
            d = dict(images=rs.randint(0, 256, size=(B, 3072)), labels=lab, labels_random=lra[k * B:(k + 1) * B], labels_biased=lb,
                     inv_weights=Cinv[lab].astype(np.float32))
            d["labels_all"] = np.concatenate([lab, d["labels_random"] if alg in ("biased", "unbiased") else lb])
            ds.append(d)
        g = dict(labels_random_G=rs.randint(10, size=2 * B), labels_biased_G=rs.randint(10, size=2 * B))
        its.append((lra, ds, g))
    return its


def _run_iterations(m, its, grads_after_first=None, losses_after_first=None):
    """The production loop (bench.py / train_cifar.py): prepare_critic_fakes + N_CRITIC critic steps + a generator step."""
    for it, (lra, ds, g) in enumerate(its):
        m.set_feed("gf", m.pack_feed("gf", labels_random_all=lra))
        m.prepare_critic_fakes()
        for k, d in enumerate(ds):
            m.set_feed("d", m.pack_feed("d", **d))
            m.d_step(iteration=it)
            if grads_after_first is not None and it == 0 and k == 0:
                grads_after_first.update(m.get_grads(m.PD))
        m.set_feed("g", m.pack_feed("g", **g))
        m.g_step(iteration=it + 1)
        if losses_after_first is not None and it == 0:
            losses_after_first.append(m.losses())
    m.ctx.sync()
    return m.get_params(), m.losses(), m.get_state()


def _same_trajectory(pa, pb, lr=2e-4):
    """Two runs that differ only in fp32 summation order: Adam (beta1 = 0) moves every weight by ~lr per step whatever the size of
    its gradient, so a ~0 gradient whose sign flips leaves the two runs 2*lr apart in that element (zero-initialised biases consist of
    nothing but such steps).  Same trajectory = finite, no element further apart than the steps taken allow, and at most 20 % of a
    tensor's elements more than one step apart."""
    for k in pa:
        assert np.isfinite(pb[k]).all(), k
        if k.startswith("Generator/") and k.endswith("/Biases"):
            continue        # a per-channel constant in front of a batch norm: the true gradient is 0, every step is a coin flip
        d = np.abs(pb[k].astype(np.float64) - pa[k])
        assert float(d.max()) <= 2 * lr * 12 + 1e-6 and float(np.mean(d > lr)) <= 0.2, (k, float(d.max()), float(np.mean(d > lr)))


def _model(alg, dtype, B, **kw):
    import rcgan_amd  # noqa: F401
    from rcgan_amd.cifar import CifarRCGAN
    perm = alg == "rcgan-u"
    return CifarRCGAN(algorithm=alg, alpha=0.6, batch_size=B, dtype=dtype, seed=3, perm_classifier=perm, confuse_init=perm,
                      use_graphs=True, device_rng=True, arena_bytes=2 << 30, **kw)


@pytest.mark.parametrize("alg,dtype,world", [("rcgan", "bf16", 2), ("rcgan-u", "bf16", 8), ("rcgan", "f16", 4)])
def test_stub_world_is_bit_identical_to_single_rank(alg, dtype, world):
    """The world-size-N schedule: in-graph all-reduce of the whole slabs + in-graph
    optimiser (bf16) / dynamic-loss-scale optimiser after the graph (fp16).  sum = N * x and grad_scale = 1/N are exact in fp32:
    weights, losses and spectral-norm state after two iterations (capture + replays) equal the single-rank run bit for bit."""
    rs = np.random.RandomState(11)
    B = 8
    its = _feeds(rs, B, 2, alg)
    outs = []
    for w in (1, world):
        m = _model(alg, dtype, B, world_size=w, comm=("stub" if w > 1 else None))
        try:
            assert m.dp_active == (w > 1)
            assert m.dp_adam_in_graph == (w > 1 and dtype != "f16")
            outs.append(_run_iterations(m, its))
        finally:
            m.ctx.close()
    (pa, la, sa), (pb, lb, sb) = outs
    assert la == lb, (la, lb)
    for k in pa:
        assert np.array_equal(pa[k], pb[k]), k
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k


def test_communicator_that_cannot_be_captured_falls_back_to_uncaptured_steps(monkeypatch):
    """A communicator whose all-reduce fails inside a stream capture (the test double with RCGAN_COMM_STUB_FAIL_IN_CAPTURE set):
    the step drops the capture (rcgan_graph_abort), warns, and keeps running launch by launch -- same weights as the single-rank
    run, bit for bit; steps without a collective (the batched critic fakes) stay captured.
    (Round 4: this case failed once in ~40 executions -- NaN generator parameters.  Cause: ParamGroup.set_hyper_device created its {lr, t}
    buffer with torch.zeros on torch's current stream, unordered against the context's stream; when the fill lost the race the first
    captured-form Adam of the group read {0, 0}.  scripts/dbg_abort_flake.py repeats the scenario -- and the all-eager data-parallel run --
    hundreds of times in one process: 3 % of the runs before the fix, 0 of 1600 after.)"""
    rs = np.random.RandomState(14)
    B = 8
    its = _feeds(rs, B, 2, "rcgan")
    outs = []
    for w in (1, 2):
        if w > 1:
            monkeypatch.setenv("RCGAN_COMM_STUB_FAIL_IN_CAPTURE", "1")
        m = _model("rcgan", "bf16", B, world_size=w, comm=("stub" if w > 1 else None))
        try:
            if w > 1:
                with pytest.warns(UserWarning, match="could not be captured"):
                    outs.append(_run_iterations(m, its))
                kinds = {k: (g is None) for k, g in m._graphs.items()}
                assert any(kinds.values()) and not all(kinds.values()), kinds
            else:
                outs.append(_run_iterations(m, its))
        finally:
            m.ctx.close()
    (pa, la, sa), (pb, lb, sb) = outs
    assert la == lb, (la, lb)
    for k in pa:
        assert np.array_equal(pa[k], pb[k]), k


@pytest.mark.parametrize("bucket", ["f32", "bf16"])
def test_one_rank_rccl_communicator_in_captured_graphs(bucket):
    """A real RCCL communicator with one rank (comm="rccl-self"): ncclAllReduce (fp32 / bfloat16 buckets) on the step's stream,
    captured into the D / G step graphs and replayed; sum over one rank = identity, so the run equals the plain one.
    Skipped when librccl cannot be loaded."""
    import ctypes as C
    import rcgan_amd  # noqa: F401
    from rcgan_amd import _lib
    if _lib.load().rcgan_comm_unique_id(C.create_string_buffer(128)) != 0:
        pytest.skip("librccl.so not loadable")
    rs = np.random.RandomState(13)
    B = 8
    its = _feeds(rs, B, 2, "rcgan")
    outs = []
    for comm in (None, "rccl-self"):
        m = _model("rcgan", "bf16", B, world_size=1, comm=comm, grad_bucket_dtype=bucket)
        try:
            assert m.dp_active == (comm is not None)
            if comm is not None:
                import ctypes as C2
                n = C2.c_int(0)
                m.ctx.check(m.ctx.lib.rcgan_comm_count(m.ctx.h, C2.byref(n)))
                assert n.value == 1                                  # what ncclCommCount reports
            g1 = {}
            outs.append(_run_iterations(m, its, g1) + (g1,))
        finally:
            m.ctx.close()
    (pa, la, sa, ga), (pb, lb, sb, gb) = outs
    gmax = max(float(np.abs(v).max()) for v in ga.values())
    for k in ga:
        if float(np.abs(ga[k]).max()) > 1e-3 * gmax:
            assert rel_err(gb[k], ga[k]) <= (2e-5 if bucket == "f32" else 4e-3), (k, rel_err(gb[k], ga[k]))
    _same_trajectory(pa, pb)


@pytest.mark.parametrize("alg,dtype", [("rcgan", "bf16"), ("rcgan-u", "bf16")])
def test_stub_world_with_early_bucket_matches_single_rank(alg, dtype, monkeypatch):
    """The overlapped schedule (RCGAN_DP_OVERLAP=1; off by default, cifar.py -- back in round 6 so that an 8-GPU box can A/B it):
    D.Block.3 .. head (G.Block.2 .. G.Output) leave on the communication stream in the middle of the backward pass.  The early flush
    regroups the filter-gradient launches (other pixel splits, other fp32 summation order), so the first critic step's gradients agree
    to 2e-5 norm-relative per tensor; after two iterations (12 optimiser steps) the weights are on the same trajectory (Adam with
    beta1 = 0 turns a sign flip of a ~0 gradient under another summation order into a 2*lr difference)."""
    rs = np.random.RandomState(12)
    B = 8
    its = _feeds(rs, B, 2, alg)
    outs = []
    monkeypatch.setenv("RCGAN_DP_OVERLAP", "1")
    for w in (1, 2):
        m = _model(alg, dtype, B, world_size=w, comm=("stub" if w > 1 else None))
        try:
            assert m.dp_overlap == (w > 1)
            g1, l1 = {}, []
            outs.append(_run_iterations(m, its, g1, l1) + (g1, l1[0]))
        finally:
            m.ctx.close()
    (pa, _, sa, ga, la), (pb, _, sb, gb, lb) = outs
    gmax = max(float(np.abs(v).max()) for v in ga.values())
    for k in ga:
        if float(np.abs(ga[k]).max()) > 1e-3 * gmax:
            assert rel_err(gb[k], ga[k]) <= 2e-5, ("gradient of the first critic step", k, rel_err(gb[k], ga[k]))
    _same_trajectory(pa, pb)
    # losses after the FIRST iteration (six optimiser steps): at B = 8 the second iteration's losses of two bf16 runs that differ in
    # summation order already sit 0.1-0.9 apart, whichever kernels run
    assert abs(la[0] - lb[0]) <= 5e-3 * max(1.0, abs(la[0])) and abs(la[1] - lb[1]) <= 5e-3 * max(1.0, abs(la[1])), (la, lb)


def test_allreduce_abi_errors_and_buckets():
    """C ABI: all-reduce without a communicator -> RCGAN_EINVALID_ARG with a message; the test double scales every bucket of a
    group by the world size, the asynchronous bucket is complete after the join."""
    import ctypes as C
    import torch
    from rcgan_amd import _lib as L
    from tests.gpu_util import make_ctx
    ctx = make_ctx("bf16")
    try:
        a = torch.arange(1000, dtype=torch.float32, device=ctx.device)
        rc = ctx.lib.rcgan_allreduce_sum(ctx.h, C.c_void_p(a.data_ptr()), a.numel())
        assert rc == -1 and b"communicator" in ctx.lib.rcgan_last_error(ctx.h)
        ctx.check(ctx.lib.rcgan_comm_init_stub(ctx.h, 4))
        assert ctx.lib.rcgan_comm_world(ctx.h) == 4
        assert ctx.lib.rcgan_comm_init_stub(ctx.h, 2) == -1            # already initialised
        b = torch.ones(77, dtype=torch.float32, device=ctx.device)
        c = torch.full((5000,), 0.5, dtype=torch.float32, device=ctx.device)
        torch.cuda.synchronize()
        ctx.check(ctx.lib.rcgan_allreduce_sum_async(ctx.h, C.c_void_p(c.data_ptr()), c.numel()))
        ptrs, counts = (C.c_void_p * 2)(a.data_ptr(), b.data_ptr()), (C.c_size_t * 2)(a.numel(), b.numel())
        ctx.check(ctx.lib.rcgan_allreduce_sum_buckets(ctx.h, 2, ptrs, counts))
        ctx.check(ctx.lib.rcgan_allreduce_join(ctx.h))
        ctx.sync()
        assert torch.equal(a.cpu(), 4 * torch.arange(1000, dtype=torch.float32)) and float(b.sum()) == 4 * 77 and float(c.sum()) == 2.0 * 5000
        ctx.check(ctx.lib.rcgan_comm_destroy(ctx.h))
        assert ctx.lib.rcgan_comm_world(ctx.h) == 1
        assert L.ERRORS[-5] == "RCGAN_ERCCL"
    finally:
        ctx.close()


def test_bf16_buckets_rank_count_and_the_stub_link_model():
    """C ABI additions of round 4: rcgan_comm_count (what the COMMUNICATOR says its size is), rcgan_allreduce_sum_bf16_buckets (fp32 -> bf16
    nearest-even, summed, widened back; the test double: world * bf16(x), inf / nan preserved), rcgan_comm_stub_model (an all-reduce group
    then occupies its stream for latency + 2 (N-1)/N * bytes / bandwidth: measured with events on the stream)."""
    import ctypes as C
    import torch
    from tests.gpu_util import make_ctx
    ctx = make_ctx("bf16")
    try:
        n = C.c_int(0)
        assert ctx.lib.rcgan_comm_count(ctx.h, C.byref(n)) == -1                    # no communicator yet
        ctx.check(ctx.lib.rcgan_comm_init_stub(ctx.h, 8))
        ctx.check(ctx.lib.rcgan_comm_count(ctx.h, C.byref(n)))
        assert n.value == 8
        rs = np.random.RandomState(0)
        xa = (rs.randn(100003) * np.exp(rs.uniform(-30, 30, size=100003))).astype(np.float32)
        xa[:4] = [np.inf, -np.inf, np.nan, 0.0]
        xb = rs.randn(77).astype(np.float32)
        a, b = torch.from_numpy(xa).to(ctx.device), torch.from_numpy(xb).to(ctx.device)
        counts = (C.c_size_t * 2)(a.numel(), b.numel())
        need = ctx.lib.rcgan_allreduce_bf16_scratch_bytes(2, counts)
        assert need >= 2 * (a.numel() + b.numel()) and need % 256 == 0
        scratch = torch.empty(need, dtype=torch.uint8, device=ctx.device)
        ptrs = (C.c_void_p * 2)(a.data_ptr(), b.data_ptr())
        assert ctx.lib.rcgan_allreduce_sum_bf16_buckets(ctx.h, 2, ptrs, counts, C.c_void_p(scratch.data_ptr()), need - 256) == -3   # workspace
        ctx.check(ctx.lib.rcgan_allreduce_sum_bf16_buckets(ctx.h, 2, ptrs, counts, C.c_void_p(scratch.data_ptr()), need))
        ctx.sync()
        for got, x in ((a.cpu().numpy(), xa), (b.cpu().numpy(), xb)):
            want = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy() * np.float32(8)      # torch rounds to nearest even too
            assert np.array_equal(got, want, equal_nan=True)
        # link model: 100 us + 2 * 7/8 * bytes / 50 GB/s; 4 MB of fp32 -> 100 + 140 us
        ctx.check(ctx.lib.rcgan_comm_stub_model(ctx.h, 50.0, 100.0))
        c = torch.zeros(1 << 20, dtype=torch.float32, device=ctx.device)
        ctx.sync()
        reps = 5
        ctx.event_record(0)
        for _ in range(reps):
            ctx.check(ctx.lib.rcgan_allreduce_sum(ctx.h, C.c_void_p(c.data_ptr()), c.numel()))
        ctx.event_record(1)
        us = ctx.event_elapsed_ms(0, 1) * 1e3 / reps
        model = 100.0 + 2 * 7 / 8 * (4 << 20) / 50e3
        assert model <= us <= model + 60.0, (us, model)
        ctx.check(ctx.lib.rcgan_comm_stub_model(ctx.h, 0.0, 0.0))
        ctx.event_record(0)
        for _ in range(reps):
            ctx.check(ctx.lib.rcgan_allreduce_sum(ctx.h, C.c_void_p(c.data_ptr()), c.numel()))
        ctx.event_record(1)
        assert ctx.event_elapsed_ms(0, 1) * 1e3 / reps < 60.0
    finally:
        ctx.close()


def test_stub_world_with_bf16_buckets_tracks_single_rank():
    """grad_bucket_dtype="bf16": the slabs travel as bfloat16 (three launches + one group per step, captured).  Against the test double the
    only difference from the single-rank run is the rounding of every gradient to 8 mantissa bits in front of Adam: first-step
    gradients agree to 2^-8 per element (4e-3 norm-relative is generous), the weights follow the same trajectory."""
    rs = np.random.RandomState(15)
    B = 8
    its = _feeds(rs, B, 2, "rcgan")
    outs = []
    for w in (1, 4):
        m = _model("rcgan", "bf16", B, world_size=w, comm=("stub" if w > 1 else None), grad_bucket_dtype="bf16")
        try:
            g1 = {}
            outs.append(_run_iterations(m, its, g1) + (g1,))
            if w > 1:
                assert m._bucket16 is not None and all(g is not None for g in m._graphs.values())
        finally:
            m.ctx.close()
    (pa, la, sa, ga), (pb, lb, sb, gb) = outs
    gmax = max(float(np.abs(v).max()) for v in ga.values())
    for k in ga:
        if float(np.abs(ga[k]).max()) > 1e-3 * gmax:
            assert rel_err(gb[k], ga[k]) <= 4e-3, (k, rel_err(gb[k], ga[k]))
            assert not np.array_equal(ga[k], gb[k]) or ga[k].size < 4, k          # the rounding really happened
    _same_trajectory(pa, pb)


# ------------------------------------------------------------------------------------------------------------------
# two processes, two GPUs, RCCL
# ------------------------------------------------------------------------------------------------------------------
def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import rcgan_amd  # noqa: F401
    from rcgan_amd.cifar import CifarRCGAN
    rs = np.random.RandomState(100 + rank)          # every rank its own shard
    B = 8
    its = _feeds(rs, B, 2, "rcgan")
    m = CifarRCGAN(algorithm="rcgan", alpha=0.6, batch_size=B, dtype="bf16", seed=3, use_graphs=True, device_rng=True, device=rank,
                   world_size=world, rank=rank, arena_bytes=2 << 30)       # the unique id travels through the TCP store (no process group)
    try:
        p, losses, st = _run_iterations(m, its)
        q.put((rank, {k: v for k, v in p.items()}, losses))
    finally:
        m.ctx.close()


def test_two_ranks_over_rccl_end_with_identical_weights():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, 29611, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, params, losses = q.get(timeout=600)
        res[r] = (params, losses)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k in res[0][0]:
        assert np.array_equal(res[0][0][k], res[1][0][k]), "ranks diverged: " + k
    assert res[0][1] != res[1][1]          # different shards: different tower losses
