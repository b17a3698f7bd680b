"""Differentiable device ops: each forward is one (or a few) C-ABI calls into the gfx950 kernels and
records a closure that runs the matching hand-written backward kernels.  This tape takes the place of
``tf.gradients`` in the reference (``compute_gradients`` at cifar10/gan_resnet.py:803,807; ``minimize``
at mnist/model.py:250-262).  No arithmetic happens in Python or in PyTorch.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from .runtime import DT


def _p(t):
    return C.c_void_p(t.ptr) if t is not None else None


def same_pad(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return out, tot // 2


def zeros_like_grad(ctx, t):
    g = ctx.empty(t.shape, t.dtype)
    ctx.check(ctx.lib.rcgan_fill_f32(ctx.h, (g.nbytes + 3) // 4, g.ptr, 0.0))
    return g


def grad_of(ctx, t):
    """(gradient buffer of t, accumulate flag)."""
    if t.grad is None:
        t.grad = ctx.empty(t.shape, t.dtype)
        return t.grad, 0
    if getattr(t.grad, "frozen", False):
        # the buffer is still to be read by a deferred filter gradient (conv2d's residual adoption): copy on write
        fresh = ctx.empty(t.shape, t.dtype)
        ctx.check(ctx.lib.rcgan_axpby(ctx.h, t.size, t.dtype, 1.0, _p(t.grad), 0.0, _p(fresh)))
        t.grad = fresh
    return t.grad, 1


def _track(ctx, out, *ins):
    out.req = ctx.recording and any(i is not None and i.req for i in ins)
    return out.req


# ----------------------------------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------------------------------
class Weight:
    """A filter / matrix as the conv and dense ops consume it: the fp32 master parameter plus, when
    spectrally normalised, the device scalar sigma (W_bar = W / sigma is never materialised in fp32:
    the division is fused into the filter-preparation / operand-load)."""

    def __init__(self, ctx, param, sigma=None):
        self.ctx, self.param, self.sigma = ctx, param, sigma
        self._prepared = {}
        self.rf_frag = None        # fragment-major copy for the register-filter kernel (fragments_batch), this step's
        self.dwbar = None
        self._conv1 = None         # this [k, n] matrix as the [1, 1, k, n] filter of a 1x1 convolution (as_conv1x1)

    @property
    def req(self):
        return self.param.req

    @staticmethod
    def _key(desc, nbytes):
        # the prepared layout depends on whether the MFMA path takes this shape and on which resampling the layer folds in: the
        # summed sub-pixel filters of CONV_IN_UPSAMPLE2X and CONV_OUT_MEANPOOL2 have the same size and different contents
        # (only 3x3 layers carry them: mfma_phase_filters)
        resample = (L.CONV_IN_UPSAMPLE2X | L.CONV_OUT_MEANPOOL2) if (desc.kh == 3 and desc.kw == 3) else 0
        return (desc.kh, desc.kw, desc.cin, desc.cout, desc.dtype, desc.stride, desc.flags & (L.CONV_FORCE_DIRECT | resample), nbytes)

    def prepared(self, desc):
        ctx = self.ctx
        nbytes = ctx.lib.rcgan_conv_prepared_bytes(C.byref(desc))
        key = self._key(desc, nbytes)
        if key not in self._prepared:
            buf = DT(ctx.arena.alloc(nbytes), (nbytes,), "u8", ctx.arena.buf)
            ctx.check(ctx.lib.rcgan_conv_prepare(ctx.h, C.byref(desc), _p(self.param), _p(self.sigma), _p(buf)))
            self._prepared[key] = buf
        return self._prepared[key]

    def as_conv1x1(self):
        """The same parameter memory viewed as an HWIO filter [1, 1, k, n] (ops.linear on the matrix cores): shares the gradient
        buffer and the prepared-layout cache."""
        if self._conv1 is None:
            p = self.param
            pv = DT(p.ptr, (1, 1) + tuple(p.shape), p.dtype, p.base, p.name)
            pv.req, pv.group = p.req, p.group
            if p.grad is not None:
                pv.grad = p.grad.reshape((1, 1) + tuple(p.shape))
            w4 = Weight(self.ctx, pv, self.sigma)
            w4._prepared = self._prepared
            self._conv1 = w4
        return self._conv1

    def grad_target(self):
        """Where d/dW (or d/dW_bar for SN weights) is accumulated."""
        if self.sigma is None:
            return self.param.grad
        if self.dwbar is None:
            self.dwbar = DT(self.ctx.arena.alloc(self.param.nbytes), self.param.shape, L.F32, self.ctx.arena.buf)
            self.ctx.check(self.ctx.lib.rcgan_fill_f32(self.ctx.h, self.param.size, self.dwbar.ptr, 0.0))
        return self.dwbar


def prepare_batch(ctx, weights_and_shapes, dtype, persistent=None, embed=None, inputs=None, frags=None):
    """Prepare the filters of many convs in one launch.  weights_and_shapes: list of (Weight, k, stride, hw):
    hw = spatial size of the conv input (the image-end layouts depend on it), 8 if irrelevant.
    persistent: None -> per-step arena buffers, skipped when already prepared this step;
                dict {param name: {key: DT}} -> (re)fill buffers that survive the step (filters that only change
                with their optimiser step).
    frags: [(Weight, fwd ptr, bwd ptr, ctn, ss)] -- fragment-major copies (the fused 8x8 stage's, the register-filter kernel's) the
           SAME launch writes for filters it prepares (rcgan_conv_prepare_batch_frags); returns (rode, set of Weights whose copies were written)
           when frags is given."""
    todo = []
    index = {}
    for w, k, stride, hw, *rest in weights_and_shapes:
        cin, cout = w.param.shape[-2:]          # HWIO filters, or a [k, n] matrix prepared as a 1x1 filter (ops.linear)
        # flags that change the prepared layout (CONV_IN_UPSAMPLE2X: the summed phase filters of the sub-pixel form ride along)
        desc = L.ConvDesc(1, hw, hw, cin, cout, k, k, stride, dtype, rest[0] if rest else 0)
        nbytes = ctx.lib.rcgan_conv_prepared_bytes(C.byref(desc))
        key = Weight._key(desc, nbytes)
        if persistent is None:
            if key in w._prepared:
                continue
            buf = DT(ctx.arena.alloc(nbytes), (nbytes,), "u8", ctx.arena.buf)
        else:
            slot = persistent.setdefault(w.param.name, {})
            if key not in slot:
                t = torch.empty(int(nbytes), dtype=torch.uint8, device=ctx.device)
                slot[key] = DT(t.data_ptr(), (nbytes,), "u8", t)
            buf = slot[key]
        w._prepared[key] = buf
        index[id(w)] = len(todo)
        todo.append(L.PrepareItem(desc, w.param.ptr, w.sigma.ptr if w.sigma is not None else None, buf.ptr))
    fitems = [(index[id(w)], f, b, ctn, ss, w) for (w, f, b, ctn, ss) in (frags or []) if id(w) in index]
    if todo:
        arr = (L.PrepareItem * len(todo))(*todo)
        if fitems and len(fitems) <= 12:
            ed = si = None
            if embed is not None:
                table, w_e, b_e, E = embed
                v, e_dim = table.shape
                ed = L.EmbedDesc(v, e_dim, w_e.param.shape[-1], table.ptr, w_e.param.ptr, w_e.sigma.ptr if w_e.sigma is not None else None,
                                 b_e.ptr if b_e is not None else None, E.ptr)
            if inputs is not None:
                si = inputs
            fa = (L.FragItem * len(fitems))(*[L.FragItem(i, ctn, ss, f, b) for i, f, b, ctn, ss, _ in fitems])
            ctx.check(ctx.lib.rcgan_conv_prepare_batch_frags(ctx.h, arr, len(todo), C.byref(ed) if ed is not None else None,
                                                             C.byref(si) if si is not None else None, fa, len(fitems)))
            return (embed is not None or inputs is not None), {id(w) for *_, w in fitems}
        if embed is not None or inputs is not None:
            ed = si = None
            if embed is not None:
                # the projection head's label embeddings ride in the same launch: embed = (table DT, W_e Weight, b_e DT or None, E DT)
                table, w_e, b_e, E = embed
                v, e_dim = table.shape
                ed = L.EmbedDesc(v, e_dim, w_e.param.shape[-1], table.ptr, w_e.param.ptr, w_e.sigma.ptr if w_e.sigma is not None else None,
                                 b_e.ptr if b_e is not None else None, E.ptr)
            if inputs is not None:
                # ... and the critic step's input work (rcgan_step_inputs_desc): a StepInputsDesc built by the caller
                si = inputs
            ctx.check(ctx.lib.rcgan_conv_prepare_batch_riders(ctx.h, arr, len(todo), C.byref(ed) if ed is not None else None,
                                                              C.byref(si) if si is not None else None))
            return True if frags is None else (True, set())
        ctx.check(ctx.lib.rcgan_conv_prepare_batch(ctx.h, arr, len(todo)))
    return False if frags is None else (False, set())


RF_CONV = os.environ.get("RCGAN_RF_CONV", "1") != "0"
# the data gradient of a layer fed by conv_cond_concat(x, labels) computed for the x channels only, straight into x's gradient
CONCAT_DIRECT = os.environ.get("RCGAN_CONCAT_DIRECT", "1") != "0"
CONCAT_WGRAD = os.environ.get("RCGAN_CONCAT_WGRAD", "1") != "0"     # (round 6) ops.deconv2d: label columns of the filter gradient from per-sample sums
LINEAR_MFMA = os.environ.get("RCGAN_LINEAR_MFMA", "1") != "0"


# (round 6) MEASURED AND LEFT OFF: the fragment-major copies written by the filter-preparation launch itself (fragment rows of
# conv_prepare_batch_kernel, rcgan_conv_prepare_batch_frags; bit-identical, tests/test_gpu_ops.py) instead of by a launch of their own behind
# it.  The launch it removes is 5.3 us on every step's chain, but the fragment layout scatters a tile's bytes into 16-byte pieces and the
# preparation launch grows by more than that: same box 5.181 / 5.174 / 5.181 ms per iteration with the separate launch, 5.209 / 5.196 /
# 5.208 with the rows.  RCGAN_FRAG_IN_PREPARE=1 turns them on.
FRAG_IN_PREPARE = os.environ.get("RCGAN_FRAG_IN_PREPARE", "0") == "1"


def fragment_requests(ctx, trunk, rf):
    """(round 6) The fragment-major copies of a step as riders of the filter-preparation launch (prepare_batch(frags=...)): allocates the
    destinations and returns (trunk fragment buffer or None, [(Weight, fwd ptr, bwd ptr, ctn, ss)]); the rf Weights get .rf_frag."""
    reqs, tf = [], None
    if trunk:
        nb = ctx.lib.rcgan_dtrunk_fragment_bytes()
        tf = DT(ctx.arena.alloc(nb), (nb,), "u8", ctx.arena.buf)
        elems2 = 9 * 128 * 128 * 2                     # bytes of one layer's fragments in one direction
        for i, w in enumerate(trunk):                  # forward: layers first to last; backward: last to first (conv_trunk.hip)
            reqs.append((w, tf.ptr + i * elems2, tf.ptr + (8 + (7 - i)) * elems2, 2, 36))
    for w, d in rf:
        nb = ctx.lib.rcgan_conv_rf_fragment_bytes(C.byref(d))
        w.rf_frag = DT(ctx.arena.alloc(nb), (nb,), "u8", ctx.arena.buf)
        reqs.append((w, w.rf_frag.ptr, w.rf_frag.ptr + nb // 2, 4, 18))
    return tf, reqs


def fragments_batch(ctx, trunk, rf):
    """ONE launch writes every fragment-major filter copy of a step (rcgan_fragments_prepare): trunk = the 8x8 stage's eight Weights in
    forward order (or None) -> returns its fragment buffer (ops.d_trunk(frag=...)); rf = [(Weight, desc)] of the layers the
    register-filter kernel takes -> each Weight gets .rf_frag.  All filters must be prepared already (prepare_batch)."""
    tp = tf = None
    if trunk:
        tdesc = L.ConvDesc(1, 8, 8, 128, 128, 3, 3, 1, ctx.act_dtype, L.CONV_IN_RELU)
        tp = (C.c_void_p * 8)(*[w.prepared(tdesc).ptr for w in trunk])
        nb = ctx.lib.rcgan_dtrunk_fragment_bytes()
        tf = DT(ctx.arena.alloc(nb), (nb,), "u8", ctx.arena.buf)
    n = len(rf)
    descs = (L.ConvDesc * max(n, 1))(*[d for _, d in rf])
    preps = (C.c_void_p * max(n, 1))(*[w.prepared(d).ptr for w, d in rf])
    frags = []
    for w, d in rf:
        nb = ctx.lib.rcgan_conv_rf_fragment_bytes(C.byref(d))
        w.rf_frag = DT(ctx.arena.alloc(nb), (nb,), "u8", ctx.arena.buf)
        frags.append(w.rf_frag.ptr)
    ctx.check(ctx.lib.rcgan_fragments_prepare(ctx.h, tp, _p(tf), n, descs, preps, (C.c_void_p * max(n, 1))(*frags)))
    return tf


def spectral_norm_batch(ctx, entries):
    """entries: list of (param DT [.., c], u DT [c] persistent, update flag).  One launch runs the power
    iteration of every weight (mnist/sn.py:37-62); returns a Weight per entry.  Records ONE tape entry
    that, at the very end of the backward pass, turns every accumulated dW_bar into dW -- through the
    power iteration, as TF does (no stop_gradient in the reference)."""
    n = len(entries)
    if n == 0:
        return []
    items = (L.SnItem * n)()
    weights = []
    saves = []
    for i, (param, u, update) in enumerate(entries):
        c = param.shape[-1]
        k = param.size // c
        sigma = ctx.empty((1,), L.F32)
        save = ctx.empty((ctx.lib.rcgan_sn_save_floats(k, c),), L.F32)
        items[i] = L.SnItem(param.ptr, u.ptr, sigma.ptr, save.ptr, k, c, 1 if update else 0)
        weights.append(Weight(ctx, param, sigma))
        saves.append((save, k, c))
    ctx.check(ctx.lib.rcgan_sn_power_iter(ctx.h, items, n))
    # gradient scratch d/dW_bar of every trainable weight: one contiguous block, zeroed by ONE fill
    if ctx.recording:
        req = [w for w in weights if w.param.req]
        total = sum((w.param.size + 63) // 64 * 64 for w in req)
        if total and all(w.param.group is not None and w.param.group.sn_scratch and w.param.group.zero_epoch == ctx.epoch for w in req):
            # the group's zero_grad() of this step already cleared its d/dW_bar slab (runtime.ParamGroup): no fill
            for w in req:
                w.dwbar = w.param.group.dwbar(w.param.name)
        elif total:
            base = ctx.arena.alloc(total * 4)
            ctx.check(ctx.lib.rcgan_fill_f32(ctx.h, total, base, 0.0))
            off = 0
            for w in req:
                w.dwbar = DT(base + off * 4, w.param.shape, L.F32, ctx.arena.buf)
                off += (w.param.size + 63) // 64 * 64

    done = set()

    def bw(only=None):
        """dW_bar -> dW through the power iteration for every weight not handled yet (only: predicate on the parameter name -- the
        overlapped data-parallel schedule finishes the gradients of its first bucket early, Context.sn_partial)."""
        ctx.flush_wgrads()          # the deferred filter gradients write the dW_bar this closure consumes
        todo = [(i, w, s) for i, (w, s) in enumerate(zip(weights, saves))
                if i not in done and w.dwbar is not None and w.param.req and (only is None or only(w.param.name))]
        if not todo:
            return
        bi = (L.SnBwdItem * len(todo))()
        for j, (i, w, (save, k, c)) in enumerate(todo):
            bi[j] = L.SnBwdItem(w.param.ptr, w.dwbar.ptr, w.param.grad.ptr, save.ptr, k, c, 1)
            done.add(i)
        fa = getattr(ctx, "sn_adam", None)
        if fa is not None and not fa.get("done") and len(todo) == len(weights) and all(w.param.group is fa["group"] for _, w, _ in todo):
            # (round 6) a single-rank step whose optimiser group is exactly this call's weights + the parameters between them: the
            # second launch also applies TF-Adam -- to the rows it has just produced, and through rider workgroups to the rest of
            # the slab (rcgan_sn_bwd_adam).  This closure is the LAST gradient work of the step (recorded first, it runs last), so
            # every other gradient of the group is final here.
            grp = fa["group"]
            cover = sorted((grp.offsets[w.param.name], grp.offsets[w.param.name] + w.param.size) for _, w, _ in todo)
            ranges, at = [], 0
            for lo, hi in cover:
                if lo > at:
                    ranges += [at, lo]
                at = hi
            if at < grp.count:
                ranges += [at, grp.count]
            ra = (C.c_size_t * max(len(ranges), 1))(*ranges)
            opt = L.SnAdam(grp.value.data_ptr(), grp.grad.data_ptr(), grp.m.data_ptr(), grp.v.data_ptr(), grp.count, grp.hyper.data_ptr(),
                           fa["beta1"], fa["beta2"], fa.get("eps", 1e-8), fa.get("clip", 0.0), fa["grad_scale"], len(ranges) // 2, ra)
            ctx.check(ctx.lib.rcgan_sn_bwd_adam(ctx.h, bi, len(todo), C.byref(opt)))
            fa["done"] = True
            return
        ctx.check(ctx.lib.rcgan_sn_bwd(ctx.h, bi, len(todo)))
    if any(p.req for p, _, _ in entries):
        ctx.record(bw)
        ctx.sn_partial.append(bw)
    return weights


# ----------------------------------------------------------------------------------------------------
# convolution / dense
# ----------------------------------------------------------------------------------------------------
# Batch-norm statistics out of the producing convolution's epilogue (rcgan_conv2d_fwd_stats; G.Block.3's two convolutions on the
# 256 x 256 kernel).  Parity-tested (test_conv_tile_statistics_equal_a_statistics_pass) and measured in round 3 on one MI355X, same
# box A/B: 6.349 ms per iteration without, 6.354-6.362 with -- the four statistics passes it removes (98 us) come back as the
# finisher launch (4 x 9.6 us), a colder apply pass (+23 us: the statistics pass had pulled the tensor into the Infinity Cache) and
# the epilogue's extra registers (the kernel sits at its 256-register budget: 16 spills).  Off unless RCGAN_FUSE_BN_STATS=1.
FUSE_BN_STATS = os.environ.get("RCGAN_FUSE_BN_STATS", "0") == "1"


def conv2d(ctx, x, weight, bias, k, stride=1, in_up=False, in_relu=False, accumulate_into=None, force_direct=False,
           residual=None, residual_up=False, want_stats=False):
    """SAME conv on NHWC x with HWIO weight [k,k,cin,cout] (tf.nn.conv2d + bias_add: mnist/ops.py:62-65,
    cifar10/common/ops/conv2d.py:181-216).  in_up / in_relu fold the preceding nearest-2x upsample
    (gan_resnet.py:263-264) and ReLU into the operand load; accumulate_into adds the result into an
    existing tensor (the residual sum of gan_resnet.py:328); residual adds another tensor in the epilogue
    (y = conv + residual: the identity-shortcut blocks, where the shortcut is the block input itself).  residual_up: the
    residual lives on the half-resolution grid and is added nearest-upsampled (the up blocks' 1x1 shortcut evaluated before
    the upsample it commutes with); its gradient is the 2x2 sum of dy.
    want_stats: the caller will batch-normalise the result -- where the producing kernel can, the per-tile column sums of the stored
    output come out of its epilogue (rcgan_conv2d_fwd_stats) and are attached as ``y.tile_stats`` for batch_norm_act."""
    if isinstance(x, BnPending):
        pend = x
        n, hs, ws_, cin = pend.shape
        cout = weight.param.shape[-1]
        assert weight.param.shape == (k, k, cin, cout), (weight.param.shape, (k, k, cin, cout))
        # the convolution as the plain path below would pose it (upsample folded into the load, half-resolution residual in the epilogue)
        bh, bw = (hs * 2, ws_ * 2) if in_up else (hs, ws_)
        bflags = (L.CONV_IN_UPSAMPLE2X if in_up else 0) | (L.CONV_RESID_UPSAMPLE2X if (residual is not None and residual_up) else 0)
        bdesc = L.ConvDesc(n, bh, bw, cin, cout, k, k, stride, pend.dtype, bflags)
        if (not in_relu and accumulate_into is None and not force_direct and not ctx.recording and stride == 1
                and pend.act in (L.ACT_NONE, L.ACT_RELU) and ctx.lib.rcgan_conv_bn_in_ok(C.byref(bdesc))
                and (residual is None or (residual.shape == ((n, bh // 2, bw // 2, cout) if residual_up else (n, bh, bw, cout))
                                          and (not residual_up or ctx.lib.rcgan_conv_resid_up_ok(C.byref(bdesc)))))):
            y = ctx.empty((n, bh, bw, cout), pend.dtype)
            pdesc = L.ConvDesc(n, bh, bw, cin, cout, k, k, stride, pend.dtype, L.CONV_IN_UPSAMPLE2X if in_up else 0)
            rc = ctx.lib.rcgan_conv2d_fwd_bn_residual(ctx.h, C.byref(bdesc), _p(pend.x), _p(weight.prepared(pdesc)), _p(bias), _p(residual), _p(y),
                                                      pend.segments, _p(pend.labels), _p(pend.gamma), _p(pend.beta), _p(pend.mean),
                                                      _p(pend.rstd), pend.act)
            if rc == 0:
                return y
            if rc != L.EUNSUPPORTED_SHAPE:      # (the library declined the fused form after all: write the normalised tensor and go on)
                ctx.check(rc)
        x = pend.materialize()
    n, hs, ws_, cin = x.shape
    h, w = (hs * 2, ws_ * 2) if in_up else (hs, ws_)
    cout = weight.param.shape[-1]
    assert weight.param.shape == (k, k, cin, cout), (weight.param.shape, (k, k, cin, cout))
    flags = (L.CONV_IN_UPSAMPLE2X if in_up else 0) | (L.CONV_IN_RELU if in_relu else 0) | (L.CONV_FORCE_DIRECT if force_direct else 0)
    oh, _ = same_pad(h, k, stride)
    ow, _ = same_pad(w, k, stride)
    desc = L.ConvDesc(n, h, w, cin, cout, k, k, stride, x.dtype, flags)
    prep = weight.prepared(desc)
    if accumulate_into is not None:
        y = accumulate_into
        assert y.shape == (n, oh, ow, cout)
        fdesc = L.ConvDesc(n, h, w, cin, cout, k, k, stride, x.dtype, flags | L.CONV_ACCUMULATE)
    else:
        y = ctx.empty((n, oh, ow, cout), x.dtype)
        fdesc = desc
    # the register-filter kernel (conv_rf.hip) for the small-grid 3x3 layers it takes, when this step's fragment-major copy exists
    use_rf = (RF_CONV and weight.rf_frag is not None and residual is None and not in_up and not force_direct
              and bool(ctx.lib.rcgan_conv_rf_ok(C.byref(fdesc))))
    sdesc = None
    if want_stats and FUSE_BN_STATS and accumulate_into is None and not in_relu and x.dtype != L.F32:
        sd = L.ConvDesc(n, h, w, cin, cout, k, k, stride, x.dtype, flags | (L.CONV_RESID_UPSAMPLE2X if (residual is not None and residual_up) else 0))
        if ctx.lib.rcgan_conv_stats_ok(C.byref(sd)):
            sdesc = sd
    if sdesc is not None:
        assert residual is None or residual.shape == ((n, oh // 2, ow // 2, cout) if residual_up else (n, oh, ow, cout))
        sums = ctx.empty((ctx.lib.rcgan_conv_stats_bytes(C.byref(sdesc)) // 4,), L.F32)
        ctx.check(ctx.lib.rcgan_conv2d_fwd_stats(ctx.h, C.byref(sdesc), _p(x), _p(prep), _p(bias), _p(residual), _p(y), _p(sums)))
        y.tile_stats = (sdesc, sums)
    elif residual is not None and residual_up:
        assert residual.shape == (n, oh // 2, ow // 2, cout) and accumulate_into is None, (residual.shape, (n, oh, ow, cout))
        rdesc = L.ConvDesc(n, h, w, cin, cout, k, k, stride, x.dtype, flags | L.CONV_RESID_UPSAMPLE2X)
        if ctx.lib.rcgan_conv_resid_up_ok(C.byref(rdesc)):
            ctx.check(ctx.lib.rcgan_conv2d_fwd_residual(ctx.h, C.byref(rdesc), _p(x), _p(prep), _p(bias), _p(residual), _p(y)))
        else:      # fp32 / odd shapes: the un-fused composition
            up = ctx.empty((n, oh, ow, cout), x.dtype)
            ctx.check(ctx.lib.rcgan_upsample2_fwd(ctx.h, n, oh, ow, cout, x.dtype, _p(residual), _p(up)))
            ctx.check(ctx.lib.rcgan_conv2d_fwd_residual(ctx.h, C.byref(fdesc), _p(x), _p(prep), _p(bias), _p(up), _p(y)))
    elif residual is not None:
        assert residual.shape == (n, oh, ow, cout) and accumulate_into is None, (residual.shape, (n, oh, ow, cout))
        ctx.check(ctx.lib.rcgan_conv2d_fwd_residual(ctx.h, C.byref(fdesc), _p(x), _p(prep), _p(bias), _p(residual), _p(y)))
    elif use_rf:
        ctx.check(ctx.lib.rcgan_conv2d_rf(ctx.h, C.byref(fdesc), 0, _p(x), _p(weight.rf_frag), _p(bias), None, None, _p(y)))
    else:
        ctx.check(ctx.lib.rcgan_conv2d_fwd(ctx.h, C.byref(fdesc), _p(x), _p(prep), _p(bias), _p(y)))
    prev_req = y.req if accumulate_into is not None else False
    if _track(ctx, y, x, weight.param, bias, residual) or prev_req:
        y.req = True
        xr, wr, br = x.req, weight.req, (bias is not None and bias.req)

        def bw():
            dy = y.grad
            if dy is None:
                return
            if br and not wr:
                raise NotImplementedError("bias-only gradient")
            # filter gradient and data gradient only share their inputs: run them side by side on two streams (most
            # layers of these nets leave the chip half empty); the filter gradient gets its own workspace
            fork = bool(xr and wr and ctx.overlap)
            defer = bool(wr and ctx.group_wgrads and not fork and x.dtype != L.F32)
            if wr and defer:
                # computed with the other layers' filter gradients in one grouped launch (Context.flush_wgrads: before the
                # spectral-norm backward reads dW_bar, at the latest at the end of the backward pass).  x is a forward
                # activation and dy is final here; the one place that would write into dy later is the residual adoption
                # below, which copies instead when this layer's gradient is deferred
                ctx.defer_wgrad(desc, x, dy, weight.grad_target(), bias.grad if br else None)
            elif wr:
                dw = weight.grad_target()
                if fork:
                    ctx.check(ctx.lib.rcgan_side_begin(ctx.h))
                ctx.check(ctx.lib.rcgan_conv2d_bwd_weight(ctx.h, C.byref(desc), _p(x), _p(dy), _p(dw),
                                                          _p(bias.grad) if br else None, 1,
                                                          C.c_void_p(ctx.ws2_ptr if fork else ctx.ws_ptr), ctx.ws_bytes))
                if fork:
                    ctx.check(ctx.lib.rcgan_side_end(ctx.h))
            if xr and x.grad is not None and getattr(x.grad, "frozen", False):
                # x.grad is a buffer a deferred filter gradient still has to read (see the residual adoption below): the
                # second contribution is added out of place, in the data-gradient kernel's epilogue
                dx = ctx.empty(x.shape, x.dtype)
                ctx.check(ctx.lib.rcgan_conv2d_bwd_data_residual(ctx.h, C.byref(desc), _p(dy), _p(prep), _p(x) if in_relu else None,
                                                                 _p(x.grad), _p(dx), C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
                x.grad = dx
            elif xr:
                dx, acc = grad_of(ctx, x)
                d2 = L.ConvDesc(n, h, w, cin, cout, k, k, stride, x.dtype, flags | (L.CONV_ACCUMULATE if acc else 0))
                if use_rf and ctx.lib.rcgan_conv_rf_ok(C.byref(d2)):
                    ctx.check(ctx.lib.rcgan_conv2d_rf(ctx.h, C.byref(d2), 1, _p(dy), _p(weight.rf_frag), None, _p(x) if in_relu else None, None, _p(dx)))
                else:
                    ctx.check(ctx.lib.rcgan_conv2d_bwd_data(ctx.h, C.byref(d2), _p(dy), _p(prep), _p(x) if in_relu else None,
                                                            _p(dx), C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
            if fork:
                ctx.check(ctx.lib.rcgan_side_join(ctx.h))
            if residual is not None and residual.req and residual_up:
                # d(residual) = the 2x2 sum of dy (adjoint of the nearest upsample)
                dres, racc = grad_of(ctx, residual)
                ctx.check(ctx.lib.rcgan_upsample2_bwd(ctx.h, n, oh, ow, cout, x.dtype, _p(dy), _p(dres), racc))
            elif residual is not None and residual.req:
                # d(residual) = dy.  dy is dead after this closure: a residual without a gradient yet adopts the
                # buffer (later contributions accumulate into it in place), otherwise one accumulate
                if residual.grad is None:
                    residual.grad = dy
                    if defer:
                        dy.frozen = True        # read again by flush_wgrads: later contributions go out of place (above)
                else:
                    ctx.check(ctx.lib.rcgan_axpby(ctx.h, residual.size, residual.dtype, 1.0, _p(dy), 1.0, _p(residual.grad)))
        ctx.record(bw)
    return y


def conv_meanpool_ok(ctx, x, weight, k=3):
    """conv2d_meanpool takes this layer (16-bit activations, 3x3, power-of-two image, channels % 64 == 0, whole tiles)."""
    if x.dtype == L.F32 or k != 3:
        return False
    n, h, w, cin = x.shape
    desc = L.ConvDesc(n, h, w, cin, weight.param.shape[-1], 3, 3, 1, x.dtype, L.CONV_OUT_MEANPOOL2)
    return bool(ctx.lib.rcgan_conv_fused_pool_ok(C.byref(desc)))


def conv2d_meanpool(ctx, x, weight, bias, in_relu=False, accumulate_into=None):
    """ConvMeanPool (gan_resnet.py:241-247): meanpool2(conv3x3(x) + bias) as ONE 4x4 stride-2 convolution with summed filters
    (RCGAN_CONV_OUT_MEANPOOL2: 4/9 of the multiply-adds, no full-resolution conv output, no pooling pass).  accumulate_into: a
    pooled-resolution tensor the result is added to (the block's shortcut).  Backward: data and filter gradient in their
    sub-pixel forms straight from the pooled dy (the filter gradient in the grouped launch; where the three-tap kernel does
    not take the shape, through the ordinary path on dy spread back to the conv resolution)."""
    n, h, w, cin = x.shape
    cout = weight.param.shape[-1]
    assert weight.param.shape == (3, 3, cin, cout)
    flags = L.CONV_OUT_MEANPOOL2 | (L.CONV_IN_RELU if in_relu else 0)
    desc = L.ConvDesc(n, h, w, cin, cout, 3, 3, 1, x.dtype, flags)
    prep = weight.prepared(desc)
    if accumulate_into is not None:
        y = accumulate_into
        assert y.shape == (n, h // 2, w // 2, cout)
        fdesc = L.ConvDesc(n, h, w, cin, cout, 3, 3, 1, x.dtype, flags | L.CONV_ACCUMULATE)
    else:
        y = ctx.empty((n, h // 2, w // 2, cout), x.dtype)
        fdesc = desc
    ctx.check(ctx.lib.rcgan_conv2d_fwd(ctx.h, C.byref(fdesc), _p(x), _p(prep), _p(bias), _p(y)))
    prev_req = y.req if accumulate_into is not None else False
    if _track(ctx, y, x, weight.param, bias) or prev_req:
        y.req = True
        xr, wr, br = x.req, weight.req, (bias is not None and bias.req)

        def bw():
            dy = y.grad
            if dy is None:
                return
            if wr:
                wdesc = L.ConvDesc(n, h, w, cin, cout, 3, 3, 1, x.dtype, flags)
                if ctx.lib.rcgan_conv_wgrad_pool_ok(C.byref(wdesc)):
                    # the sub-pixel filter gradient straight from the pooled dy (16 products over the pooled grid instead of 9
                    # over the full one; dy is final here and nothing below writes into it)
                    dyf = dy
                else:
                    # dL/d(conv output) = dy spread over each 2x2 block / 4: the filter (and bias) gradient of the plain convolution
                    dyf = ctx.empty((n, h, w, cout), x.dtype)
                    ctx.check(ctx.lib.rcgan_meanpool2_bwd(ctx.h, n, h, w, cout, x.dtype, _p(dy), _p(dyf), 0))
                    wdesc = L.ConvDesc(n, h, w, cin, cout, 3, 3, 1, x.dtype, L.CONV_IN_RELU if in_relu else 0)
                if ctx.group_wgrads:
                    ctx.defer_wgrad(wdesc, x, dyf, weight.grad_target(), bias.grad if br else None)
                else:
                    ctx.check(ctx.lib.rcgan_conv2d_bwd_weight(ctx.h, C.byref(wdesc), _p(x), _p(dyf), _p(weight.grad_target()),
                                                              _p(bias.grad) if br else None, 1, C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
            if xr:
                dx, acc = grad_of(ctx, x)
                d2 = L.ConvDesc(n, h, w, cin, cout, 3, 3, 1, x.dtype, flags | (L.CONV_ACCUMULATE if acc else 0))
                ctx.check(ctx.lib.rcgan_conv2d_bwd_data(ctx.h, C.byref(d2), _p(dy), _p(prep), _p(x) if in_relu else None,
                                                        _p(dx), C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
        ctx.record(bw)
    return y


def d_trunk_ok(ctx, x):
    """The fused 8x8 discriminator stage takes 16-bit [n, 8, 8, 128] activations."""
    return x.dtype != L.F32 and tuple(x.shape[1:]) == (8, 8, 128)


def d_trunk(ctx, x, blocks, pool=None, frag=None):
    """D.Block.3 .. D.Block.6 of the CIFAR discriminator -- four identity-shortcut residual blocks
    x' = x + Conv2(relu(Conv1(relu(x)))) (gan_resnet.py:275-328 with resample=None, :398-404) -- as ONE launch each way
    (rcgan_dtrunk): a workgroup carries one image through all eight 3x3 convolutions with the activations in LDS.
    blocks: [(Weight conv1, bias1, Weight conv2, bias2)] x 4.  Same values as eight conv2d calls up to the summation order
    inside a convolution (the rounding points -- 16-bit activations between layers, fp32 residual adds -- are the same).
    pool=ACT_RELU: the launch also leaves mean_hw(relu(y)) -- the discriminator's features, gan_resnet.py:405-407 -- in y.pooled, and
    a consumer that takes them (act_meanhw_later -> proj_head) hands the features' gradient back through y.pool_grad: the backward
    launch forms the incoming gradient itself, nobody reads or writes the [n, 8, 8, 128] activations around the head."""
    n = x.shape[0]
    assert d_trunk_ok(ctx, x) and len(blocks) == 4 and pool in (None, L.ACT_RELU)
    desc = L.ConvDesc(n, 8, 8, 128, 128, 3, 3, 1, x.dtype, L.CONV_IN_RELU)
    arr = lambda ts: (C.c_void_p * 8)(*[(t.ptr if t is not None else None) for t in ts])
    flat = []                                                  # (Weight, bias) of the eight layers in execution order
    for w1, b1, w2, b2 in blocks:
        flat += [(w1, b1), (w2, b2)]
    preps = [w.prepared(desc) for w, _ in flat]
    outs = [ctx.empty(x.shape, x.dtype) for _ in range(8)]
    # the stage's filters re-laid fragment-major, both directions, once per set of weights (the arena keeps it until the backward pass)
    if frag is None:           # (the CIFAR step writes it with the other fragment copies: fragments_batch)
        frag = DT(ctx.arena.alloc(ctx.lib.rcgan_dtrunk_fragment_bytes()), (ctx.lib.rcgan_dtrunk_fragment_bytes(),), "u8", ctx.arena.buf)
        ctx.check(ctx.lib.rcgan_dtrunk_prepare(ctx.h, arr(preps), _p(frag)))
    feat = ctx.empty((n, 128), L.F32) if pool is not None else None
    ctx.check(ctx.lib.rcgan_dtrunk_pooled(ctx.h, n, 0, _p(x), _p(frag), arr([b for _, b in flat]), None, arr(outs), _p(feat), None, None))
    y = outs[7]
    y.pooled, y.pool_grad = ((feat, pool) if feat is not None else None), None
    params = [t for w, b in flat for t in (w.param, b)]
    if _track(ctx, y, x, *params):
        def bw():
            dy, dfeat = y.grad, y.pool_grad
            if dy is None and dfeat is None:
                return
            if dfeat is not None:
                assert dy is None, "the stage's output has one consumer: the pooled features or the activations"
                dy = ctx.empty(x.shape, x.dtype)      # written by the launch (formed from dfeat): the last layer's filter gradient reads it
            # layers last to first: block 6 conv2 (mask h_6), block 6 conv1 (mask x_6), block 5 conv2, ...
            hs = [outs[0], outs[2], outs[4], outs[6]]                    # h_3 .. h_6
            xs = [x, outs[1], outs[3], outs[5]]                          # x_3 .. x_6 (block inputs)
            order = [(3, 1), (3, 0), (2, 1), (2, 0), (1, 1), (1, 0), (0, 1), (0, 0)]      # (block index, conv index)
            masks = [hs[bi] if ci == 1 else xs[bi] for bi, ci in order]
            gouts = [ctx.empty(x.shape, x.dtype) for _ in range(8)]
            if dfeat is not None:
                ctx.check(ctx.lib.rcgan_dtrunk_pooled(ctx.h, n, 1, None, _p(frag), None, arr(masks), arr(gouts), _p(dfeat), _p(y), _p(dy)))
            else:
                ctx.check(ctx.lib.rcgan_dtrunk(ctx.h, n, 1, _p(dy), _p(frag), None, arr(masks), arr(gouts)))
            for j, bi in enumerate((3, 2, 1, 0)):
                dh, dy_k = gouts[2 * j], (dy if j == 0 else gouts[2 * j - 1])
                for (w, b), xin, g in ((flat[2 * bi + 1], hs[bi], dy_k), (flat[2 * bi], xs[bi], dh)):
                    if not w.req:
                        continue
                    db = b.grad if (b is not None and b.req) else None
                    if ctx.group_wgrads:
                        ctx.defer_wgrad(desc, xin, g, w.grad_target(), db)      # read again at flush_wgrads: never written below
                    else:
                        ctx.check(ctx.lib.rcgan_conv2d_bwd_weight(ctx.h, C.byref(desc), _p(xin), _p(g), _p(w.grad_target()), _p(db), 1,
                                                                  C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
            if x.req:
                if x.grad is None:
                    x.grad = gouts[7]
                else:
                    ctx.check(ctx.lib.rcgan_axpby(ctx.h, x.size, x.dtype, 1.0, _p(gouts[7]), 1.0, _p(x.grad)))
        ctx.record(bw)
    return y


def deconv2d(ctx, x, w, bias, out_shape, k=5, stride=2):
    """tf.nn.conv2d_transpose(x, w[kh,kw,Cout,Cin], out_shape, strides) + bias (mnist/ops.py:78-86)."""
    n, oh, ow, cout = out_shape
    cin = x.shape[-1]
    assert w.shape == (k, k, cout, cin), (w.shape, (k, k, cout, cin))
    desc = L.ConvDesc(n, oh, ow, cout, cin, k, k, stride, x.dtype, 0)     # the forward conv it transposes
    y = ctx.empty(out_shape, x.dtype)
    ctx.check(ctx.lib.rcgan_deconv2d_fwd(ctx.h, C.byref(desc), _p(x), _p(w), _p(bias), _p(y)))
    if _track(ctx, y, x, w, bias):
        def bw():
            dy = y.grad
            if dy is None:
                return
            src = getattr(x, "concat_src", None)
            if x.req and src is not None and CONCAT_DIRECT:
                # x = conv_cond_concat(xs, labels): only the xs channels need a gradient -- produced straight into xs's gradient
                # (x.grad stays None: the concatenation's own backward then has nothing to split)
                xs, c1 = src
                dxs, acc = grad_of(ctx, xs)
                dsc = L.ConvDesc(desc.n, desc.h, desc.w, desc.cin, desc.cout, desc.kh, desc.kw, desc.stride, desc.dtype,
                                 desc.flags | (L.CONV_ACCUMULATE if acc else 0))
                ctx.check(ctx.lib.rcgan_deconv2d_bwd_data_cols(ctx.h, C.byref(dsc), _p(dy), _p(w), _p(dxs), c1))
            elif x.req:
                dx, acc = grad_of(ctx, x)
                if not acc:
                    ctx.check(ctx.lib.rcgan_deconv2d_bwd_data(ctx.h, C.byref(desc), _p(dy), _p(w), _p(dx)))
                else:
                    tmp = ctx.empty(x.shape, x.dtype)
                    ctx.check(ctx.lib.rcgan_deconv2d_bwd_data(ctx.h, C.byref(desc), _p(dy), _p(w), _p(tmp)))
                    ctx.check(ctx.lib.rcgan_axpby(ctx.h, x.size, x.dtype, 1.0, _p(tmp), 1.0, _p(dx)))
            if w.req:
                db = _p(bias.grad) if (bias is not None and bias.req) else None
                yb = getattr(x, "concat_labels", None)
                if (src is not None and yb is not None and CONCAT_WGRAD and 0 < cin - src[1] <= 16 and k <= 5 and src[1] >= 64 and src[1] % 4 == 0 and cout >= 64 and ow <= 32
                        and ctx.lib.rcgan_deconv2d_bwd_weight_concat_bytes(C.byref(desc), src[1]) <= ctx.ws_bytes):
                    # (round 6) x = conv_cond_concat(xs, yb): the GEMM over the real channels only, the label columns from per-sample sums
                    ctx.check(ctx.lib.rcgan_deconv2d_bwd_weight_concat(ctx.h, C.byref(desc), _p(x), _p(dy), src[1], _p(yb), _p(w.grad), db, 1,
                                                                       C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
                else:
                    ctx.check(ctx.lib.rcgan_deconv2d_bwd_weight(ctx.h, C.byref(desc), _p(x), _p(dy), _p(w.grad), db, 1,
                                                                C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
        ctx.record(bw)
    return y


def linear(ctx, x, weight, bias, out_dtype=None):
    """x[m,k] @ W[k,n] + b with W a Weight (optionally / sigma).  tf.matmul + bias: mnist/ops.py:114-116,
    cifar10/common/ops/linear.py:161-180."""
    m, kk = x.shape
    n = weight.param.shape[-1]
    assert weight.param.shape == (kk, n), (weight.param.shape, (kk, n))
    if LINEAR_MFMA and x.dtype != L.F32 and weight.sigma is None and kk % 64 == 0 and n % 64 == 0 and n >= 1024 and out_dtype is None:
        # a wide dense layer on 16-bit activations (G.Input: 128 -> 16384) as a 1x1 convolution on the bf16 / fp16 matrix cores with
        # its prepared 16-bit filter -- the layout is the same ([m, 1, 1, k] / [1, 1, k, n]); the fp32 gather GEMM ran it at 45 TFLOP/s
        y4 = conv2d(ctx, reshape(ctx, x, (m, 1, 1, kk)), weight.as_conv1x1(), bias, 1)
        return reshape(ctx, y4, (m, n))
    y = ctx.empty((m, n), x.dtype)
    ctx.check(ctx.lib.rcgan_linear_fwd(ctx.h, m, kk, n, x.dtype, _p(x), _p(weight.param), _p(weight.sigma), _p(bias), _p(y)))
    if _track(ctx, y, x, weight.param, bias):
        xr, wr, br = x.req, weight.req, (bias is not None and bias.req)

        def bw():
            dy = y.grad
            if dy is None:
                return
            src = getattr(x, "concat_src", None)
            if xr and src is not None and CONCAT_DIRECT:
                # x = concat(xs, labels) (model.py:710-714): the first c1 rows of W give xs's gradient, written straight into it
                xs, c1 = src
                dxs, acc = grad_of(ctx, xs)
                ctx.check(ctx.lib.rcgan_linear_bwd_data(ctx.h, m, c1, n, x.dtype, _p(dy), _p(weight.param), _p(weight.sigma), _p(dxs), acc))
            elif xr:
                dx, acc = grad_of(ctx, x)
                ctx.check(ctx.lib.rcgan_linear_bwd_data(ctx.h, m, kk, n, x.dtype, _p(dy), _p(weight.param), _p(weight.sigma), _p(dx), acc))
            if wr:
                dw = weight.grad_target()
                ctx.check(ctx.lib.rcgan_linear_bwd_weight(ctx.h, m, kk, n, x.dtype, _p(x), _p(dy), _p(dw),
                                                          _p(bias.grad) if br else None, 1, C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
        ctx.record(bw)
    return y


# ----------------------------------------------------------------------------------------------------
# normalisation
# ----------------------------------------------------------------------------------------------------
def _rows(x):
    if len(x.shape) == 4:
        return x.shape[0], x.shape[1] * x.shape[2], x.shape[3]
    return x.shape[0], 1, x.shape[1]


class BnPending:
    """act(batch_norm(x)) whose statistics exist and whose affine has NOT been applied: ops.conv2d applies it to its staged input
    (rcgan_conv2d_fwd_bn: the normalised tensor is never written) where the convolution can, else materialize() writes it.
    Forward-only passes (batch_norm_act(defer_apply=True) without a tape)."""

    def __init__(self, ctx, x, gamma, beta, labels, n_labels, mean, rstd, act, segments):
        self.ctx, self.x, self.gamma, self.beta, self.labels, self.n_labels = ctx, x, gamma, beta, labels, n_labels
        self.mean, self.rstd, self.act, self.segments = mean, rstd, act, segments
        self.shape, self.dtype, self.req = x.shape, x.dtype, False

    def materialize(self):
        ctx, x = self.ctx, self.x
        n, rps, c = _rows(x)
        y = ctx.empty(x.shape, x.dtype)
        if self.segments > 1:
            ctx.check(ctx.lib.rcgan_bn_apply_segments(ctx.h, self.segments, n // self.segments, rps, c, self.n_labels, x.dtype, _p(x), _p(self.labels),
                                                      _p(self.gamma), _p(self.beta), _p(self.mean), _p(self.rstd), self.act, _p(y),
                                                      C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
        else:
            ctx.check(ctx.lib.rcgan_bn_apply_fwd(ctx.h, n, rps, c, self.n_labels, x.dtype, _p(x), _p(self.labels), _p(self.gamma), _p(self.beta),
                                                 _p(self.mean), _p(self.rstd), self.act, _p(y), C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
        return y


BN_INTO_CONV = os.environ.get("RCGAN_BN_INTO_CONV", "1") != "0"


def batch_norm_act(ctx, x, gamma, beta, act=L.ACT_NONE, labels=None, n_labels=1, moving=None, decay=0.9, eps=1e-5, segments=1,
                   defer_apply=False):
    """Batch statistics + (conditional) affine + activation, one fused op.
    labels=None: tf.contrib.layers.batch_norm (mnist/ops.py:38-44), ``moving`` = (moving_mean, moving_var)
    DTs updated in place.  labels=int32 DT [n]: cond_batchnorm (cifar10/common/ops/normalization.py:27-59),
    gamma/beta are [n_labels, c] tables.  segments > 1 (forward only): x holds that many independent batches back to
    back, each normalised with its own statistics -- several Generator() calls of the reference evaluated as one."""
    n, rps, c = _rows(x)
    rows = n * rps
    tile = getattr(x, "tile_stats", None) if moving is None else None      # statistics left by the producing convolution's epilogue
    if (defer_apply and BN_INTO_CONV and not ctx.recording and moving is None and tile is None and x.dtype != L.F32
            and n % max(segments, 1) == 0):
        # forward-only pass: statistics now, the affine + activation inside the consuming convolution (BnPending)
        mean = ctx.empty((max(segments, 1), c), L.F32)
        rstd = ctx.empty((max(segments, 1), c), L.F32)
        if segments > 1:
            ctx.check(ctx.lib.rcgan_bn_fwd_segments(ctx.h, segments, n // segments, rps, c, n_labels, x.dtype, _p(x), _p(labels), _p(gamma),
                                                    _p(beta), eps, act, _p(mean), _p(rstd), None, C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
        else:
            ctx.check(ctx.lib.rcgan_bn_stats(ctx.h, rows, c, x.dtype, _p(x), eps, _p(mean), _p(rstd), None, None, decay,
                                             C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
        return BnPending(ctx, x, gamma, beta, labels, n_labels, mean, rstd, act, max(segments, 1))
    if segments > 1 and labels is None and n % segments == 0 and ((ctx.recording and (x.req or gamma.req or beta.req)) or moving is not None):
        # Several independent batches back to back THROUGH the tape (the MNIST critic step: D(real) and D(fake) as one pass of 2B
        # images, model.py:131-179): every segment gets its own statistics, its own moving-average update -- in segment order, as the
        # reference's consecutive discriminator() calls apply them -- and its own backward reduction; the convolutions around it run
        # once on all of them.
        ns = n // segments
        y = ctx.empty(x.shape, x.dtype)
        mm, mv = moving if moving is not None else (None, None)
        stats = []
        for sg in range(segments):
            xs, ys = x.rows(sg * ns, (sg + 1) * ns), y.rows(sg * ns, (sg + 1) * ns)
            mean, rstd = ctx.empty((c,), L.F32), ctx.empty((c,), L.F32)
            ctx.check(ctx.lib.rcgan_bn_stats(ctx.h, ns * rps, c, x.dtype, _p(xs), eps, _p(mean), _p(rstd), _p(mm), _p(mv), decay,
                                             C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
            ctx.check(ctx.lib.rcgan_bn_apply_fwd(ctx.h, ns, rps, c, n_labels, x.dtype, _p(xs), None, _p(gamma), _p(beta), _p(mean), _p(rstd),
                                                 act, _p(ys), C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
            stats.append((mean, rstd))
        if _track(ctx, y, x, gamma, beta):
            def bw_seg():
                dy = y.grad
                if dy is None:
                    return
                dx, acc_dx = grad_of(ctx, x)
                # (frozen parameters: a scratch nobody reads -- the first segment overwrites it, no fill)
                dg, db = (gamma.grad, beta.grad) if gamma.req else (ctx.empty(gamma.shape, gamma.dtype), ctx.empty(beta.shape, beta.dtype))
                for sg, (mean, rstd) in enumerate(stats):
                    lo, hi = sg * ns, (sg + 1) * ns
                    ctx.check(ctx.lib.rcgan_bn_bwd2(ctx.h, ns, rps, c, n_labels, x.dtype, _p(x.rows(lo, hi)), _p(y.rows(lo, hi)), _p(dy.rows(lo, hi)),
                                                    None, _p(gamma), _p(beta), _p(mean), _p(rstd), act, _p(dx.rows(lo, hi)), acc_dx, _p(dg), _p(db),
                                                    1 if (gamma.req or sg) else 0, C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
            ctx.record(bw_seg)
        return y
    if segments > 1:
        if (ctx.recording and (x.req or gamma.req or beta.req)) or moving is not None or n % segments:
            raise NotImplementedError("segmented conditional batch norm is forward-only (no gradient, no moving statistics)")
        mean = ctx.empty((segments, c), L.F32)
        rstd = ctx.empty((segments, c), L.F32)
        y = ctx.empty(x.shape, x.dtype)
        if tile is not None:
            ctx.check(ctx.lib.rcgan_bn_stats_from_tiles(ctx.h, C.byref(tile[0]), segments, eps, _p(tile[1]), _p(mean), _p(rstd)))
            ctx.check(ctx.lib.rcgan_bn_apply_segments(ctx.h, segments, n // segments, rps, c, n_labels, x.dtype, _p(x), _p(labels), _p(gamma), _p(beta),
                                                      _p(mean), _p(rstd), act, _p(y), C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
            return y
        ctx.check(ctx.lib.rcgan_bn_fwd_segments(ctx.h, segments, n // segments, rps, c, n_labels, x.dtype, _p(x), _p(labels), _p(gamma),
                                                _p(beta), eps, act, _p(mean), _p(rstd), _p(y), C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
        return y
    mean = ctx.empty((c,), L.F32)
    rstd = ctx.empty((c,), L.F32)
    mm, mv = moving if moving is not None else (None, None)
    if tile is not None:
        ctx.check(ctx.lib.rcgan_bn_stats_from_tiles(ctx.h, C.byref(tile[0]), 1, eps, _p(tile[1]), _p(mean), _p(rstd)))
    else:
        ctx.check(ctx.lib.rcgan_bn_stats(ctx.h, rows, c, x.dtype, _p(x), eps, _p(mean), _p(rstd), _p(mm), _p(mv), decay,
                                         C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
    y = ctx.empty(x.shape, x.dtype)
    ctx.check(ctx.lib.rcgan_bn_apply_fwd(ctx.h, n, rps, c, n_labels, x.dtype, _p(x), _p(labels), _p(gamma), _p(beta), _p(mean), _p(rstd),
                                         act, _p(y), C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
    if _track(ctx, y, x, gamma, beta):
        def bw():
            dy = y.grad
            if dy is None:
                return
            dx, acc_dx = grad_of(ctx, x)
            # parameter gradients accumulate into the (zeroed) slab; frozen params get a scratch that is overwritten (no fill)
            if gamma.req:
                dg, db = gamma.grad, beta.grad
            else:
                dg = ctx.empty(gamma.shape, gamma.dtype)
                db = ctx.empty(beta.shape, beta.dtype)
            # beta lets the fused kernels recompute the ReLU mask from x (the forward's exact arithmetic) instead of reading y
            ctx.check(ctx.lib.rcgan_bn_bwd2(ctx.h, n, rps, c, n_labels, x.dtype, _p(x), _p(y), _p(dy), _p(labels), _p(gamma), _p(beta),
                                            _p(mean), _p(rstd), act, _p(dx), acc_dx, _p(dg), _p(db), 1 if gamma.req else 0,
                                            C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
        ctx.record(bw)
    return y


def batch_norm_infer(ctx, x, gamma, beta, mm, mv, act=L.ACT_NONE, eps=1e-5):
    """Inference-mode batch norm (moving statistics; gen_sampler, mnist/model.py:745-754).  Differentiable w.r.t. x only:
    recover_labels (model.py:494-640) optimises the sampler's INPUT with every parameter and statistic frozen."""
    n, rps, c = _rows(x)
    y = ctx.empty(x.shape, x.dtype)
    ctx.check(ctx.lib.rcgan_bn_infer(ctx.h, n * rps, c, x.dtype, _p(x), _p(gamma), _p(beta), _p(mm), _p(mv), eps, act, _p(y)))
    if _track(ctx, y, x):
        if gamma.req or beta.req:
            raise NotImplementedError("inference-mode batch norm has no parameter gradients")

        def bw():
            if y.grad is None:
                return
            dx, acc = grad_of(ctx, x)
            ctx.check(ctx.lib.rcgan_bn_infer_bwd(ctx.h, n * rps, c, x.dtype, _p(y), _p(y.grad), _p(gamma), _p(mv), eps, act, _p(dx), acc))
        ctx.record(bw)
    return y


def recover_mse(ctx, gen, actual, yrec, loss):
    """recover_labels objective (mnist/model.py:533-537): mean_r sum_y yrec[r,y] * mean_pix (actual[r] - gen[r*ydim+y])^2,
    written to ``loss`` [1]; gradients flow to gen and yrec."""
    r, ydim = yrec.shape
    pix = gen.size // (r * ydim)
    assert gen.shape[0] == r * ydim and actual.size == r * pix and gen.dtype == actual.dtype
    rec = _track(ctx, loss, gen, yrec)
    dgen = ctx.empty(gen.shape, gen.dtype) if (rec and gen.req) else None
    dyr = ctx.empty(yrec.shape, L.F32) if (rec and yrec.req) else None
    ctx.check(ctx.lib.rcgan_recover_mse_fwd_bwd(ctx.h, r, ydim, pix, gen.dtype, _p(gen), _p(actual), _p(yrec), _p(loss), _p(dgen), _p(dyr),
                                                C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
    if rec:
        def bw():
            for t, d in ((gen, dgen), (yrec, dyr)):
                if d is None:
                    continue
                if t.grad is None:
                    t.grad = d
                else:
                    ctx.check(ctx.lib.rcgan_axpby(ctx.h, t.size, t.dtype, 1.0, _p(d), 1.0, _p(t.grad)))
        ctx.record(bw)
    return loss


# ----------------------------------------------------------------------------------------------------
# elementwise / resampling
# ----------------------------------------------------------------------------------------------------
def act(ctx, x, kind, out=None):
    y = out if out is not None else ctx.empty(x.shape, x.dtype)
    ctx.check(ctx.lib.rcgan_act_fwd(ctx.h, x.size, x.dtype, kind, _p(x), _p(y)))
    if _track(ctx, y, x):
        def bw():
            if y.grad is None:
                return
            dx, acc = grad_of(ctx, x)
            s = y if kind in (L.ACT_TANH, L.ACT_SIGMOID) else x
            ctx.check(ctx.lib.rcgan_act_bwd(ctx.h, x.size, x.dtype, kind, _p(s), _p(y.grad), _p(dx), acc))
        ctx.record(bw)
    return y


def add(ctx, a, b):
    y = ctx.empty(a.shape, a.dtype)
    assert a.shape == b.shape
    ctx.check(ctx.lib.rcgan_add(ctx.h, a.size, a.dtype, _p(a), _p(b), _p(y)))
    if _track(ctx, y, a, b):
        def bw():
            if y.grad is None:
                return
            # y.grad is dead after this closure, so ONE input without a gradient yet may simply adopt the buffer
            # (later contributions accumulate into it in place); every other input gets a copy / an accumulate
            adopted = False
            for t in (b, a):
                if not t.req:
                    continue
                if t.grad is None and not adopted and t.shape == y.grad.shape:
                    t.grad = y.grad
                    adopted = True
                    continue
                dx, acc = grad_of(ctx, t)
                ctx.check(ctx.lib.rcgan_axpby(ctx.h, t.size, t.dtype, 1.0, _p(y.grad), float(acc), _p(dx)))
        ctx.record(bw)
    return y


def meanpool2(ctx, x):
    n, h, w, c = x.shape
    y = ctx.empty((n, h // 2, w // 2, c), x.dtype)
    ctx.check(ctx.lib.rcgan_meanpool2_fwd(ctx.h, n, h, w, c, x.dtype, _p(x), _p(y)))
    if _track(ctx, y, x):
        def bw():
            if y.grad is None:
                return
            dx, acc = grad_of(ctx, x)
            ctx.check(ctx.lib.rcgan_meanpool2_bwd(ctx.h, n, h, w, c, x.dtype, _p(y.grad), _p(dx), acc))
        ctx.record(bw)
    return y


def upsample2(ctx, x):
    n, h, w, c = x.shape
    y = ctx.empty((n, h * 2, w * 2, c), x.dtype)
    ctx.check(ctx.lib.rcgan_upsample2_fwd(ctx.h, n, h * 2, w * 2, c, x.dtype, _p(x), _p(y)))
    if _track(ctx, y, x):
        def bw():
            if y.grad is None:
                return
            dx, acc = grad_of(ctx, x)
            ctx.check(ctx.lib.rcgan_upsample2_bwd(ctx.h, n, h * 2, w * 2, c, x.dtype, _p(y.grad), _p(dx), acc))
        ctx.record(bw)
    return y


def concat_channels(ctx, x, yb):
    """conv_cond_concat (mnist/ops.py:46-51): append the one-hot label yb[n,c2] (fp32) to every pixel."""
    if len(x.shape) == 4:
        n, h, w, c1 = x.shape
        hw = h * w
        oshape = (n, h, w, c1 + yb.shape[1])
    else:
        n, c1 = x.shape
        hw = 1
        oshape = (n, c1 + yb.shape[1])
    c2 = yb.shape[1]
    y = ctx.empty(oshape, x.dtype)
    ctx.check(ctx.lib.rcgan_concat_channels_fwd(ctx.h, n, hw, c1, c2, x.dtype, _p(x), _p(yb), _p(y)))
    y.concat_src = (x, c1)       # a consumer whose data gradient can be limited to the x channels bypasses the split below
    y.concat_labels = yb         # ... and one whose filter gradient knows the label columns are per-sample constants skips their column tile
    if _track(ctx, y, x):
        def bw():
            if y.grad is None:
                return
            dx, acc = grad_of(ctx, x)
            if not acc:
                ctx.check(ctx.lib.rcgan_concat_channels_bwd(ctx.h, n, hw, c1, c2, x.dtype, _p(y.grad), _p(dx)))
            else:
                tmp = ctx.empty(x.shape, x.dtype)
                ctx.check(ctx.lib.rcgan_concat_channels_bwd(ctx.h, n, hw, c1, c2, x.dtype, _p(y.grad), _p(tmp)))
                ctx.check(ctx.lib.rcgan_axpby(ctx.h, x.size, x.dtype, 1.0, _p(tmp), 1.0, _p(dx)))
        ctx.record(bw)
    return y


def tile_rows(ctx, x, reps):
    """[n, ...] -> [reps * n, ...]: the batch `reps` times back to back (one discriminator pass over every label on the same
    images, mnist/model.py:152-163,187-197); the adjoint sums the copies' gradients."""
    y = ctx.empty((reps * x.shape[0],) + tuple(x.shape[1:]), x.dtype)
    ctx.check(ctx.lib.rcgan_tile_rows_fwd(ctx.h, x.size, reps, x.dtype, _p(x), _p(y)))
    if _track(ctx, y, x):
        def bw():
            if y.grad is None:
                return
            dx, acc = grad_of(ctx, x)
            ctx.check(ctx.lib.rcgan_tile_rows_bwd(ctx.h, x.size, reps, x.dtype, _p(y.grad), _p(dx), acc))
        ctx.record(bw)
    return y


def transpose2d(ctx, x):
    """fp32 [r, c] -> [c, r] (tf.concat(D_logits_all, 1) of per-label passes laid out label-major, mnist/model.py:165,199)."""
    assert x.dtype == L.F32 and len(x.shape) == 2, (x.dtype, x.shape)
    r, c = x.shape
    y = ctx.empty((c, r), L.F32)
    ctx.check(ctx.lib.rcgan_transpose_f32(ctx.h, r, c, _p(x), _p(y), 0))
    if _track(ctx, y, x):
        def bw():
            if y.grad is None:
                return
            dx, acc = grad_of(ctx, x)
            ctx.check(ctx.lib.rcgan_transpose_f32(ctx.h, c, r, _p(y.grad), _p(dx), acc))
        ctx.record(bw)
    return y


def reshape(ctx, x, shape):
    y = x.reshape(shape)
    y.req = x.req
    if x.req and ctx.recording:
        if x.grad is None:
            x.grad = zeros_like_grad(ctx, x)
        y.grad = x.grad.reshape(y.shape)
    return y


def rows(ctx, x, lo, hi):
    y = x.rows(lo, hi)
    y.req = x.req
    if x.req and ctx.recording:
        if x.grad is None:
            x.grad = zeros_like_grad(ctx, x)
        y.grad = x.grad.rows(lo, hi)
    return y


def cast(ctx, x, dtype):
    if x.dtype == dtype:
        return x
    y = ctx.empty(x.shape, dtype)
    ctx.check(ctx.lib.rcgan_cast(ctx.h, x.size, x.dtype, _p(x), dtype, _p(y)))
    if _track(ctx, y, x):
        def bw():
            if y.grad is None:
                return
            dx, acc = grad_of(ctx, x)
            if not acc:
                ctx.check(ctx.lib.rcgan_cast(ctx.h, x.size, dtype, _p(y.grad), x.dtype, _p(dx)))
            else:
                tmp = ctx.empty(x.shape, x.dtype)
                ctx.check(ctx.lib.rcgan_cast(ctx.h, x.size, dtype, _p(y.grad), x.dtype, _p(tmp)))
                ctx.check(ctx.lib.rcgan_axpby(ctx.h, x.size, x.dtype, 1.0, _p(tmp), 1.0, _p(dx)))
        ctx.record(bw)
    return y


# ----------------------------------------------------------------------------------------------------
# discriminator head / losses (all fp32)
# ----------------------------------------------------------------------------------------------------
def act_meanhw(ctx, x, kind):
    """mean over (H,W) of act(x) -> fp32 [n,c]  (gan_resnet.py:405-407; mnist/model.py:679 with ACT_NONE)."""
    n, h, w, c = x.shape
    y = ctx.empty((n, c), L.F32)
    ctx.check(ctx.lib.rcgan_act_meanhw_fwd(ctx.h, n, h * w, c, x.dtype, kind, _p(x), _p(y)))
    if _track(ctx, y, x):
        def bw():
            if y.grad is None:
                return
            assert x.grad is None, "act_meanhw input has a single consumer in both models"
            dx, _ = grad_of(ctx, x)
            ctx.check(ctx.lib.rcgan_act_meanhw_bwd(ctx.h, n, h * w, c, x.dtype, kind, _p(x), _p(y.grad), _p(dx)))
        ctx.record(bw)
    return y


class PooledLater:
    """act_meanhw(x, kind) that has not run: ops.proj_head pools the features inside its own launch and sends the gradient
    straight back to x (the two act_meanhw launches around the head disappear); any other consumer calls materialize()."""

    def __init__(self, ctx, x, kind):
        self.ctx, self.x, self.kind = ctx, x, kind
        n, h, w, c = x.shape
        self.shape = (n, c)
        self.req = x.req

    def materialize(self):
        return act_meanhw(self.ctx, self.x, self.kind)


class PooledByProducer:
    """act_meanhw(x, kind) whose values the producer of x already left in x.pooled (d_trunk): a consumer that computes the
    features' gradient itself (proj_head) hands it back through x.pool_grad; any other consumer calls materialize()."""

    def __init__(self, ctx, x, kind):
        self.ctx, self.x, self.kind = ctx, x, kind
        self.feat = x.pooled[0]
        self.shape = self.feat.shape
        self.req = x.req

    def materialize(self):
        return act_meanhw(self.ctx, self.x, self.kind)


def act_meanhw_later(ctx, x, kind):
    """act_meanhw for a consumer that can pool by itself (proj_head with d % 128 == 0 channels), else act_meanhw."""
    pooled = x.pooled
    if pooled is not None and pooled[1] == kind:
        return PooledByProducer(ctx, x, kind)
    if os.environ.get("RCGAN_HEAD_POOL", "1") == "1" and x.shape[-1] % 128 == 0 and x.shape[-1] <= HEAD_MAX_D:
        return PooledLater(ctx, x, kind)
    return act_meanhw(ctx, x, kind)


def gather_rows(ctx, table, idx, n):
    """tf.nn.embedding_lookup(table, idx) (embedding.py:51); idx: int32 DT [n]."""
    v, d = table.shape
    y = ctx.empty((n, d), L.F32)
    ctx.check(ctx.lib.rcgan_gather_rows(ctx.h, n, d, _p(table), _p(idx), _p(y)))
    if _track(ctx, y, table):
        def bw():
            if y.grad is None:
                return
            if table.grad is None:
                table.grad = zeros_like_grad(ctx, table)
            ctx.check(ctx.lib.rcgan_scatter_add_rows(ctx.h, n, d, v, _p(y.grad), _p(idx), _p(table.grad), 1))
        ctx.record(bw)
    return y


def proj_logit(ctx, feat, psi, emb):
    """psi[n] + sum(feat*emb, axis=1)  (gan_resnet.py:588; mnist/model.py:685)."""
    n, d = feat.shape
    y = ctx.empty((n,), L.F32)
    ctx.check(ctx.lib.rcgan_proj_logit_fwd(ctx.h, n, d, _p(feat), _p(psi), _p(emb), _p(y)))
    if _track(ctx, y, feat, psi, emb):
        def bw():
            if y.grad is None:
                return
            dfeat = dpsi = demb = None
            accf = 0
            if feat.req:
                dfeat, accf = grad_of(ctx, feat)
            if psi is not None and psi.req:
                assert psi.grad is None or psi.grad.ptr != 0
                if psi.grad is None:
                    psi.grad = ctx.empty(psi.shape, L.F32)
                    dpsi = psi.grad
                else:
                    # psi gradient already holds another consumer's contribution: go through a scratch
                    dpsi = ctx.empty(psi.shape, L.F32)
            if emb.req:
                assert emb.grad is None
                demb, _ = grad_of(ctx, emb)
            ctx.check(ctx.lib.rcgan_proj_logit_bwd(ctx.h, n, d, _p(feat), _p(emb), _p(y.grad), _p(dfeat), _p(dpsi), _p(demb), accf))
            if dpsi is not None and dpsi is not psi.grad:
                ctx.check(ctx.lib.rcgan_axpby(ctx.h, psi.size, L.F32, 1.0, _p(dpsi), 1.0, _p(psi.grad)))
        ctx.record(bw)
    return y


def proj_logit_all(ctx, feat, psi, E):
    """logits[n,v] = psi[n] + feat @ E^T for every label (gan_resnet.py:654-660)."""
    n, d = feat.shape
    v = E.shape[0]
    y = ctx.empty((n, v), L.F32)
    ctx.check(ctx.lib.rcgan_proj_logit_all_fwd(ctx.h, n, d, v, _p(feat), _p(psi), _p(E), _p(y)))
    if _track(ctx, y, feat, psi, E):
        def bw():
            if y.grad is None:
                return
            dfeat, accf = grad_of(ctx, feat)
            if psi.grad is None:
                psi.grad = ctx.empty(psi.shape, L.F32)
                dpsi = psi.grad
            else:
                dpsi = ctx.empty(psi.shape, L.F32)
            assert E.grad is None
            dE, _ = grad_of(ctx, E)
            ctx.check(ctx.lib.rcgan_proj_logit_all_bwd(ctx.h, n, d, v, _p(feat), _p(E), _p(y.grad), _p(dfeat), _p(dpsi), _p(dE), accf))
            if dpsi is not psi.grad:
                ctx.check(ctx.lib.rcgan_axpby(ctx.h, psi.size, L.F32, 1.0, _p(dpsi), 1.0, _p(psi.grad)))
        ctx.record(bw)
    return y


HEAD_MAX_N = 1024
HEAD_MAX_D = 256
# the head's parameter-gradient launches ride in the 8x8 stage's backward launch and the grouped filter-gradient launch (head_rider.h)
HEAD_RIDERS = os.environ.get("RCGAN_HEAD_RIDERS", "1") != "0"


def proj_head(ctx, feat, w_out, b_out, table, w_e, b_e, parts, weight, loss_acc, logits=None, E_pre=None):
    """The whole projection head in one launch (rcgan_proj_head_fwd_bwd): psi = Linear_SN(feat) (gan_resnet.py:408-411),
    E = Linear_SN(embed_y(l)) for every label l (:414-421), logit[s,l] = psi[s] + <feat[s], E[l]> (:588, :654-660), the loss terms
    and every gradient.  w_out / w_e: Weight handles (sigma fused); table: the embedding_map parameter.
    parts: one or two (rows, kind, labels, wts) tuples covering feat's rows in order -- labels: int32 DT [rows] (one-hot
    weighting) or wts: fp32 DT [rows, v] (may require a gradient).  Adds weight * (mean over each part's rows) to loss_acc.
    Like loss_term this computes its gradients in the forward launch: feat (and a wts that requires one) get their .grad
    here; the parameter gradients are accumulated into the step's zeroed buffers by two launches that are DEFERRED (HEAD_RIDERS:
    they ride in the 8x8 stage's backward launch and the grouped filter-gradient launch) -- complete after Context.flush_wgrads() /
    backward(), which every consumer of those gradients (the spectral-norm backward, the optimiser) sits behind."""
    n, d = feat.shape
    v, e_dim = table.shape
    assert 1 <= len(parts) <= 2 and sum(p[0] for p in parts) == n, (parts, n)
    rec = ctx.recording
    hd = L.HeadDesc(n, d, v, e_dim, parts[0][0], parts[0][1], parts[1][1] if len(parts) > 1 else 0, float(weight))
    for (rows, kind, labels, wts), sfx in zip(parts, ("a", "b")):
        assert (labels is None) != (wts is None)
        setattr(hd, "labels_" + sfx, labels.ptr if labels is not None else None)
        setattr(hd, "wts_" + sfx, wts.ptr if wts is not None else None)
        if wts is not None and wts.req and rec:
            dw, _ = grad_of(ctx, wts)
            setattr(hd, "dwts_" + sfx, dw.ptr)
    if E_pre is not None:          # label embeddings of this step's parameters, computed with the filter preparation
        assert E_pre.shape == (v, d)
        hd.E_pre = E_pre.ptr
    dfeat = None
    if isinstance(feat, PooledByProducer):
        x = feat.x
        feat = feat.feat
        if x.req and rec:
            assert x.grad is None and x.pool_grad is None, "the pooled features have a single consumer in both models"
            dfeat = ctx.empty((n, d), L.F32)
            x.pool_grad = dfeat
    elif isinstance(feat, PooledLater):
        # pooled inside the launch from the trunk's output; feat becomes an output buffer for the parameter-gradient kernels
        x = feat.x
        hd.x, hd.x_dtype, hd.hw, hd.act = x.ptr, x.dtype, x.shape[1] * x.shape[2], feat.kind
        if x.req and rec:
            assert x.grad is None, "act_meanhw input has a single consumer in both models"
            dx, _ = grad_of(ctx, x)
            hd.dx = dx.ptr
        feat = ctx.empty((n, d), L.F32)
    elif feat.req and rec:
        assert feat.grad is None, "the projection head is its features' only consumer"
        dfeat, _ = grad_of(ctx, feat)
    pg = lambda t: _p(t) if (rec and t is not None) else None
    if rec and HEAD_RIDERS and (w_out.req or w_e.req or table.req):
        # deferred parameter gradients (rcgan_head_desc::defer_ws): dlogit and dE in the step arena until the launches that carry them
        nbytes = (n * (v + 1) + (v + 1) * d) * 4
        hd.defer_ws, hd.defer_ws_bytes = ctx.arena.alloc(nbytes), nbytes
    dw_out = w_out.grad_target() if (rec and w_out.req) else None
    dw_e = w_e.grad_target() if (rec and w_e.req) else None
    db_out = b_out.grad if (b_out is not None and b_out.req) else None
    db_e = b_e.grad if (b_e is not None and b_e.req) else None
    dtable = table.grad if table.req else None
    ctx.check(ctx.lib.rcgan_proj_head_fwd_bwd(
        ctx.h, C.byref(hd), _p(feat), _p(w_out.param), _p(w_out.sigma), _p(b_out), _p(table), _p(w_e.param), _p(w_e.sigma), _p(b_e),
        _p(loss_acc), _p(logits), _p(dfeat), pg(dw_out), pg(db_out), pg(dtable), pg(dw_e), pg(db_e), C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))


def loss_term(ctx, kind, x, weight, loss_acc, wts=None):
    """Adds weight * L(x) to the device scalar loss_acc; the gradient weight*dL/dx is computed in the
    same launch (the total cost is a weighted sum of such terms, so no upstream gradient is needed)."""
    if len(x.shape) == 1:
        r, c = x.shape[0], 1
    else:
        r, c = x.shape
    need = x.req and ctx.recording
    # the kernel WRITES its gradients: x (and wts) must have this loss term as their only consumer
    dx = None
    if need:
        dx, _ = grad_of(ctx, x)
    dw = None
    if wts is not None and wts.req and ctx.recording:
        dw, _ = grad_of(ctx, wts)
    ctx.check(ctx.lib.rcgan_loss_fwd_bwd(ctx.h, kind, r, c, _p(x), _p(wts), float(weight), _p(loss_acc), _p(dx), _p(dw)))


def bce_onehot_term(ctx, x, labels, weight, loss_acc):
    """weight * reduce_mean(sigmoid_cross_entropy_with_logits(x, one_hot(labels))) (gan_resnet.py:693-694)."""
    r, c = x.shape
    dx = None
    if x.req and ctx.recording:
        dx, _ = grad_of(ctx, x)
    ctx.check(ctx.lib.rcgan_bce_onehot_fwd_bwd(ctx.h, r, c, _p(x), _p(labels), float(weight), _p(loss_acc), _p(dx)))


def softmax_rows(ctx, logits):
    r, c = logits.shape
    p = ctx.empty((r, c), L.F32)
    ctx.check(ctx.lib.rcgan_softmax_rows_fwd(ctx.h, r, c, _p(logits), _p(p)))
    if _track(ctx, p, logits):
        def bw():
            if p.grad is None:
                return
            dl, acc = grad_of(ctx, logits)
            ctx.check(ctx.lib.rcgan_softmax_rows_bwd(ctx.h, r, c, _p(p), _p(p.grad), _p(dl), acc))
        ctx.record(bw)
    return p
