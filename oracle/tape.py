"""Minimal reverse-mode tape over oracle.nn (oracle; test infrastructure).

The reference builds a TF graph and lets ``tf.gradients`` differentiate it
(``compute_gradients`` at cifar10/gan_resnet.py:803,807; ``minimize`` at
mnist/model.py:250-262).  This tape plays that role for the numpy restatement:
ops record a closure, ``Tape.backward`` replays them in reverse.
"""
import numpy as np
from . import nn


class Var:
    __slots__ = ("v", "g", "req", "name")

    def __init__(self, v, req=False, name=None):
        self.v = v
        self.g = None
        self.req = req
        self.name = name

    @property
    def shape(self):
        return self.v.shape


def _acc(var, g):
    if var is None or not var.req:
        return
    g = np.asarray(g, dtype=var.v.dtype).reshape(var.v.shape)
    var.g = g if var.g is None else var.g + g


class Kinks:
    """Rectifier inputs within rounding of zero.  A float32 device and this float64 restatement can land on different sides of
    max(x, 0) / max(x, 0.2 x) when |x| is of the order of the fp32 summation error; the value does not care, the derivative
    does, and at batch 8 one such unit moves whole gradient tensors by a few per cent.  With ``eps`` set, every rectifier call
    lists its inputs with |x| <= eps * max(1, max|x|) in ``found`` (in call order) and those whose ordinal is in ``flip`` take
    the OTHER branch's derivative -- so a test can ask "is the device's gradient the oracle's gradient for SOME assignment of the
    undecidable units" instead of widening its tolerance for every tensor.  Off (eps None) everywhere else."""
    eps = None
    flip = frozenset()
    found = []

    @classmethod
    def reset(cls, eps=None, flip=()):
        cls.eps, cls.flip, cls.found = eps, frozenset(flip), []

    @classmethod
    def mask(cls, v):
        on = v > 0
        if cls.eps is not None:
            near = np.flatnonzero(np.abs(v) <= cls.eps * max(1.0, float(np.abs(v).max())))
            if near.size:
                on = on.copy()
                for i in near:
                    if len(cls.found) in cls.flip:
                        on.flat[i] = not on.flat[i]
                    cls.found.append((v.shape, int(i), float(v.flat[i])))
        return on


class Tape:
    def __init__(self):
        self.ops = []

    def rec(self, out, fn, *ins):
        if any(i is not None and i.req for i in ins):
            out.req = True
            self.ops.append(fn)
        return out

    def backward(self, loss, seed=1.0):
        loss.g = np.asarray(seed, dtype=loss.v.dtype).reshape(loss.v.shape)
        for fn in reversed(self.ops):
            fn()
        self.ops = []

    # ---------------------------------------------------------------- dense / conv
    def conv2d(self, x, w, b=None, stride=1):
        y = nn.conv2d_fwd(x.v, w.v, stride)
        if b is not None:
            y = y + b.v
        out = Var(y)

        def bw():
            if out.g is None:
                return
            if x.req:
                _acc(x, nn.conv2d_bwd_input(out.g, w.v, x.v.shape, stride))
            if w.req:
                _acc(w, nn.conv2d_bwd_filter(x.v, out.g, w.v.shape, stride))
            if b is not None and b.req:
                _acc(b, out.g.sum(axis=(0, 1, 2)))
        return self.rec(out, bw, x, w, b)

    def conv2d_transpose(self, x, w, b, out_shape, stride=2):
        y = nn.conv2d_transpose_fwd(x.v, w.v, out_shape, stride)
        if b is not None:
            y = y + b.v
        out = Var(y)

        def bw():
            if out.g is None:
                return
            if x.req:
                _acc(x, nn.conv2d_transpose_bwd_input(out.g, w.v, stride))
            if w.req:
                _acc(w, nn.conv2d_transpose_bwd_filter(x.v, out.g, w.v.shape, stride))
            if b is not None and b.req:
                _acc(b, out.g.sum(axis=(0, 1, 2)))
        return self.rec(out, bw, x, w, b)

    def linear(self, x, w, b=None):
        y = x.v @ w.v
        if b is not None:
            y = y + b.v
        out = Var(y)

        def bw():
            if out.g is None:
                return
            _acc(x, out.g @ w.v.T)
            _acc(w, x.v.T @ out.g)
            if b is not None:
                _acc(b, out.g.sum(axis=0))
        return self.rec(out, bw, x, w, b)

    # ---------------------------------------------------------------- elementwise
    def _unary(self, x, y, dfn):
        out = Var(y)

        def bw():
            if out.g is not None:
                _acc(x, dfn(out.g))
        return self.rec(out, bw, x)

    def relu(self, x):
        on = Kinks.mask(x.v)
        return self._unary(x, np.maximum(x.v, 0), lambda g: g * on)

    def lrelu(self, x, leak=0.2):
        # tf.maximum(x, leak*x): mnist/ops.py:94-95
        lk = x.v.dtype.type(leak)
        on = Kinks.mask(x.v)
        return self._unary(x, np.maximum(x.v, lk * x.v), lambda g: g * np.where(on, x.v.dtype.type(1), lk))

    def tanh(self, x):
        y = np.tanh(x.v)
        return self._unary(x, y, lambda g: g * (1 - y * y))

    def sigmoid(self, x):
        y = 1.0 / (1.0 + np.exp(-x.v))
        y = y.astype(x.v.dtype)
        return self._unary(x, y, lambda g: g * y * (1 - y))

    def scale(self, x, s):
        s = x.v.dtype.type(s)
        return self._unary(x, x.v * s, lambda g: g * s)

    def reshape(self, x, shape):
        return self._unary(x, x.v.reshape(shape), lambda g: g.reshape(x.v.shape))

    def add(self, a, b):
        out = Var(a.v + b.v)

        def bw():
            if out.g is None:
                return
            _acc(a, _unbroadcast(out.g, a.v.shape))
            _acc(b, _unbroadcast(out.g, b.v.shape))
        return self.rec(out, bw, a, b)

    def mul(self, a, b):
        out = Var(a.v * b.v)

        def bw():
            if out.g is None:
                return
            _acc(a, _unbroadcast(out.g * b.v, a.v.shape))
            _acc(b, _unbroadcast(out.g * a.v, b.v.shape))
        return self.rec(out, bw, a, b)

    def meanpool2(self, x):
        return self._unary(x, nn.meanpool2(x.v), nn.meanpool2_bwd)

    def upsample2(self, x):
        return self._unary(x, nn.upsample2(x.v), nn.upsample2_bwd)

    def mean_hw(self, x):
        n, h, w, c = x.v.shape
        s = x.v.dtype.type(1.0 / (h * w))
        return self._unary(x, x.v.mean(axis=(1, 2)),
                           lambda g: np.broadcast_to(g[:, None, None, :] * s, x.v.shape))

    def sum_axis(self, x, axis, keepdims=False):
        def d(g):
            if not keepdims:
                g = np.expand_dims(g, axis)
            return np.broadcast_to(g, x.v.shape)
        return self._unary(x, x.v.sum(axis=axis, keepdims=keepdims), d)

    def mean_all(self, x):
        s = x.v.dtype.type(1.0 / x.v.size)
        return self._unary(x, np.asarray(x.v.mean(), dtype=x.v.dtype),
                           lambda g: np.full(x.v.shape, g * s, dtype=x.v.dtype))

    def rows(self, x, lo, hi):
        def d(g):
            z = np.zeros_like(x.v)
            z[lo:hi] = g
            return z
        return self._unary(x, x.v[lo:hi], d)

    def concat(self, xs, axis):
        out = Var(np.concatenate([x.v for x in xs], axis=axis))

        def bw():
            if out.g is None:
                return
            o = 0
            for x in xs:
                k = x.v.shape[axis]
                sl = [slice(None)] * out.g.ndim
                sl[axis] = slice(o, o + k)
                _acc(x, out.g[tuple(sl)])
                o += k
        return self.rec(out, bw, *xs)

    def gather_rows(self, table, idx):
        """tf.nn.embedding_lookup / one_hot(idx) @ table."""
        idx = np.asarray(idx, dtype=np.int64)
        out = Var(table.v[idx])

        def bw():
            if out.g is None or not table.req:
                return
            g = np.zeros_like(table.v)
            np.add.at(g, idx, out.g)
            _acc(table, g)
        return self.rec(out, bw, table)

    # ---------------------------------------------------------------- normalisation
    def cond_batchnorm(self, x, labels, scale_m, offset_m):
        y, stats = nn.cond_batchnorm_fwd(x.v, labels, scale_m.v, offset_m.v)
        out = Var(y)

        def bw():
            if out.g is None:
                return
            dx, ds, do = nn.cond_batchnorm_bwd(out.g, x.v, labels, scale_m.v, stats)
            _acc(x, dx)
            _acc(scale_m, ds)
            _acc(offset_m, do)
        return self.rec(out, bw, x, scale_m, offset_m)

    def batch_norm_train(self, x, gamma, beta, state, decay=0.9, eps=1e-5):
        """state = dict(moving_mean=..., moving_variance=...) updated in place (updates_collections=None)."""
        y, stats, mm, mv = nn.batch_norm_train_fwd(x.v, gamma.v, beta.v, state["moving_mean"],
                                                   state["moving_variance"], decay, eps)
        state["moving_mean"], state["moving_variance"] = mm, mv
        out = Var(y)

        def bw():
            if out.g is None:
                return
            dx, dg, db = nn.batch_norm_train_bwd(out.g, x.v, gamma.v, stats)
            _acc(x, dx)
            _acc(gamma, dg)
            _acc(beta, db)
        return self.rec(out, bw, x, gamma, beta)

    def batch_norm_infer(self, x, gamma, beta, mm, mv, eps=1e-5):
        """Inference-mode batch norm (moving statistics), differentiable w.r.t. x only: the frozen sampler that
        recover_labels (mnist/model.py:494-640) optimises through."""
        out = Var(nn.batch_norm_infer(x.v, gamma.v, beta.v, mm, mv, eps))

        def bw():
            if out.g is None:
                return
            _acc(x, out.g * (gamma.v / np.sqrt(mv + eps)))
        return self.rec(out, bw, x)

    def spectral_norm(self, w, u_read, u_write, key, update):
        """u_read[key] is the persistent ``u`` ([1,C]) as it stood when the step began: every SN
        evaluation of one weight inside a step sees the same u (the reference leaves the order of
        ``u.assign`` vs reads inside one sess.run undefined; this build defines it).  ``update``
        False == update_collection NO_OPS: u_write is left alone."""
        u = np.asarray(u_read[key], dtype=w.v.dtype)
        wbar, sigma, u2, cache = nn.spectral_norm_fwd(w.v, u)
        if update:
            u_write[key] = u2
        out = Var(wbar)

        def bw():
            if out.g is not None:
                _acc(w, nn.spectral_norm_bwd(out.g, w.v, u, cache))
        return self.rec(out, bw, w)

    # ---------------------------------------------------------------- losses
    def sigmoid_ce_mean(self, logits, targets):
        """reduce_mean over ALL elements of sigmoid_cross_entropy_with_logits."""
        l = nn.sigmoid_ce_logits(logits.v, targets)
        s = logits.v.dtype.type(1.0 / l.size)
        return self._unary(logits, np.asarray(l.mean(), dtype=logits.v.dtype),
                           lambda g: nn.sigmoid_ce_logits_bwd(logits.v, targets).astype(logits.v.dtype) * (g * s))

    def sigmoid_ce(self, logits, targets):
        l = nn.sigmoid_ce_logits(logits.v, targets).astype(logits.v.dtype)
        return self._unary(logits, l,
                           lambda g: nn.sigmoid_ce_logits_bwd(logits.v, targets).astype(logits.v.dtype) * g)

    def softmax_rows(self, l):
        p = nn.softmax_rows(l.v)
        return self._unary(l, p, lambda g: nn.softmax_rows_bwd(g, p))


def _unbroadcast(g, shape):
    if g.shape == tuple(shape):
        return g
    while g.ndim > len(shape):
        g = g.sum(axis=0)
    for i, s in enumerate(shape):
        if s == 1 and g.shape[i] != 1:
            g = g.sum(axis=i, keepdims=True)
    return g
