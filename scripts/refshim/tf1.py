"""A small TensorFlow-1.x look-alike on top of PyTorch-CPU -- GOLDEN-VECTOR TOOLING ONLY (scripts/make_golden_reference.py).

TensorFlow 1.5, which the reference is written against, cannot be installed here.  This module implements just enough of its
graph-mode API (placeholders, variables + scopes with reuse checking, lazily evaluated ops, Session.run with feed_dict,
tf.gradients / AdamOptimizer, tf.while_loop with a static trip count, control_dependencies + assign) for the reference's OWN
model-building and training code (cifar10/gan_resnet.py, cifar10/common/ops/*.py, mnist/model.py, mnist/ops.py, mnist/sn.py) to
run unmodified when this module is installed as ``tensorflow`` in sys.modules.  What such a run pins is everything the
reference's Python decides: which variables exist (names, shapes, creation order, numpy initial values), how the towers / losses /
optimisers are wired, which feeds each session.run receives, the order of the training loop.  What it does NOT pin is the
arithmetic of the TensorFlow kernels themselves: each op below restates the documented TF semantics (SAME padding, biased
moments, ApplyAdam, ...) on torch float64/float32 tensors -- SURVEY.md Appendix C lists them.

Nothing in the product, the oracle or the tests imports this module; the vectors it produces are committed under tests/golden/.
"""
import contextlib
import types

import numpy as np
import torch
import torch.nn.functional as F

DT = torch.float64          # arithmetic dtype of the emulation (set_dtype)


def set_dtype(dt):
    global DT
    DT = dt


class State:
    """Everything global TF keeps in its default graph."""

    def __init__(self):
        self.variables = {}            # name -> Variable, creation order
        self.scope = []                # [(name, reuse)]
        self.control = []              # stack of lists of ops
        self.rng = np.random.RandomState(0)      # stands in for TensorFlow's own random ops / initialisers
        self.draws = []                # (kind, array) of the current Session.run
        self.collections = {}
        self.run_log = []              # one record per Session.run (filled by Session)
        self.adam_slots = {}
        self.flags = types.SimpleNamespace()


S = State()


def reset(tf_seed=0):
    global S
    S = State()
    S.rng = np.random.RandomState(tf_seed)


# ----------------------------------------------------------------------------------------------------------------------
# tensors
# ----------------------------------------------------------------------------------------------------------------------
class Dim(int):
    @property
    def value(self):
        return int(self)


class Shape:
    def __init__(self, dims):
        self.dims = None if dims is None else [None if d is None else int(d) for d in dims]

    def as_list(self):
        return list(self.dims)

    @property
    def ndims(self):
        return None if self.dims is None else len(self.dims)

    def __len__(self):
        return len(self.dims)

    def __iter__(self):
        return iter([Dim(d) for d in self.dims])

    def __getitem__(self, i):
        r = self.dims[i]
        return Shape(r) if isinstance(i, slice) else (None if r is None else Dim(r))

    def __repr__(self):
        return "Shape(%s)" % (self.dims,)


def _wrap(x):
    return x if isinstance(x, Tensor) else constant(x)


def _tt(x, dtype=None):
    """python / numpy value -> torch tensor in the emulation's dtype (integers stay integers)."""
    a = np.asarray(x)
    if a.dtype.kind in "iub":
        return torch.as_tensor(a.astype(np.int64))
    return torch.as_tensor(a.astype(np.float64)).to(dtype or DT)


class Tensor:
    def __init__(self, fn, inputs, name=None, shape=None):
        self.fn, self.inputs, self.name = fn, [_wrap(i) for i in inputs], name
        self.deps = [d for frame in S.control for d in frame]
        self._shape = shape
        self._const = None

    # -- static shape: run the op on zero tensors of the inputs' static shapes
    def _static(self):
        if self._shape is None:
            with torch.no_grad():
                zs = []
                for i in self.inputs:
                    v = i._probe()
                    zs.append(v)
                out = self.fn(*zs)
            self._probe_val = out
            self._shape = tuple(out.shape) if isinstance(out, torch.Tensor) else ()
        return self._shape

    def _probe(self):
        if getattr(self, "_probe_val", None) is None:
            if self._const is not None:
                self._probe_val = self._const
            else:
                self._static()
        return self._probe_val

    @property
    def shape(self):
        return Shape(self._static())

    def get_shape(self):
        return self.shape

    @property
    def dtype(self):
        return self._probe().dtype

    def eval(self, feed_dict=None, session=None):
        return Session._default.run(self, feed_dict)

    # -- operators
    def _bin(self, other, f, rev=False):
        a, b = (other, self) if rev else (self, other)
        return Tensor(f, [a, b])

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, lambda a, b: a + b, True)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: a - b, True)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, lambda a, b: a * b, True)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._bin(o, lambda a, b: a / b, True)
    __div__, __rdiv__ = __truediv__, __rtruediv__
    def __pow__(self, o): return self._bin(o, lambda a, b: a ** b)
    def __neg__(self): return Tensor(lambda a: -a, [self])
    def __lt__(self, o): return self._bin(o, lambda a, b: a < b)
    def __gt__(self, o): return self._bin(o, lambda a, b: a > b)
    def __hash__(self): return id(self)
    def __eq__(self, o): return self is o

    def __getitem__(self, idx):
        return Tensor(lambda a: a[idx], [self])


def _promote(a, b):
    return a, b


def constant(value, dtype=None, shape=None, name=None):
    v = _tt(value)
    if dtype is not None:
        v = v.to(_dtype(dtype))
    if shape is not None:
        v = v.expand(tuple(shape)).clone() if v.numel() == 1 else v.reshape(tuple(shape))
    t = Tensor(lambda: v, [], name)
    t._const = v
    t._shape = tuple(v.shape)
    return t


convert_to_tensor = lambda value, dtype=None, name=None: value if isinstance(value, Tensor) else constant(value, dtype)


class DTypeTag:
    def __init__(self, name, kind):
        self.name, self.kind = name, kind

    def __repr__(self):
        return "tf." + self.name


float32, float64, int32, int64, bool_ = DTypeTag("float32", "f"), DTypeTag("float64", "f"), DTypeTag("int32", "i"), DTypeTag("int64", "i"), DTypeTag("bool", "b")


def _dtype(tag):
    if isinstance(tag, torch.dtype):
        return tag
    return {"f": DT, "i": torch.int64, "b": torch.bool}[tag.kind]


NONE_DIM = 1          # what an unknown (None) placeholder dimension is probed with: the harness sets it to the batch size


class Placeholder(Tensor):
    def __init__(self, dtype, shape=None, name=None):
        super().__init__(None, [], name)
        self._dt = _dtype(dtype)
        self._shape = tuple(NONE_DIM if d is None else builtins_int(d) for d in shape) if shape is not None else ()
        self._probe_val = torch.zeros(self._shape, dtype=self._dt)


def placeholder(dtype, shape=None, name=None):
    return Placeholder(dtype, shape, name)


# ----------------------------------------------------------------------------------------------------------------------
# variables and scopes
# ----------------------------------------------------------------------------------------------------------------------
class Variable(Tensor):
    def __init__(self, initial_value=None, name=None, trainable=True, dtype=None):
        super().__init__(None, [], name)
        v = _tt(initial_value) if not isinstance(initial_value, torch.Tensor) else initial_value
        self.value = v.clone().detach().requires_grad_(v.dtype.is_floating_point)
        self.initial = self.value.detach().clone().numpy()
        self.trainable = trainable
        self._shape = tuple(v.shape)
        self._probe_val = self.value.detach()
        self.name = (name or "Variable") + ":0"
        self.constraint = None
        S.variables[self.name] = self
        self.op = types.SimpleNamespace(name=self.name[:-2])

    def assign(self, value):
        var = self

        def fn(v):
            Session._pending.append((var, v.detach()))
            return v
        return Tensor(fn, [value], name="assign/" + self.name)

    def read_value(self):
        return self


class _Scope:
    def __init__(self, name, reuse, depth=None):
        self.name, self.reuse, self.depth = name, reuse, depth

    def reuse_variables(self):
        self.reuse = True
        if self.depth is not None:
            S.scope[self.depth] = (S.scope[self.depth][0], True)


@contextlib.contextmanager
def variable_scope(name_or_scope=None, reuse=None, default_name=None):
    name = name_or_scope.name if isinstance(name_or_scope, _Scope) else name_or_scope
    S.scope.append((name, bool(reuse)))
    try:
        yield _Scope("/".join(n for n, _ in S.scope), reuse, len(S.scope) - 1)
    finally:
        S.scope.pop()


@contextlib.contextmanager
def name_scope(name=None, default_name=None, values=None):
    yield name


@contextlib.contextmanager
def device(name):
    yield


@contextlib.contextmanager
def control_dependencies(ops):
    S.control.append(list(ops))
    try:
        yield
    finally:
        S.control.pop()


class _Init:
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, shape, dtype=None):
        return self.fn(shape)


def constant_initializer(value=0.0, dtype=None):
    def fn(shape):
        a = np.asarray(value, np.float64)
        return np.broadcast_to(a, shape).astype(np.float32).copy() if a.ndim == 0 else a.reshape(shape).astype(np.float32)
    return _Init(fn)


def _trunc_normal(shape, mean=0.0, stddev=1.0):
    x = S.rng.normal(0.0, 1.0, size=shape)
    while True:
        bad = np.abs(x) > 2.0
        if not bad.any():
            break
        x[bad] = S.rng.normal(0.0, 1.0, size=builtins_int(bad.sum()))
    return (mean + stddev * x).astype(np.float32)


import builtins  # noqa: E402
builtins_int = builtins.int


def truncated_normal_initializer(mean=0.0, stddev=1.0, seed=None, dtype=None):
    t = _Init(lambda shape: _trunc_normal(shape, mean, stddev))
    t.kind = ("truncated_normal", mean, stddev)
    return t


def random_normal_initializer(mean=0.0, stddev=1.0, seed=None, dtype=None):
    t = _Init(lambda shape: (mean + stddev * S.rng.normal(size=shape)).astype(np.float32))
    t.kind = ("random_normal", mean, stddev)
    return t


def glorot_uniform_initializer(seed=None, dtype=None):
    def fn(shape):
        fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (np.prod(shape[:-1]), shape[-1])
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        return S.rng.uniform(-lim, lim, size=shape).astype(np.float32)
    t = _Init(fn)
    t.kind = ("glorot_uniform",)
    return t


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True, regularizer=None, collections=None, constraint=None):
    full = "/".join([n for n, _ in S.scope] + [name]) + ":0"
    reuse = any(r for _, r in S.scope)
    if full in S.variables:
        if not reuse:
            raise ValueError("Variable %s already exists, disallowed. Did you mean to set reuse=True in VarScope?" % full[:-2])
        return S.variables[full]
    if reuse:
        raise ValueError("Variable %s does not exist, or was not created with tf.get_variable()." % full[:-2])
    init_kind = None
    if isinstance(initializer, np.ndarray):
        val, init_kind = initializer, ("numpy",)
    elif initializer is None:            # TensorFlow's default for float variables: glorot_uniform_initializer
        ini = glorot_uniform_initializer()
        val, init_kind = ini(tuple(shape)), ini.kind
    elif isinstance(initializer, _Init):
        val, init_kind = initializer(tuple(shape)), getattr(initializer, "kind", ("constant",))
    else:
        val, init_kind = np.asarray(initializer), ("value",)
    v = Variable(val, name=full[:-2], trainable=trainable)
    v.init_kind = init_kind
    v.constraint = constraint          # applied to the variable after every optimiser update (tf.get_variable(constraint=...))
    return v


def trainable_variables():
    return [v for v in S.variables.values() if v.trainable]


def global_variables():
    return list(S.variables.values())


def add_to_collection(name, value):
    S.collections.setdefault(name, []).append(value)


def get_collection(name, scope=None):
    return list(S.collections.get(name, []))


class GraphKeys:
    GLOBAL_VARIABLES = "variables"
    TRAINABLE_VARIABLES = "trainable_variables"
    UPDATE_OPS = "update_ops"


# ----------------------------------------------------------------------------------------------------------------------
# ops
# ----------------------------------------------------------------------------------------------------------------------
def _op(fn, *inputs, **kw):
    return Tensor(fn, list(inputs), kw.get("name"))


def reshape(t, shape, name=None):
    if isinstance(shape, Shape):
        shape = shape.as_list()
    shp = [builtins_int(s) for s in shape] if not isinstance(shape, Tensor) else None
    if shp is None:
        raise NotImplementedError("dynamic reshape")
    return _op(lambda a: a.reshape(shp), t)


def transpose(t, perm=None, name=None):
    return _op(lambda a: a.permute(*perm) if perm is not None else a.t(), t)


def concat(values, axis, name=None):
    return Tensor(lambda *a: torch.cat([x.to(DT) if any(y.dtype.is_floating_point for y in a) and not x.dtype.is_floating_point else x for x in a], dim=axis), list(values))


def split(value, num_or_size_splits, axis=0, name=None):
    n = num_or_size_splits
    return [Tensor((lambda k: (lambda a: torch.chunk(a, n, dim=axis)[k]))(k), [value]) for k in range(n)]


def stack(values, axis=0, name=None):
    return Tensor(lambda *a: torch.stack(a, dim=axis), list(values))


def expand_dims(t, axis, name=None):
    return _op(lambda a: a.unsqueeze(axis), t)


def cast(t, dtype, name=None):
    return _op(lambda a: a.to(_dtype(dtype)), t)


def add_n(ts, name=None):
    def fn(*a):
        out = a[0]
        for x in a[1:]:
            out = out + x
        return out
    return Tensor(fn, list(ts))


def _red(f):
    def op(t, axis=None, keep_dims=False, name=None, reduction_indices=None, keepdims=None):
        ax = axis if axis is not None else reduction_indices
        kd = keep_dims or bool(keepdims)
        if ax is None:
            return _op(lambda a: f(a), t)
        ax = tuple(ax) if isinstance(ax, (list, tuple)) else (ax,)
        return _op(lambda a: f(a, dim=ax, keepdim=kd), t)
    return op


reduce_mean = _red(torch.mean)
reduce_sum = _red(torch.sum)
matmul = lambda a, b, name=None: _op(lambda x, y: x @ y, a, b)
tanh = lambda t, name=None: _op(torch.tanh, t)
sqrt = lambda t, name=None: _op(torch.sqrt, t)
square = lambda t, name=None: _op(lambda a: a * a, t)
log = lambda t, name=None: _op(torch.log, t)
maximum = lambda a, b, name=None: _op(lambda x, y: torch.maximum(x.to(DT), y.to(DT)), a, b)
minimum = lambda a, b, name=None: _op(lambda x, y: torch.minimum(x.to(DT), y.to(DT)), a, b)
less = lambda a, b, name=None: _op(lambda x, y: x < y, a, b)
where = lambda c, a, b, name=None: _op(lambda z, x, y: torch.where(z, x.to(DT), y.to(DT)), c, a, b)
ones_like = lambda t, name=None: _op(torch.ones_like, t)
zeros_like = lambda t, name=None: _op(torch.zeros_like, t)
one_hot = lambda idx, depth, name=None: _op(lambda i: F.one_hot(i.long(), depth).to(DT), idx)
tensordot = lambda a, b, axes, name=None: _op(lambda x, y: torch.tensordot(x, y, dims=(list(axes[0]), list(axes[1]))), a, b)
no_op = lambda name=None: constant(0)
tile = lambda t, multiples, name=None: _op(lambda a: a.repeat(*multiples), t)
clip_by_value = lambda t, lo, hi, name=None: _op(lambda a: a.clamp(lo, hi), t)
sigmoid = lambda t, name=None: _op(torch.sigmoid, t)


def zeros(shape, dtype=None, name=None):
    return constant(np.zeros([builtins_int(s) for s in shape], np.float32))


def ones(shape, dtype=None, name=None):
    return constant(np.ones([builtins_int(s) for s in shape], np.float32))


def depth_to_space(t, block_size, name=None):
    b = block_size

    def fn(a):
        n, h, w, c = a.shape
        return a.reshape(n, h, w, b, b, c // (b * b)).permute(0, 1, 3, 2, 4, 5).reshape(n, h * b, w * b, c // (b * b))
    return _op(fn, t)


def _random(kind, shape, draw):
    shp = tuple(builtins_int(s) for s in shape)

    def fn():
        a = draw(shp)
        S.draws.append((kind, a.copy()))
        return torch.as_tensor(a).to(DT)
    t = Tensor(fn, [])
    t._shape = shp
    t._probe_val = torch.zeros(shp, dtype=DT)
    t.volatile = True
    return t


def random_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):
    return _random("random_normal", shape, lambda s: (mean + stddev * S.rng.normal(size=s)).astype(np.float32))


def random_uniform(shape, minval=0.0, maxval=1.0, dtype=None, seed=None, name=None):
    return _random("random_uniform", shape, lambda s: S.rng.uniform(minval, maxval, size=s).astype(np.float32))


def while_loop(cond, body, loop_vars, **kw):
    """Unrolled at graph-construction time: the trip count must be static (sn.py: i < num_iters with a python int)."""
    vs = tuple(loop_vars)
    for _ in range(1000):
        c = cond(*vs)
        cv = _static_value(c)
        if not bool(cv):
            return vs
        vs = tuple(body(*vs))
    raise RuntimeError("while_loop did not terminate")


def _static_value(t):
    if not isinstance(t, Tensor):
        return t
    if t._const is not None:
        return t._const
    if isinstance(t, (Placeholder, Variable)) or getattr(t, "volatile", False):
        raise ValueError("not a compile-time constant")
    return t.fn(*[_static_value(i) for i in t.inputs])


def gradients(ys, xs, name=None):
    ys = ys if isinstance(ys, (list, tuple)) else [ys]
    assert len(ys) == 1
    return _grad_tensors(ys[0], list(xs))


def _grad_tensors(loss, xs):
    def gfn(l, *vals):
        gs = torch.autograd.grad(l, list(vals), retain_graph=True, allow_unused=True)
        return tuple(torch.zeros_like(v) if g is None else g for g, v in zip(gs, vals))
    node = Tensor(gfn, [loss] + xs)
    node._shape = ()
    node._probe_val = tuple(x._probe() for x in xs)
    outs = []
    for k, x in enumerate(xs):
        t = Tensor((lambda k: (lambda tup: tup[k]))(k), [node])
        t._shape = x._static()
        t._probe_val = x._probe()
        outs.append(t)
    return outs


# ---- nn
def _same_pad(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return tot // 2, tot - tot // 2


def _conv2d(x, w, strides, padding):
    sh, sw = strides[1], strides[2]
    kh, kw = w.shape[0], w.shape[1]
    xn = x.permute(0, 3, 1, 2)
    if padding == "SAME":
        pt, pb = _same_pad(x.shape[1], kh, sh)
        pl, pr = _same_pad(x.shape[2], kw, sw)
        xn = F.pad(xn, (pl, pr, pt, pb))
    return F.conv2d(xn, w.permute(3, 2, 0, 1), stride=(sh, sw)).permute(0, 2, 3, 1)


def _conv2d_transpose(x, w, output_shape, strides, padding):
    """conv2d_transpose(value, filter[kh,kw,out,in], output_shape) = the gradient of conv2d(SAME) w.r.t. its input of shape
    output_shape: the un-padded transposed convolution cropped at the forward convolution's leading pad."""
    n, oh, ow, oc = [builtins_int(s) for s in output_shape]
    sh, sw = strides[1], strides[2]
    kh, kw = w.shape[0], w.shape[1]
    full = F.conv_transpose2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), stride=(sh, sw))
    pt, _ = _same_pad(oh, kh, sh) if padding == "SAME" else (0, 0)
    pl, _ = _same_pad(ow, kw, sw) if padding == "SAME" else (0, 0)
    full = F.pad(full, (0, max(0, pl + ow - full.shape[3]), 0, max(0, pt + oh - full.shape[2])))
    return full[:, :, pt:pt + oh, pl:pl + ow].permute(0, 2, 3, 1)


nn = types.SimpleNamespace(
    relu=lambda t, name=None: _op(torch.relu, t),
    sigmoid=lambda t, name=None: _op(torch.sigmoid, t),
    tanh=lambda t, name=None: _op(torch.tanh, t),
    softplus=lambda t, name=None: _op(F.softplus, t),
    softmax=lambda t, dim=-1, axis=None, name=None: _op(lambda a: torch.softmax(a, dim=dim if axis is None else axis), t),
    conv2d=lambda input, filter, strides, padding, data_format="NHWC", name=None, use_cudnn_on_gpu=True:
        _op(lambda a, w: _conv2d(a, w, strides, padding), input, filter),
    conv2d_transpose=lambda value, filter, output_shape, strides, padding="SAME", data_format="NHWC", name=None:
        _op(lambda a, w: _conv2d_transpose(a, w, output_shape, strides, padding), value, filter),
    bias_add=lambda value, bias, data_format="NHWC", name=None: _op(lambda a, b: a + b, value, bias),
    embedding_lookup=lambda params, ids, name=None: _op(lambda p, i: p[i.long()], params, ids),
    moments=lambda x, axes, keep_dims=False, name=None: (
        _op(lambda a: a.mean(dim=tuple(axes), keepdim=keep_dims), x),
        _op(lambda a: ((a - a.mean(dim=tuple(axes), keepdim=True)) ** 2).mean(dim=tuple(axes), keepdim=keep_dims), x)),
    batch_normalization=lambda x, mean, variance, offset, scale, variance_epsilon, name=None:
        _op(lambda a, m, v, o, s: (lambda inv: a * inv + (o - m * inv))(torch.rsqrt(v + variance_epsilon) * s), x, mean, variance, offset, scale),
    sigmoid_cross_entropy_with_logits=lambda logits=None, labels=None, targets=None, name=None:
        _op(lambda x, z: torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-torch.abs(x))), logits, labels if labels is not None else targets),
    avg_pool=None, softsign=lambda t: _op(F.softsign, t),
)


# ----------------------------------------------------------------------------------------------------------------------
# session
# ----------------------------------------------------------------------------------------------------------------------
class Session:
    _default = None
    _pending = []

    def __init__(self, config=None, graph=None):
        self.graph = object()
        Session._default = self

    def __enter__(self):
        Session._default = self
        return self

    def __exit__(self, *a):
        return False

    def as_default(self):
        return self

    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        fl = [fetches] if single else list(fetches)
        feed = {}
        for k, v in (feed_dict or {}).items():
            feed[k] = _tt(v).to(k._dt) if isinstance(k, Placeholder) else _tt(v)
        cache = {}
        S.draws = []
        Session._pending = []

        def ev(t):
            if not isinstance(t, Tensor):
                return t
            key = id(t)
            if key in cache:
                return cache[key]
            if isinstance(t, Placeholder):
                if t not in feed:
                    raise ValueError("placeholder %s was not fed" % t.name)
                val = feed[t]
            elif isinstance(t, Variable):
                val = t.value
            else:
                for d in t.deps:
                    ev(d)
                val = t.fn(*[ev(i) for i in t.inputs])
            cache[key] = val
            return val
        outs = []
        for f in fl:
            if f is None or isinstance(f, (_NoOp, str)):
                outs.append(None)
                continue
            v = ev(f)
            outs.append(v.detach().numpy().copy() if isinstance(v, torch.Tensor) else None)
        # assignments take effect when the run is over (every read of a variable inside one run sees the value of its start)
        with torch.no_grad():
            for var, val in Session._pending:
                var.value = val.clone().detach().to(var.value.dtype).requires_grad_(var.value.dtype.is_floating_point)
        rec = {"feeds": {(k.name or "ph%d" % i): np.asarray(v) for i, (k, v) in enumerate((feed_dict or {}).items())},
               "draws": list(S.draws), "assigned": [var.name for var, _ in Session._pending]}
        S.run_log.append(rec)
        return outs[0] if single else outs


class _NoOp:
    def run(self, *a, **k):
        return None


def global_variables_initializer():
    return _NoOp()


class ConfigProto:
    def __init__(self, **kw):
        self.gpu_options = types.SimpleNamespace(allow_growth=False)


class GPUOptions:
    def __init__(self, **kw):
        pass


# ----------------------------------------------------------------------------------------------------------------------
# optimiser (tf.train.AdamOptimizer: ApplyAdam)
# ----------------------------------------------------------------------------------------------------------------------
class AdamOptimizer:
    _count = 0

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, name="Adam"):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon
        self.t = 0
        self.m, self.v = {}, {}
        self.index = AdamOptimizer._count
        AdamOptimizer._count += 1
        S.adam_slots[self.index] = self

    def compute_gradients(self, loss, var_list=None):
        vs = list(var_list) if var_list is not None else trainable_variables()
        return list(zip(_grad_tensors(loss, vs), vs))

    def apply_gradients(self, grads_and_vars, global_step=None, name=None):
        gv = [(g, v) for g, v in grads_and_vars if g is not None]
        opt = self

        def fn(lr, *grads):
            opt.t += 1
            lr = builtins.float(lr)
            lr_t = lr * np.sqrt(1.0 - opt.b2 ** opt.t) / (1.0 - opt.b1 ** opt.t)
            with torch.no_grad():
                for g, (_, var) in zip(grads, gv):
                    m = opt.m.get(var.name, torch.zeros_like(var.value))
                    v = opt.v.get(var.name, torch.zeros_like(var.value))
                    m = m + (g - m) * (1.0 - opt.b1)
                    v = v + (g * g - v) * (1.0 - opt.b2)
                    opt.m[var.name], opt.v[var.name] = m, v
                    new = var.value.detach() - (m * lr_t) / (torch.sqrt(v) + opt.eps)
                    if var.constraint is not None:
                        new = _static_value(var.constraint(constant(new)))
                    Session._pending.append((var, new))
            return torch.zeros(())
        return Tensor(fn, [_wrap(self.lr)] + [g for g, _ in gv], name="apply_adam_%d" % self.index)

    def minimize(self, loss, var_list=None, global_step=None):
        return self.apply_gradients(self.compute_gradients(loss, var_list))


class _Saver:
    def __init__(self, *a, **kw):
        self.saved = []

    def save(self, sess, path, global_step=None):
        self.saved.append((path, global_step))
        return path

    def restore(self, sess, path):
        raise RuntimeError("no checkpoints in the emulation")


train = types.SimpleNamespace(AdamOptimizer=AdamOptimizer, Saver=_Saver, latest_checkpoint=lambda d: None,
                              NewCheckpointReader=None)


class _Writer:
    def __init__(self, *a, **kw):
        pass

    def add_summary(self, *a, **kw):
        pass

    def flush(self):
        pass

    def close(self):
        pass


summary = types.SimpleNamespace(scalar=lambda *a, **k: None, histogram=lambda *a, **k: None, image=lambda *a, **k: None,
                                merge_all=lambda: None, merge=lambda *a, **k: None, FileWriter=_Writer)


# ----------------------------------------------------------------------------------------------------------------------
# flags / app / misc
# ----------------------------------------------------------------------------------------------------------------------
class _Flags:
    """tf.app.flags: DEFINE_* record defaults; values come from ``overrides`` (set by the harness before the import)."""
    overrides = {}

    def __init__(self):
        object.__setattr__(self, "_vals", {})

    def _define(self, name, default, help=None):
        self._vals[name] = _Flags.overrides.get(name, default)

    def __getattr__(self, name):
        if name.endswith("__flags"):
            return dict(self._vals)
        try:
            return self._vals[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self._vals[name] = value


class _FlagsModule:
    def __init__(self):
        self.FLAGS = _Flags()
        for n in ("string", "integer", "float", "boolean", "bool", "list"):
            setattr(self, "DEFINE_" + n, self.FLAGS._define)


flags = _FlagsModule()
app = types.SimpleNamespace(flags=flags, run=lambda main=None, argv=None: None)


class StopReference(Exception):
    """Raised where the reference would leave the part of its program the emulation covers (e.g. loading a frozen GraphDef)."""


class _GFile:
    def __init__(self, *a, **kw):
        raise StopReference("tf.gfile.GFile")


gfile = types.SimpleNamespace(GFile=_GFile)


def _contrib_batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, updates_collections="update_ops", is_training=True,
                        scope=None, reuse=None, fused=None, zero_debias_moving_mean=False, **kw):
    """tf.contrib.layers.batch_norm as mnist/ops.py:38-44 calls it (decay, updates_collections=None, epsilon, scale=True, is_training,
    scope): variables beta / gamma / moving_mean / moving_variance under ``scope``; training mode normalises with the biased batch
    variance and moves the averages in place (control dependency of the output) -- the moving variance with the UNBIASED batch
    variance, as the fused kernel does; inference mode uses the moving statistics."""
    assert updates_collections is None and center and scale
    c = builtins_int(inputs.get_shape()[-1])
    with variable_scope(scope, reuse=reuse):
        beta = get_variable("beta", [c], initializer=constant_initializer(0.0))
        gamma = get_variable("gamma", [c], initializer=constant_initializer(1.0))
        mm = get_variable("moving_mean", [c], initializer=constant_initializer(0.0), trainable=False)
        mv = get_variable("moving_variance", [c], initializer=constant_initializer(1.0), trainable=False)
    if not is_training:
        return _op(lambda x, g, b, m, v: (x - m) * torch.rsqrt(v + epsilon) * g + b, inputs, gamma, beta, mm, mv)
    nd = len(inputs.get_shape())
    axes = tuple(range(nd - 1))
    mean = _op(lambda x: x.mean(dim=axes), inputs)
    var = _op(lambda x: ((x - x.mean(dim=axes, keepdim=True)) ** 2).mean(dim=axes), inputs)

    def unbiased(x, v):
        n = 1
        for a in axes:
            n *= x.shape[a]
        return v * (n / max(n - 1, 1))
    uvar = _op(unbiased, inputs, var)
    up_m = mm.assign(_op(lambda m, b: (m - (m - b) * (1.0 - decay)).detach(), mm, mean))
    up_v = mv.assign(_op(lambda m, b: (m - (m - b) * (1.0 - decay)).detach(), mv, uvar))
    with control_dependencies([up_m, up_v]):
        return _op(lambda x, m, v, g, b: (x - m) * torch.rsqrt(v + epsilon) * g + b, inputs, mean, var, gamma, beta)
contrib = types.SimpleNamespace(layers=types.SimpleNamespace(variance_scaling_initializer=lambda **kw: glorot_uniform_initializer(),
                                                            batch_norm=_contrib_batch_norm, layer_norm=None, instance_norm=None),
                                slim=types.SimpleNamespace(model_analyzer=types.SimpleNamespace(analyze_vars=lambda *a, **k: None)))
image = types.SimpleNamespace()
logging = types.SimpleNamespace(info=lambda *a, **k: None, warn=lambda *a, **k: None)
__version__ = "1.5.0-emulated"
