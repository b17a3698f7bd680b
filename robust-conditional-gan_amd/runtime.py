"""Device runtime for the RCGAN engine: context, stream, arena allocator, device tensors.

PyTorch-ROCm is used only as plumbing: it owns the HIP stream, allocates the few large buffers
(parameter/gradient/optimiser slabs, the activation arena, the kernel workspace), moves host data and
bootstraps RCCL.  Every arithmetic op goes through the C ABI in ``_lib`` to the hand-written gfx950
kernels.  Layout in HBM (one process per GPU, sized for 288 GB):

  params   : one flat fp32 slab per optimiser group (Generator / Discriminator / confusion_logits),
             each variable 256-byte aligned; gradients and Adam m, v are slabs of the same shape so the
             optimiser and the RCCL all-reduce are single launches over contiguous memory.
  state    : SN ``u`` vectors, BN moving statistics (non-trainable, checkpointed, never all-reduced).
  arena    : bump allocator for activations / activation gradients / per-step scratch; reset at the
             start of every step so captured hipGraphs see identical addresses on replay.
  workspace: split-K slabs of the filter-gradient kernels and the upsample-folded dgrad scratch.
"""
import ctypes as C

import os

import numpy as np
import torch

from . import _lib as L

ALIGN = 256


def _np_dtype(code):
    return np.float32 if code == L.F32 else np.uint16


class DT:
    """A device tensor: raw pointer + shape + dtype code.  ``base`` keeps the owning torch storage alive."""
    __slots__ = ("ptr", "shape", "dtype", "base", "grad", "req", "name", "frozen", "group", "tile_stats", "pooled", "pool_grad", "concat_src", "concat_labels")

    def __init__(self, ptr, shape, dtype, base=None, name=None):
        self.ptr = int(ptr)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = dtype
        self.base = base
        self.grad = None
        self.req = False
        self.name = name
        self.frozen = False        # a gradient buffer that must not be written any more (ops.grad_of copies on write)
        self.group = None          # the ParamGroup a parameter lives in
        self.tile_stats = None     # (conv desc, per-tile column sums) left by the producing convolution (ops.conv2d(want_stats=True))
        self.pooled = None         # (features, activation kind) the producer left beside the tensor (ops.d_trunk(pool=...))
        self.pool_grad = None      # gradient of those features, handed back by their consumer (ops.proj_head)
        self.concat_src = None     # (x, channels of x) when this tensor is conv_cond_concat(x, labels) (ops.concat_channels)
        self.concat_labels = None  # ... and the label rows yb [n, c2] (fp32) it appended

    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def itemsize(self):
        if self.dtype == "u8":
            return 1
        return 4 if self.dtype in (L.F32, "i32") else 2

    @property
    def nbytes(self):
        return self.size * self.itemsize

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        shape = list(shape)
        if -1 in shape:
            known = 1
            for s in shape:
                if s != -1:
                    known *= s
            shape[shape.index(-1)] = self.size // known
        t = DT(self.ptr, shape, self.dtype, self.base, self.name)
        assert t.size == self.size, (self.shape, shape)
        return t

    def rows(self, lo, hi):
        """Contiguous leading-dimension slice (a view)."""
        stride = self.size // self.shape[0]
        return DT(self.ptr + lo * stride * self.itemsize, (hi - lo,) + self.shape[1:], self.dtype, self.base, self.name)


class Arena:
    def __init__(self, nbytes, device):
        self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        self.base = self.buf.data_ptr()
        self.cap = int(nbytes)
        self.off = 0
        self.peak = 0

    def reset(self):
        self.off = 0

    def alloc(self, nbytes):
        off = (self.off + ALIGN - 1) // ALIGN * ALIGN
        if off + nbytes > self.cap:
            raise MemoryError("activation arena exhausted: need %d more bytes (capacity %d); raise arena_bytes"
                              % (off + nbytes - self.cap, self.cap))
        self.off = off + int(nbytes)
        self.peak = max(self.peak, self.off)
        return self.base + off


class Context:
    """One per process / GPU.  Wraps rcgan_ctx, the stream, the arena and the workspace."""

    def __init__(self, device=0, dtype="bf16", arena_bytes=6 << 30, ws_bytes=1 << 30):
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the RCGAN engine has no CPU fallback")
        if dtype in ("f16", "fp16", L.F16):
            self.act_dtype, self.lib = L.F16, L.load("f16")       # the fp16 build of the same kernels
        else:
            self.act_dtype, self.lib = (L.BF16 if dtype in ("bf16", L.BF16) else L.F32), L.load()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        h = C.c_void_p()
        rc = self.lib.rcgan_create(C.byref(h), device, C.c_void_p(self.stream.cuda_stream))
        if rc != 0:
            raise L.RcganError(rc, "rcgan_create failed")
        self.h = h
        self.arena = Arena(arena_bytes, self.device)
        self.ws = torch.empty(int(ws_bytes), dtype=torch.uint8, device=self.device)
        self.ws_ptr, self.ws_bytes = self.ws.data_ptr(), int(ws_bytes)
        # opt-in (RCGAN_OVERLAP=1): filter gradients launched on a side stream next to their data gradient, with their own
        # workspace.  Measured on MI355X: 12.1 ms/iteration with the fork/join in the captured graph vs 11.4 without --
        # the graph's cross-stream dependencies cost more than the idle CUs they fill -- so it stays off.
        self.epoch = 0             # bumped by new_step(): ParamGroup.zero_grad stamps it (ops.spectral_norm_batch)
        self.overlap = os.environ.get("RCGAN_OVERLAP", "0") == "1"
        self.ws2 = torch.empty(int(ws_bytes), dtype=torch.uint8, device=self.device) if self.overlap else None
        self.ws2_ptr = self.ws2.data_ptr() if self.overlap else 0
        self.tape = []
        self.sn_partial = []        # spectral-norm backward closures of this step that take a parameter-name predicate (RCGAN_DP_OVERLAP)
        self.pending_wgrads = []
        self.group_wgrads = os.environ.get("RCGAN_GROUP_WGRAD", "1") == "1"     # see defer_wgrad
        self.recording = True
        self.capturing = False      # between graph_begin() and graph_end(): launches are recorded into a hipGraph
        self._keep = []
        self.check(self.lib.rcgan_selftest(self.h))
        self.uses_tr_read = self.lib.rcgan_query(self.h, L.QUERY_TR_READ)

    # ------------------------------------------------------------------ errors
    def check(self, rc):
        if rc != 0:
            raise L.RcganError(rc, self.lib.rcgan_last_error(self.h).decode())

    def close(self):
        for c in getattr(self, "also_close", []):       # (contexts that exist only beside this one: the generator-forward stream)
            c.close()
        self.also_close = []
        if self.h is not None:
            self.lib.rcgan_destroy(self.h)
            self.h = None

    # ------------------------------------------------------------------ memory
    def empty(self, shape, dtype=None, name=None):
        dtype = self.act_dtype if dtype is None else dtype
        n = 1
        for s in shape:
            n *= int(s)
        isz = 4 if dtype in (L.F32, "i32") else 2
        return DT(self.arena.alloc(max(n, 1) * isz), shape, dtype, self.arena.buf, name)

    def persistent(self, shape, dtype=L.F32, fill=None):
        """A buffer outside the arena (survives arena resets)."""
        tdt = {L.F32: torch.float32, L.BF16: torch.bfloat16, L.F16: torch.float16, "i32": torch.int32}[dtype]
        t = torch.empty(tuple(int(s) for s in shape) or (1,), dtype=tdt, device=self.device)
        if fill is not None:
            with torch.cuda.stream(self.stream):
                t.fill_(fill)
        self._keep.append(t)
        return DT(t.data_ptr(), shape, dtype, t)

    def view(self, t):
        """torch tensor aliasing a DT (for uploads, downloads and RCCL)."""
        tdt = {L.F32: torch.float32, L.BF16: torch.bfloat16, L.F16: torch.float16, "i32": torch.int32}[t.dtype]
        if isinstance(t.base, torch.Tensor):
            base = t.base
            off = t.ptr - base.data_ptr()
            raw = base.view(torch.uint8).reshape(-1)[off:off + t.nbytes]
            return raw.view(tdt).reshape(t.shape if t.shape else (1,))
        raise ValueError("DT has no torch base")

    def upload(self, arr, dtype=None, out=None):
        """numpy -> device (into ``out`` or a fresh arena tensor); float arrays go to the activation dtype
        unless ``dtype`` says otherwise, integer arrays to int32."""
        arr = np.asarray(arr)
        if arr.dtype.kind in "iub":
            dtype = "i32"
            src = torch.from_numpy(np.ascontiguousarray(arr.astype(np.int32)))
        else:
            dtype = self.act_dtype if dtype is None else dtype
            src = torch.from_numpy(np.ascontiguousarray(arr.astype(np.float32)))
        if out is None:
            out = self.empty(arr.shape, dtype)
        with torch.cuda.stream(self.stream):
            self.view(out).copy_(src.reshape(out.shape if out.shape else (1,)), non_blocking=False)
        return out

    def download(self, t):
        with torch.cuda.stream(self.stream):
            v = self.view(t).float() if t.dtype != "i32" else self.view(t)
            out = v.cpu()
        self.stream.synchronize()
        return out.numpy().reshape(t.shape)

    def sync(self):
        self.check(self.lib.rcgan_stream_sync(self.h))

    def zeros(self, shape, dtype=None):
        t = self.empty(shape, dtype)
        with torch.cuda.stream(self.stream):
            self.view(t).zero_()
        return t

    # ------------------------------------------------------------------ tape
    def record(self, fn):
        if self.recording:
            self.tape.append(fn)

    def backward(self):
        for fn in reversed(self.tape):
            fn()
        self.tape = []
        self.flush_wgrads()

    # ------------------------------------------------------------------ grouped filter gradients
    def defer_wgrad(self, desc, x, dy, dw, dbias):
        """Queue one conv layer's filter gradient; flush_wgrads() computes all queued ones in one grouped call.
        The caller guarantees x and dy stay untouched until then."""
        self.pending_wgrads.append((desc, x, dy, dw, dbias))

    def flush_wgrads(self):
        """All queued filter gradients as one grouped call; the projection head's deferred parameter gradients (ops.proj_head) ride
        in its launch, and whatever of them is still pending afterwards is launched on the spot."""
        pend = self.pending_wgrads
        if pend:
            self.pending_wgrads = []
            n = len(pend)
            descs = (L.ConvDesc * n)(*[p[0] for p in pend])
            arr = lambda k: (C.c_void_p * n)(*[(p[k].ptr if p[k] is not None else None) for p in pend])
            xs, dys, dws, dbs = arr(1), arr(2), arr(3), arr(4)
            self.check(self.lib.rcgan_conv2d_bwd_weight_group(self.h, n, descs, xs, dys, dws, dbs, 1, C.c_void_p(self.ws_ptr), self.ws_bytes))
        self.check(self.lib.rcgan_head_flush(self.h))

    def new_step(self):
        self.check(self.lib.rcgan_head_flush(self.h))      # (a head whose backward pass never ran: its buffers are still valid here)
        self.tape = []
        self.sn_partial = []
        self.pending_wgrads = []
        self.arena.reset()
        self.epoch += 1

    # ------------------------------------------------------------------ graphs
    def reserve_scratch(self, nbytes):
        """Grow the context's hidden scratch (split reductions, narrow data gradients of the fp32 path) BEFORE a capture: inside one an
        entry point that has to grow it fails (include/rcgan_hip.h, capture contract)."""
        self.check(self.lib.rcgan_reserve_scratch(self.h, C.c_size_t(int(nbytes))))

    def graph_begin(self):
        self.check(self.lib.rcgan_graph_begin(self.h))
        self.capturing = True

    def graph_end(self):
        gid = C.c_int(-1)
        self.capturing = False
        self.check(self.lib.rcgan_graph_end(self.h, C.byref(gid)))
        return gid.value

    def graph_abort(self):
        self.capturing = False
        self.check(self.lib.rcgan_graph_abort(self.h))

    def graph_launch(self, gid):
        self.check(self.lib.rcgan_graph_launch(self.h, gid))

    def event_record(self, slot):
        self.check(self.lib.rcgan_event_record(self.h, slot))

    def event_elapsed_ms(self, a, b):
        ms = C.c_float(0)
        self.check(self.lib.rcgan_event_elapsed_ms(self.h, a, b, C.byref(ms)))
        return ms.value


class ParamGroup:
    """Flat fp32 slabs (value, grad, Adam m, Adam v) for one optimiser group."""

    def __init__(self, ctx, specs, sn_scratch=False):
        """specs: list of (name, shape, init ndarray).  sn_scratch: the gradient buffer carries a second slab for the
        d/dW_bar scratch of spectrally normalised weights (same offsets), so that ONE fill zeroes the gradients, that scratch
        and the group's loss scalars at the start of a step."""
        self.ctx = ctx
        self.names, self.offsets, self.shapes = [], {}, {}
        off = 0
        for name, shape, _ in specs:
            off = (off + 63) // 64 * 64
            self.names.append(name)
            self.offsets[name] = off
            self.shapes[name] = tuple(shape)
            n = 1
            for s in shape:
                n *= int(s)
            off += n
        self.count = max((off + 63) // 64 * 64, 64)
        dev = ctx.device
        self.value = torch.zeros(self.count, dtype=torch.float32, device=dev)
        self.sn_scratch = bool(sn_scratch)
        self._extra = self.count if sn_scratch else 0
        # [gradients (count) | d/dW_bar scratch (count, optional) | 64 scalars]: zero_grad() clears all of it in one launch
        self.gradbuf = torch.zeros(self.count + self._extra + 64, dtype=torch.float32, device=dev)
        self.grad = self.gradbuf[:self.count]
        self.zero_epoch = -1       # Context.epoch of the last zero_grad()
        self.m = torch.zeros(self.count, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.count, dtype=torch.float32, device=dev)
        self._lr, self._t = 0.0, 0.0      # {lr, t} of the next Adam launch (set_hyper)
        self.t = 0
        self.hyper = None       # DEVICE float[2] = {lr, t} of the captured Adam launch (set_hyper_device)
        self.t_dev = None       # dynamic loss scaling: DEVICE float[1], the number of APPLIED updates (adam_dyn)
        self.version = 0        # bumped by every write to the values (set / adam): keys caches of derived layouts
        host = np.zeros(self.count, np.float32)
        for name, shape, init in specs:
            o = self.offsets[name]
            a = np.asarray(init, np.float32).reshape(-1)
            host[o:o + a.size] = a
        self.value.copy_(torch.from_numpy(host))
        torch.cuda.synchronize()

    @classmethod
    def alias(cls, ctx, other):
        """The same slabs seen from another context (another stream / arena): parameter lookups only -- the optimiser, the
        zero-fills and the version counter stay with the owning group."""
        g = cls.__new__(cls)
        g.__dict__.update(other.__dict__)
        g.ctx = ctx
        return g

    def _on_stream(self):
        return torch.cuda.stream(self.ctx.stream)

    def param(self, name):
        o = self.offsets[name] * 4
        p = DT(self.value.data_ptr() + o, self.shapes[name], L.F32, self.value, name)
        p.grad = DT(self.grad.data_ptr() + o, self.shapes[name], L.F32, self.gradbuf, name + ":grad")
        p.group = self
        return p

    def dwbar(self, name):
        """The d/dW_bar scratch of a spectrally normalised parameter (zeroed by zero_grad)."""
        assert self.sn_scratch
        o = (self.count + self.offsets[name]) * 4
        return DT(self.gradbuf.data_ptr() + o, self.shapes[name], L.F32, self.gradbuf, name + ":dwbar")

    def scalar(self, i):
        """One of 64 fp32 scalars behind the gradients (loss accumulators): zeroed by zero_grad with everything else."""
        assert 0 <= i < 64
        return DT(self.gradbuf.data_ptr() + (self.count + self._extra + i) * 4, (1,), L.F32, self.gradbuf, "scalar%d" % i)

    def get(self, name, which="value"):
        o = self.offsets[name]
        n = int(np.prod(self.shapes[name])) if self.shapes[name] else 1
        with self._on_stream():
            out = getattr(self, which)[o:o + n].detach().cpu()
        self.ctx.stream.synchronize()
        return out.numpy().reshape(self.shapes[name])

    def set(self, name, arr, which="value"):
        o = self.offsets[name]
        a = torch.from_numpy(np.ascontiguousarray(np.asarray(arr, np.float32).reshape(-1)))
        with self._on_stream():
            getattr(self, which)[o:o + a.numel()].copy_(a)
        if which == "value":
            self.version += 1

    def zero_grad(self, defer=False):
        """Clear gradients, d/dW_bar scratch and the loss scalars.  defer: do not launch -- return (pointer, float count) for a
        launch of this step that zero-fills on the side (ops.prepare_batch(inputs=...)) BEFORE anything accumulates into them."""
        c = self.ctx
        self.zero_epoch = c.epoch
        if defer:
            return self.gradbuf.data_ptr(), self.gradbuf.numel()
        c.check(c.lib.rcgan_fill_f32(c.h, self.gradbuf.numel(), self.gradbuf.data_ptr(), 0.0))

    def set_hyper_device(self, lr, t):
        """{lr, t} of the next CAPTURED Adam launch (adam_captured): written to device memory by a one-thread launch on the stream
        in front of the step's graph -- no host-to-device copy, and a replayed graph reads fresh values."""
        c = self.ctx
        if self.hyper is None:
            # (no fill: the launch below writes both words.  A torch.zeros here ran its fill on torch's CURRENT stream, unordered against the
            # context's stream -- when the fill lost the race against the first rcgan_set2_f32 the group's first captured-form Adam read
            # {lr, t} = {0, 0} and divided 0 by 0: the generator's parameters became NaN in ~3 % of the data-parallel runs that reach that
            # Adam launch by launch, found through tests/test_gpu_dp.py's capture-fallback case, round 4)
            with torch.cuda.stream(c.stream):
                self.hyper = torch.empty(2, dtype=torch.float32, device=c.device)
        if float(t) >= float(1 << 24):
            raise OverflowError("Adam step counter %d is not exactly representable as fp32" % int(t))
        c.check(c.lib.rcgan_set2_f32(c.h, self.hyper.data_ptr(), float(lr), float(t)))

    def adam_captured(self, beta1, beta2, eps=1e-8, clip=0.0, grad_scale=1.0):
        """The capturable Adam launch: {lr, t} read from device memory (set_hyper_device), rcgan_adam_tf."""
        c = self.ctx
        assert self.hyper is not None, "set_hyper_device first"
        c.check(c.lib.rcgan_adam_tf(c.h, self.count, self.value.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                    self.hyper.data_ptr(), beta1, beta2, eps, clip, grad_scale))

    def finite_check(self, ls_state):
        """Raise ls_state's non-finite flag if this group's gradient slab holds an inf / nan (rcgan_grad_finite_check)."""
        c = self.ctx
        if self.t_dev is None:
            with torch.cuda.stream(c.stream):       # (the fill on the stream that reads it)
                self.t_dev = torch.full((1,), float(max(self.t - 1, 0)), dtype=torch.float32, device=c.device)
        c.check(c.lib.rcgan_grad_finite_check(c.h, self.count, self.grad.data_ptr(), ls_state.data_ptr()))

    def adam_dyn(self, ls_state, beta1, beta2, eps=1e-8, clip=0.0, grad_scale=1.0):
        """Adam under dynamic loss scaling (rcgan_adam_tf_dyn): skipped when ls_state's non-finite flag is raised, the gradient divided
        by the current scale, t = applied updates + 1 read from device memory.  lr from set_hyper."""
        c = self.ctx
        if c.capturing:
            raise RuntimeError("ParamGroup.adam_dyn() bakes lr into the launch and must not be captured into a hipGraph")
        c.check(c.lib.rcgan_adam_tf_dyn(c.h, self.count, self.value.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                        self._lr, self.t_dev.data_ptr(), beta1, beta2, eps, clip, grad_scale, ls_state.data_ptr()))
        self.version += 1

    def steps_applied(self):
        """Optimiser updates actually applied: the device's count under dynamic loss scaling (skipped steps excluded), else t."""
        if self.t_dev is None:
            return self.t
        self.ctx.sync()
        return int(round(float(self.t_dev.cpu()[0])))

    def set_hyper(self, lr, t):
        """{lr, t} of the next Adam launch.  Adam is launched eagerly behind the step's captured graph and the all-reduce, so the
        two values travel as kernel arguments (rcgan_adam_tf_host): no host-to-device copy per optimiser step."""
        self._lr, self._t = float(lr), float(t)

    def adam(self, beta1, beta2, eps=1e-8, clip=0.0, grad_scale=1.0, lo=0, hi=None):
        """Eager TF-form Adam launch with {lr, t} as kernel arguments.  NOT capturable: a hipGraph would replay the learning
        rate and the bias-correction step of capture time for ever -- inside a captured step use adam_captured (the
        device-side {lr, t} of rcgan_adam_tf).  t travels as fp32: exact up to 2^24 optimiser steps."""
        c = self.ctx
        if c.capturing:
            raise RuntimeError("ParamGroup.adam() bakes lr / t into the launch and must not be captured into a hipGraph: "
                               "use adam_captured()")
        if self._t >= float(1 << 24):
            raise OverflowError("Adam step counter %d is not exactly representable as fp32" % int(self._t))
        hi = self.count if hi is None else hi
        o = lo * 4
        c.check(c.lib.rcgan_adam_tf_host(c.h, hi - lo, self.value.data_ptr() + o, self.grad.data_ptr() + o,
                                         self.m.data_ptr() + o, self.v.data_ptr() + o, self._lr, self._t,
                                         beta1, beta2, eps, clip, grad_scale))
        self.version += 1
