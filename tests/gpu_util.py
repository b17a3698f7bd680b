"""Helpers shared by the -m gpu parity tests (HIP path vs the numpy oracle)."""
import numpy as np
import torch


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32))).bfloat16().float().numpy()


def f16_round(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


HALF = ("bf16", "f16")


def half_round(mode, a):
    """Round to the 16-bit activation format of ``mode`` (identity for f32)."""
    return bf16_round(a) if mode == "bf16" else (f16_round(a) if mode == "f16" else np.asarray(a, np.float32))


def rel_err(a, ref):
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.linalg.norm(a - ref) / (np.linalg.norm(ref) + 1e-30))


def assert_close(a, ref, tol, what=""):
    """Max-abs error relative to the reference's scale: |a-ref| <= tol * (max|ref| + tiny)."""
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    assert a.shape == ref.shape, (what, a.shape, ref.shape)
    assert np.isfinite(a).all(), what + ": non-finite values"
    scale = np.abs(ref).max() + 1e-12
    err = np.abs(a - ref).max() / scale
    assert err <= tol, "%s: max err %.3e (rel to max|ref|=%.3e) > %.1e; norm-rel %.3e" % (what, err, scale, tol, rel_err(a, ref))


def make_ctx(dtype, arena=1 << 30):
    import rcgan_amd  # noqa: F401
    from rcgan_amd.runtime import Context
    return Context(0, dtype, arena_bytes=arena, ws_bytes=1 << 30)


class FakeParam:
    """A trainable fp32 tensor with its own gradient buffer, for op-level tests."""

    def __init__(self, ctx, arr):
        from rcgan_amd import _lib as L
        self.t = ctx.persistent(np.shape(arr), L.F32)
        ctx.view(self.t).copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(arr, np.float32))))
        self.t.grad = ctx.persistent(np.shape(arr), L.F32, fill=0.0)
        self.t.req = True
        torch.cuda.synchronize()

    def grad(self, ctx):
        return ctx.download(self.t.grad)
