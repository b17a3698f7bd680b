"""Debug helper: elements where the conditional batch-norm backward differs from the oracle (mask taken from the kernel's y)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import nn
from tests.gpu_util import FakeParam, half_round, make_ctx
from rcgan_amd import _lib as L, ops as O

mode = sys.argv[1] if len(sys.argv) > 1 else "f16"
shape = (128, 8, 8, 512)
ctx = make_ctx(mode)
rs = np.random.RandomState(len(shape) * 100 + shape[-1])
c = shape[-1]
x = half_round(mode, rs.randn(*shape) * 1.5 + 0.3)
gamma = (1 + 0.3 * rs.randn(10, c)).astype(np.float32); beta = (0.2 * rs.randn(10, c)).astype(np.float32)
labels = rs.randint(10, size=shape[0]).astype(np.int32)
ctx.new_step()
xd = ctx.upload(x); xd.req = True
gp, bp = FakeParam(ctx, gamma), FakeParam(ctx, beta)
y = O.batch_norm_act(ctx, xd, gp.t, bp.t, act=L.ACT_RELU, labels=ctx.upload(labels), n_labels=10)
yk = ctx.download(y)
pre, st = nn.cond_batchnorm_fwd(x.astype(np.float64), labels, gamma.astype(np.float64), beta.astype(np.float64))
dy = half_round(mode, rs.randn(*shape))
y.grad = ctx.upload(dy)
ctx.backward()
dxk = ctx.download(xd.grad)
dpre = dy.astype(np.float64) * (yk > 0)
dx, dg, db = nn.cond_batchnorm_bwd(dpre, x.astype(np.float64), labels, gamma.astype(np.float64), st)
err = np.abs(dxk - dx)
idx = np.argwhere(err > 6e-3 * np.abs(dx).max())
print("bad elements:", len(idx), "of", err.size)
for i in idx[:10]:
    i = tuple(i)
    print(i, "pre(oracle) %.3e  y_kernel %.3e  dy %.3f  dx_kernel %.4f  dx_ref %.4f" % (pre[i], yk[i], dy[i], dxk[i], dx[i]))
print("dgamma err", np.abs(gp.grad(ctx) - dg).max(), "dbeta err", np.abs(bp.grad(ctx) - db).max())
