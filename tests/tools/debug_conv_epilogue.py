"""Debug helper: error pattern of one MFMA conv forward against the numpy oracle (which rows / channels are wrong)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import nn
from tests.gpu_util import FakeParam, half_round, make_ctx
from rcgan_amd import ops as O

n, h, w, cin, cout, k = [int(v) for v in sys.argv[1:7]] if len(sys.argv) > 6 else (2, 8, 8, 64, 64, 3)
ctx = make_ctx("bf16")
rs = np.random.RandomState(0)
x = half_round("bf16", rs.randn(n, h, w, cin))
wgt = half_round("bf16", (rs.randn(k, k, cin, cout) / np.sqrt(k * k * cin)).astype(np.float32))
b = rs.randn(cout).astype(np.float32)
ctx.new_step()
y = O.conv2d(ctx, ctx.upload(x), O.Weight(ctx, FakeParam(ctx, wgt).t), FakeParam(ctx, b).t, k)
got = ctx.download(y).reshape(-1, cout)
ref = (nn.conv2d_fwd(x.astype(np.float64), wgt.astype(np.float64), 1) + b).reshape(-1, cout)
bad = np.abs(got - ref) > 2e-2 * np.abs(ref).max()
print("bad fraction", bad.mean())
print("bad by pixel%32:", bad.reshape(-1, 32, cout).mean((0, 2)).round(2))
print("bad by channel:", bad.mean(0).round(2))
# is a wrong value some other element of the reference?
m, c = np.argwhere(bad)[0] if bad.any() else (0, 0)
print("first bad", m, c, got[m, c], ref[m, c], "matches ref at", np.argwhere(np.abs(ref - got[m, c]) < 2e-2)[:6].tolist())
print("got - bias matches?", np.argwhere(np.abs(ref - b - (got[m, c] - b[c])) < 2e-2)[:4].tolist())
