import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rcgan_amd
from rcgan_amd import _lib as L, ops as O
from rcgan_amd.runtime import Context
from tests.gpu_util import FakeParam
ctx = Context(0, "bf16", arena_bytes=1 << 28, ws_bytes=1 << 28)
n, h, w = 1, 8, 8
cin = cout = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = n * h * w
x = ((np.arange(M * cin).reshape(M, cin)) % 251).astype(np.float32).reshape(n, h, w, cin)
wt = np.eye(cin, cout, dtype=np.float32).reshape(1, 1, cin, cout)
ctx.new_step()
xd = ctx.upload(x)
wp = FakeParam(ctx, wt)
y = O.conv2d(ctx, xd, O.Weight(ctx, wp.t), None, 1)
out = ctx.download(y).reshape(M, cout)
ref = x.reshape(M, cin)
bad = np.argwhere(out != ref)
print("mismatches", len(bad), "of", out.size)
for m in (0, 1, 2, 9, 17):
    print("row", m, "got", out[m, :12].astype(int), "ref", ref[m, :12].astype(int))
# which (row, col) of ref does each output equal? (values are unique mod 251 within small windows)
bc = sorted(set(int(b[1]) // 8 for b in bad))
br = sorted(set(int(b[0]) for b in bad))
print("bad column chunks", bc[:40])
print("bad rows", br[:70])
for m in (0, 1, 8):
    srcs = []
    for c in range(0, cin, 8):
        v = out[m, c]
        cand = np.argwhere(ref == v)
        srcs.append(tuple(cand[0]) if len(cand) else None)
    print("row", m, "chunk sources", srcs)
