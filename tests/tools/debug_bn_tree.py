"""Debug helper: rcgan_bn_stats on the tree path vs numpy."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.gpu_util import make_ctx
from rcgan_amd import _lib as L

mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
ctx = make_ctx(mode)
rs = np.random.RandomState(0)
for rows, c in ((32768, 128), (16384, 256), (8192, 512), (65536, 64)):
    x = (rs.randn(rows, c) * 1.5 + 0.3).astype(np.float32)
    ctx.new_step()
    xd = ctx.upload(x)
    mean = ctx.empty((c,), L.F32); rstd = ctx.empty((c,), L.F32)
    for rep in range(3):
        ctx.check(ctx.lib.rcgan_bn_stats(ctx.h, rows, c, xd.dtype, C.c_void_p(xd.ptr), 1e-5, C.c_void_p(mean.ptr), C.c_void_p(rstd.ptr), None, None, 0.0,
                                         C.c_void_p(ctx.ws_ptr), ctx.ws_bytes))
        xx = ctx.download(xd).astype(np.float64)
        m, r = ctx.download(mean), ctx.download(rstd)
        em = np.abs(m - xx.mean(0)); er = np.abs(r - 1 / np.sqrt(xx.var(0) + 1e-5))
        print(rows, c, "rep", rep, "mean err max %.3e at ch %d; rstd err max %.3e; bad channels %d" % (em.max(), em.argmax(), er.max(), (em > 1e-4).sum()),
              "sum ratio", (m[:4] / xx.mean(0)[:4]).round(4))
