import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import rcgan_amd
from oracle import mnist as om
from oracle import labels as LB
from rcgan_amd.mnist import MnistRCGAN, create_variables
alg, disc, est, loss, concat = "unbiased", "projection", False, "hinge", False
rs = np.random.RandomState(41)
B = 8
variables = create_variables(0, disc, est, True, True, ())
gs, ds, cs, S, U = variables
jit = np.random.RandomState(3)
def jitter(specs):
    out = []
    for n, shp, v in specs:
        if n.endswith(("/bias", "/biases", "/beta", "/gamma")):
            v = (v + 0.1 * jit.randn(*shp)).astype(np.float32)
        out.append((n, shp, v))
    return out
variables = (jitter(gs), jitter(ds), cs, S, U)
C = LB.one_coin(0.3)
eye = np.eye(10, dtype=np.float32)
yr = rs.randint(10, size=B)
b = dict(images=rs.rand(B, 28, 28, 1).astype(np.float32), z=rs.uniform(-1, 1, size=(B, 100)).astype(np.float32),
         y_real=eye[yr], y_gen=eye[rs.randint(10, size=B)], y_fake=eye[rs.randint(10, size=B)],
         y_real_weights=np.linalg.inv(C)[yr].astype(np.float32))
m = MnistRCGAN(algorithm=alg, alpha=0.3, batch_size=B, dtype="f32", disc_type=disc, loss_fn=loss, estimate_confuse=est,
               perm_regularizer=True, perm_multiplier=10.0, spectral_norm=True, max_norm=True, use_graphs=False, variables=variables)
P = {n: v.copy() for n, _, v in variables[0] + variables[1] + variables[2]}
So = {k: v.copy() for k, v in S.items()}
Uo = {k: v.copy() for k, v in U.items()}
cfg = dict(algorithm=alg, disc_type=disc, estimate_confuse=est, loss_fn=loss, perm_regularizer=True, perm_multiplier=10.0,
           spectral_norm=True, C=C, concat_y=False, concat_y_layers=(), max_norm=True, confuse_multiplier=10.0)
m.set_inputs(**b)
m.d_step()
tr = om.Trainer(P, So, Uo, cfg)
tr.d_step(b)
P2 = m.get_params(); S2 = m.get_state()
Sx = {k: S2[k].reshape(np.shape(So[k])) for k in So}; Ux = {k: S2[k].reshape(np.shape(Uo[k])) for k in Uo}
LOAD = os.environ.get("DBG_LOAD")
for run in range(2):
    P2 = m.get_params(); S2 = m.get_state()
    if run == 1 and not LOAD:
        np.savez("/tmp/dbg_state.npz", **{"P/" + k: v for k, v in P2.items()}, **{"S/" + k: v for k, v in S2.items()})
    if run == 1 and LOAD:          # the OTHER build's weights and state at this point
        z = np.load("/tmp/dbg_state.npz")
        for grp in m.groups:
            for n in grp.names:
                grp.set(n, z["P/" + n])
        import torch
        for k, t in m.state.items():
            m.ctx.view(t).copy_(torch.from_numpy(np.ascontiguousarray(z["S/" + k].reshape(-1))))
        m.ctx.sync()
        P2 = m.get_params(); S2 = m.get_state()
    Sx = {k: S2[k].reshape(np.shape(So[k])) for k in So}; Ux = {k: S2[k].reshape(np.shape(Uo[k])) for k in Uo}
    L64, g64 = om.g_grads(P2, {k: v.copy() for k, v in Sx.items()}, dict(Ux), cfg, b, dtype=np.float64)
    m.g_step()
    got = m.get_grads(m.PG)
    print("run", run)
    for k, gref in g64.items():
        if k in got and gref.ndim == 2:
            a = got[k]
            d = (a - gref) ** 2
            col = d.sum(0)
            print("  %-36s norm-rel %.3e  share of the squared error in the worst column: %.3f (column %d of %d)"
                  % (k, np.linalg.norm(a - gref) / (np.linalg.norm(gref) + 1e-30), col.max() / (col.sum() + 1e-300), int(col.argmax()), col.size))
m.ctx.close()
