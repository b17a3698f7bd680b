import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.test_gpu_cifar_step import _batches, _make, _labels_all
from tests.gpu_util import rel_err
from oracle import cifar as oc
alg = sys.argv[1] if len(sys.argv) > 1 else "rcgan"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
perm = alg == "rcgan-u"
rs = np.random.RandomState(21)
B = 4
C, raw, gb = _batches(rs, B)
m, P, Uo = _make(alg, perm, B, dtype)
cfg = dict(algorithm=alg, C=C, perm_classifier=perm, perm_multiplier=1.0)
m.set_inputs(labels_all=_labels_all(alg, raw), **raw)
m.d_step(iteration=0)
ob = dict(real=oc.preprocess_real(raw["images"], raw["noise"]), labels=raw["labels"], labels_random=raw["labels_random"],
          labels_biased=raw["labels_biased"], inv_weights=raw["inv_weights"], z=raw["z"])
c64, g64 = oc.d_grads(P, dict(Uo), cfg, ob, dtype=np.float64)
c32, g32 = oc.d_grads(P, dict(Uo), cfg, ob, dtype=np.float32)
print("D loss gpu %.8f o32 %.8f o64 %.8f" % (m.losses()[0], c32, c64))
got = m.get_grads(m.PD)
mx = lambda a, r: np.abs(a - r).max() / (np.abs(r).max() + 1e-30)
for k, g in g64.items():
    print("D %-55s gpu-vs-o64 %.2e  o32-vs-o64 %.2e" % (k, mx(got[k], g), mx(g32[k], g)))
# G step from the SAME (pre-update) oracle weights is not possible on the device (D already stepped): re-make
m.ctx.close()
m, P, Uo = _make(alg, perm, B, dtype)
m.set_inputs(**gb)
m.g_step(iteration=1)
og = dict(labels_random_G=gb["labels_random_G"], labels_biased_G=gb["labels_biased_G"], z=gb["z_G"])
c64, g64 = oc.g_grads(P, dict(Uo), cfg, og, dtype=np.float64)
c32, g32 = oc.g_grads(P, dict(Uo), cfg, og, dtype=np.float32)
print("G loss gpu %.8f o32 %.8f o64 %.8f" % (m.losses()[1], c32, c64))
got = m.get_grads(m.PG)
if m.PC is not None:
    got.update(m.get_grads(m.PC))
for k, g in g64.items():
    print("G %-55s gpu-vs-o64 %.2e  o32-vs-o64 %.2e" % (k, mx(got[k], g), mx(g32[k], g)))
