import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rcgan_amd
from rcgan_amd.mnist import MnistRCGAN, create_variables
from oracle import mnist as om
from tests.test_gpu_mnist_step import _batch
rs = np.random.RandomState(41)
B = 8
layers = (1, 3)
variables = create_variables(0, "projection", False, True, True, layers)
gs, ds, cs, S, U = variables
C, b = _batch(rs, B)
m = MnistRCGAN(algorithm="rcgan", batch_size=B, dtype="f32", concat_y=True, concat_y_layers=layers, use_graphs=False, variables=variables)
cfg = dict(algorithm="rcgan", disc_type="projection", loss_fn="hinge", perm_regularizer=True, perm_multiplier=10.0, spectral_norm=True, C=C,
           concat_y=True, concat_y_layers=layers)
m.set_inputs(**b)
m.d_step()
for run in range(3):
    P, st = m.get_params(), m.get_state()
    So = {k: v for k, v in st.items() if "moving" in k}
    Uo = {k: v for k, v in st.items() if "spectral" in k}
    L64, g64 = om.g_grads(P, {k: v.copy() for k, v in So.items()}, dict(Uo), cfg, b, dtype=np.float64)
    m.g_step()
    gg = m.get_grads(m.PG)
    print("run", run, m.losses()["g_loss"], L64["g_loss"])
    for k, g in g64.items():
        print("  %-32s err %.2e  |ref| %.2e" % (k, np.abs(gg[k] - g).max() / (np.abs(g).max() + 1e-30), np.abs(g).max()))
