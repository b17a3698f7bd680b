#!/usr/bin/env python3
"""How faithful is the storage-matched oracle (oracle/torch_port.py, CifarTorch(storage="bf16"))?  One critic step at B = 64 on the
device; the generator's images and the critic's logits against the float oracle and the storage-matched one."""
import os
import sys

os.environ["RCGAN_BN_INTO_CONV"] = "0"      # forward-only passes then store what the taped pass stores (G.OutputNorm written out)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import rcgan_amd  # noqa: F401
from oracle import cifar as oc
from oracle.torch_port import CifarTorch
from rcgan_amd import _lib as L
from rcgan_amd.cifar import CifarRCGAN, create_variables

B = 64
variables = create_variables(0, "rcgan", False, "linear", True, 0.2)
m = CifarRCGAN(algorithm="rcgan", alpha=0.6, batch_size=B, dtype="bf16", seed=11, use_graphs=False, device_rng=False, variables=variables)
m.head_logits = m.ctx.persistent((2 * B, 10), L.F32, fill=0.0)
rs = np.random.RandomState(5)
Cm = oc.c_alpha(0.6)
lab = rs.randint(10, size=B)
z = rs.randn(B, 128).astype(np.float32)
z = torch.from_numpy(z).to(torch.bfloat16).to(torch.float32).numpy()
raw = dict(images=rs.randint(0, 256, size=(B, 3072)), noise=rs.uniform(0, 1 / 128., size=(B, 3072)).astype(np.float32), labels=lab,
           labels_random=rs.randint(10, size=B), labels_biased=rs.randint(10, size=B), inv_weights=np.linalg.inv(Cm)[lab].astype(np.float32), z=z)
m.set_inputs(labels_all=np.concatenate([raw["labels"], raw["labels_biased"]]), **raw)
P, U = m.get_params(), m.get_state()
m.d_step(iteration=0)
fake_dev = m.sample(raw["labels_random"], z).reshape(B, 3072).astype(np.float64)
logits = m.ctx.download(m.head_logits)
d_dev = np.concatenate([logits[np.arange(B), lab], logits[B + np.arange(B), raw["labels_biased"]]])
real = oc.preprocess_real(raw["images"], raw["noise"]).astype(np.float32)
for st in (None, "bf16"):
    net = CifarTorch(P, U, torch.float32, storage=st)
    with torch.no_grad():
        fake = net.generator(raw["labels_random"], z)
        feat, wgan = net.discriminator(torch.cat([torch.as_tensor(real), fake], 0), True)
        d = wgan + (feat * net.projection(np.concatenate([lab, raw["labels_biased"]]))).sum(1)
    f = fake.numpy().astype(np.float64)
    print("storage %-5s  images: norm-rel %.3e, fraction of elements that differ %.4f | logits: max abs diff %.3e (scale %.2f)"
          % (st, np.linalg.norm(f - fake_dev) / np.linalg.norm(fake_dev), float(np.mean(f != fake_dev)),
             float(np.abs(d.numpy() - d_dev).max()), float(np.abs(d_dev).max())))
m.ctx.close()
