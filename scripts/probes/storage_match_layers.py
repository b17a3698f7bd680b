#!/usr/bin/env python3
"""Where does the storage-matched oracle part from the device?  Generator forward at B = 64, block by block: fraction of stored
elements that differ and norm-relative difference, float oracle / storage-matched oracle."""
import os
import sys

os.environ["RCGAN_BN_INTO_CONV"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.nn.functional as F

import rcgan_amd  # noqa: F401
from rcgan_amd import cifar as CF
from oracle.torch_port import CifarTorch, cond_bn, upsample2
from rcgan_amd.cifar import CifarRCGAN, create_variables

B = 64
variables = create_variables(0, "rcgan", False, "linear", True, 0.2)
m = CifarRCGAN(algorithm="rcgan", alpha=0.6, batch_size=B, dtype="bf16", seed=11, use_graphs=False, device_rng=False, variables=variables)
rs = np.random.RandomState(5)
labels = rs.randint(10, size=B)
z = torch.from_numpy(rs.randn(B, 128).astype(np.float32)).to(torch.bfloat16).to(torch.float32).numpy()
P, U = m.get_params(), m.get_state()

dev = {}
ctx = m.ctx
orig_block, orig_lin, orig_bn, orig_conv = CF.G_ResidualBlock, CF.Linear, CF.cond_batchnorm, CF.Conv2D


def block(inputs, input_dim, output_dim, filter_size, name, labels_, segments=1):
    out = orig_block(inputs, input_dim, output_dim, filter_size, name, labels_, segments)
    dev[name] = ctx.download(out).astype(np.float64)
    return out


def lin(*a, **k):
    out = orig_lin(*a, **k)
    dev[a[3]] = ctx.download(out).astype(np.float64)
    return out


def bn(name, *a, **k):
    out = orig_bn(name, *a, **k)
    try:
        dev[name] = ctx.download(out).astype(np.float64)
    except Exception as e:      # (a deferred apply has nothing to download)
        dev[name] = None
    return out


def conv(inputs, i, o, fs, st, name, **k):
    out = orig_conv(inputs, i, o, fs, st, name, **k)
    dev[name] = ctx.download(out).astype(np.float64)
    return out


CF.G_ResidualBlock, CF.Linear, CF.cond_batchnorm, CF.Conv2D = block, lin, bn, conv
img = m.sample(labels, z).astype(np.float64)
CF.G_ResidualBlock, CF.Linear, CF.cond_batchnorm, CF.Conv2D = orig_block, orig_lin, orig_bn, orig_conv


def cmp(name, a, d):
    if d is None:
        return
    a = a.detach().numpy().astype(np.float64).reshape(d.shape)
    print("  %-28s differ %.4f  norm-rel %.3e" % (name, float(np.mean(a != d)), np.linalg.norm(a - d) / np.linalg.norm(d)))


for st in (None, "bf16"):
    print("storage", st)
    net = CifarTorch(P, U, torch.float32, storage=st)
    q = net.q
    lab = torch.as_tensor(labels, dtype=torch.long)
    with torch.no_grad():
        zz = torch.as_tensor(z)
        o = q(zz @ net.qw(net.P["Generator/G.Input/W"]) + net.P["Generator/G.Input/b"]).reshape(-1, 4, 4, 1024)
        cmp("G.Input", o, dev["G.Input"])
        for k in (1, 2, 3):
            name = "Generator/G.Block.%d" % k
            x = o
            sc = q(net.conv(x, name + ".Shortcut"))
            cmp("G.Block.%d.Shortcut" % k, sc, dev["G.Block.%d.Shortcut" % k])
            o1 = q(F.relu(cond_bn(x, lab, net.P[name + ".N1/CondBatchNorm/scale"], net.P[name + ".N1/CondBatchNorm/offset"])))
            cmp("G.Block.%d.N1" % k, o1, dev["G.Block.%d.N1" % k])
            o2 = q(net.conv(o1, name + ".Conv1", form="up" if st else None) if st else net.conv(upsample2(o1), name + ".Conv1"))
            cmp("G.Block.%d.Conv1" % k, o2, dev["G.Block.%d.Conv1" % k])
            o3 = q(F.relu(cond_bn(o2, lab, net.P[name + ".N2/CondBatchNorm/scale"], net.P[name + ".N2/CondBatchNorm/offset"])))
            cmp("G.Block.%d.N2" % k, o3, dev["G.Block.%d.N2" % k])
            o = q(net.conv(o3, name + ".Conv2") + upsample2(sc))
            cmp("G.Block.%d (out)" % k, o, dev["G.Block.%d" % k])
        on = q(F.relu(cond_bn(o, lab, net.P["Generator/G.OutputNorm/CondBatchNorm/scale"], net.P["Generator/G.OutputNorm/CondBatchNorm/offset"])))
        cmp("G.OutputNorm", on, dev["G.OutputNorm"])
        oc_ = q(net.conv(on, "Generator/G.Output"))
        cmp("G.Output (conv)", oc_, dev["G.Output"])
        im = q(torch.tanh(oc_))
        cmp("image", im, img)
m.ctx.close()
