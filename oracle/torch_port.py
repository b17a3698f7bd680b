"""PyTorch-CPU autograd re-implementation of the CIFAR / MNIST RCGAN graphs (TEST INFRASTRUCTURE, like the rest of oracle/).

A SECOND, independent implementation, written against the same reference lines the numpy oracle cites.  Two uses:
  * tests/test_oracle_vs_torch.py cross-checks the numpy oracle against it in float64 (it is not the reference and not
    the product);
  * bench.py's ``cpu_baseline`` leg times it in float32 on all host cores ("a threaded restatement (PyTorch-CPU
    ops) with threads = all host cores") -- TensorFlow 1.5, the reference's own CPU path, is not installable here.
Nothing on the product path imports it.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def same_pad(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return out, tot // 2, tot - tot // 2


def conv2d_same(x, w, stride=1):
    """x NHWC, w HWIO -> NHWC with TF SAME padding."""
    kh, kw = w.shape[0], w.shape[1]
    _, pt, pb = same_pad(x.shape[1], kh, stride)
    _, pl, pr = same_pad(x.shape[2], kw, stride)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn, w.permute(3, 2, 0, 1), stride=stride)
    return y.permute(0, 2, 3, 1)


def conv2d_transpose_same(x, w, out_shape, stride=2):
    """tf.nn.conv2d_transpose: w [kh,kw,Cout,Cin]; defined as the autograd gradient of conv2d_same."""
    probe = torch.zeros(out_shape, dtype=x.dtype, requires_grad=True)
    y = conv2d_same(probe, w, stride)
    (g,) = torch.autograd.grad(y, probe, grad_outputs=x, create_graph=True)
    return g


def l2n(v, eps=1e-12):
    return v / ((v ** 2).sum() ** 0.5 + eps)


def spectral_norm(w, u):
    wr = w.reshape(-1, w.shape[-1])
    v = l2n(u @ wr.t())
    u2 = l2n(v @ wr)
    sigma = (v @ wr @ u2.t())[0, 0]
    return (wr / sigma).reshape(w.shape), u2.detach()


def meanpool2(x):
    return (x[:, ::2, ::2, :] + x[:, 1::2, ::2, :] + x[:, ::2, 1::2, :] + x[:, 1::2, 1::2, :]) / 4.


def upsample2(x):
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def cond_bn(x, labels, scale_m, offset_m, eps=1e-5):
    mean = x.mean(dim=(0, 1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2), keepdim=True)
    g = scale_m[labels][:, None, None, :]
    b = offset_m[labels][:, None, None, :]
    inv = torch.rsqrt(var + eps) * g
    return x * inv + (b - mean * inv)


# ---- the sub-pixel forms of the resampling convolutions, as the product evaluates them with 16-bit filters ----
# upsample -> 3x3: output pixel (2i + ph, 2j + pw) = four taps (a, b) on low-resolution pixel (i + a - 1 + ph, j + b - 1 + pw) with the SUM
# of the filter taps kh in U(ph, a), kw in U(pw, b); 3x3 -> mean pool: pooled pixel (i, j) = sixteen taps (u, v) on pixel (2i + u - 1,
# 2j + v - 1) with a quarter of the sum over kh in P(u), kw in P(v).  The sums are taken in fp32 and rounded ONCE to the storage format:
# that rounding -- not the per-tap one -- is what the device's filters carry.
_U = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}
_PU = {0: (0,), 1: (0, 1), 2: (1, 2), 3: (2,)}


def conv_up_subpixel(x, w, qw):
    """x [n, h, w, ci] low resolution, w HWIO 3x3 (already divided by sigma) -> [n, 2h, 2w, co] = conv3x3_SAME(upsample2(x))."""
    n, h, wd, _ = x.shape
    xp = F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1))
    out = x.new_zeros((n, w.shape[3], 2 * h, 2 * wd))
    for ph in (0, 1):
        for pw in (0, 1):
            k = torch.stack([torch.stack([sum(w[kh, kw] for kh in _U[(ph, a)] for kw in _U[(pw, b)]) for b in (0, 1)]) for a in (0, 1)])
            y = F.conv2d(xp, qw(k).permute(3, 2, 0, 1))              # [n, co, h + 1, w + 1]: position p reads padded rows p, p + 1
            out[:, :, ph::2, pw::2] = y[:, :, ph:ph + h, pw:pw + wd]
    return out.permute(0, 2, 3, 1)


def conv_pool_subpixel(x, w, qw):
    """x [n, h, w, ci], w HWIO 3x3 -> [n, h/2, w/2, co] = meanpool2(conv3x3_SAME(x)) as one 4x4 stride-2 convolution."""
    k = torch.stack([torch.stack([0.25 * sum(w[kh, kw] for kh in _PU[u] for kw in _PU[v]) for v in range(4)]) for u in range(4)])
    xp = F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1))
    return F.conv2d(xp, qw(k).permute(3, 2, 0, 1), stride=2).permute(0, 2, 3, 1)


class _RoundBoth(torch.autograd.Function):
    """A tensor the product STORES in 16 bits: the value is rounded to the storage format on the way forward and its gradient on the
    way back (the product stores the gradient of a stored activation in the same format)."""

    @staticmethod
    def forward(ctx, x, st, gscale=1.0):
        ctx.st, ctx.gscale = st, gscale
        return x.to(st).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        # (fp16: the product stores loss-scaled gradients -- unscaled, most of them would be fp16 denormals or zero)
        return (g * ctx.gscale).to(ctx.st).to(g.dtype) / ctx.gscale, None, None


class _RoundFwd(torch.autograd.Function):
    """A prepared 16-bit filter: rounded on the way forward; its gradient is accumulated and kept in fp32 (straight through)."""

    @staticmethod
    def forward(ctx, x, st):
        return x.to(st).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


_STORAGE = {"bf16": torch.bfloat16, "f16": torch.float16}


class CifarTorch:
    """storage=None: the reference graph in ``dtype``.  storage="bf16" / "f16": the same graph with every tensor the PRODUCT keeps in
    16 bits rounded where the product rounds it (robust-conditional-gan_amd/cifar.py: convolution outputs behind their fused bias /
    residual / pooled sum, batch-norm + ReLU outputs, the pooled images, tanh output, prepared filters W / sigma) and the gradients of
    those tensors rounded on the way back -- a storage-matched comparison for the 16-bit step tests: what is left between it and the
    device is fp32 summation order, not 45 layers of independent rounding noise."""

    def __init__(self, P, U, dtype=torch.float64, storage=None, grad_scale=1.0):
        self.storage = _STORAGE[storage] if storage else None
        self.grad_scale = float(grad_scale)      # the product's loss scale (fp16 storage): stored gradients are rounded at that scale
        self.P = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in P.items()}
        self.U = {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in U.items()}
        self.U_new = {}
        self.dtype = dtype
        # hinge terms of the last disc_cost: name -> the relu ARGUMENT (1 - D(real), 1 + D(fake)) as evaluated here; and an optional
        # imposed activity pattern name -> bool array (tests/test_gpu_fullbatch_steps.py: the device's own pattern, so that a
        # logit within rounding distance of the hinge does not turn into a whole-sample difference of the gradients)
        self.hinge_args = {}
        self.hinge_mask = None

    def hinge(self, name, t):
        """relu(t) of a hinge term (gan_resnet.py:604-605,639-640,673-674); with an imposed activity pattern: t * mask."""
        self.hinge_args[name] = t.detach().clone()
        if self.hinge_mask is not None and name in self.hinge_mask:
            return t * torch.as_tensor(np.asarray(self.hinge_mask[name]), dtype=self.dtype).reshape(t.shape)
        return F.relu(t)

    def q(self, x):
        return _RoundBoth.apply(x, self.storage, self.grad_scale) if self.storage is not None else x

    def qw(self, w):
        return _RoundFwd.apply(w, self.storage) if self.storage is not None else w

    def conv(self, x, name, sn=False, update=True, form=None):
        """form (storage-matched graph only): "up" = x is the LOW-resolution input of an upsample-3x3 layer, "pool" = the layer is
        followed by the 2x2 mean pool; both in the product's sub-pixel form (summed filters rounded once)."""
        w = self.P[name + "/Filters"]
        if sn:
            key = name + "/filters/spectral_norm/u"
            w, u2 = spectral_norm(w, self.U[key])
            if update:
                self.U_new[key] = u2
        if form == "up":
            return conv_up_subpixel(x, w, self.qw) + self.P[name + "/Biases"]
        if form == "pool":
            return conv_pool_subpixel(x, w, self.qw) + self.P[name + "/Biases"]
        return conv2d_same(x, self.qw(w), 1) + self.P[name + "/Biases"]

    def lin(self, x, name, sn=False, update=True):
        w = self.P[name + "/W"]
        if sn:
            key = name + "/spectral_norm/u"
            w, u2 = spectral_norm(w, self.U[key])
            if update:
                self.U_new[key] = u2
        return x @ w + self.P[name + "/b"]

    def gblock_stored(self, x, name, labels):
        """G_ResidualBlock as the product evaluates and stores it (cifar.py): the 1x1 shortcut on the low-resolution input, added
        upsampled inside Conv2's epilogue (one rounding of the sum)."""
        lab = torch.as_tensor(labels, dtype=torch.long)
        q = self.q
        sc = q(self.conv(x, name + ".Shortcut"))
        o = q(F.relu(cond_bn(x, lab, self.P[name + ".N1/CondBatchNorm/scale"], self.P[name + ".N1/CondBatchNorm/offset"])))
        o = q(self.conv(o, name + ".Conv1", form="up"))
        o = q(F.relu(cond_bn(o, lab, self.P[name + ".N2/CondBatchNorm/scale"], self.P[name + ".N2/CondBatchNorm/offset"])))
        return q(self.conv(o, name + ".Conv2") + upsample2(sc))

    def gblock(self, x, name, labels):
        if self.storage is not None:
            return self.gblock_stored(x, name, labels)
        lab = torch.as_tensor(labels, dtype=torch.long)
        sc = self.conv(upsample2(x), name + ".Shortcut")
        o = cond_bn(x, lab, self.P[name + ".N1/CondBatchNorm/scale"], self.P[name + ".N1/CondBatchNorm/offset"])
        o = self.conv(upsample2(F.relu(o)), name + ".Conv1")
        o = cond_bn(o, lab, self.P[name + ".N2/CondBatchNorm/scale"], self.P[name + ".N2/CondBatchNorm/offset"])
        o = self.conv(F.relu(o), name + ".Conv2")
        return sc + o

    def generator(self, labels, z):
        lab = torch.as_tensor(labels, dtype=torch.long)
        z = torch.as_tensor(z, dtype=self.dtype)
        if self.storage is not None:      # G.Input runs on the 16-bit matrix cores: rounded weights, stored output
            o = self.q(z @ self.qw(self.P["Generator/G.Input/W"]) + self.P["Generator/G.Input/b"]).reshape(-1, 4, 4, 1024)
        else:
            o = self.lin(z, "Generator/G.Input").reshape(-1, 4, 4, 1024)
        for k in (1, 2, 3):
            o = self.gblock(o, "Generator/G.Block.%d" % k, labels)
        o = cond_bn(o, lab, self.P["Generator/G.OutputNorm/CondBatchNorm/scale"],
                    self.P["Generator/G.OutputNorm/CondBatchNorm/offset"])
        o = self.q(torch.tanh(self.q(self.conv(self.q(F.relu(o)), "Generator/G.Output"))))      # (the image-end convolution stores its output; tanh is a launch of its own)
        return o.reshape(-1, 3072)

    def discriminator_stored(self, x, update):
        """Discriminator as the product evaluates and stores it with 16-bit activations (cifar.py, fused-pool path): MeanPoolConv
        shortcuts on the pooled input, ConvMeanPool with the pool folded into the convolution and the shortcut added in its epilogue,
        identity blocks with the residual added in Conv2's epilogue; pooled features and the head in fp32."""
        p = "Discriminator/"
        kw = dict(sn=True, update=update)
        q = self.q
        x = q(x.reshape(-1, 32, 32, 3))
        t = q(self.conv(q(meanpool2(x)), p + "D.Block.1.Shortcut", **kw))
        h = q(self.conv(x, p + "D.Block.1.Conv1", **kw))
        x = q(self.conv(F.relu(h), p + "D.Block.1.Conv2", form="pool", **kw) + t)
        t = q(self.conv(q(meanpool2(x)), p + "D.Block.2.Shortcut", **kw))
        h = q(self.conv(F.relu(x), p + "D.Block.2.Conv1", **kw))
        x = q(self.conv(F.relu(h), p + "D.Block.2.Conv2", form="pool", **kw) + t)
        for k in (3, 4, 5, 6):
            h = q(self.conv(F.relu(x), p + "D.Block.%d.Conv1" % k, **kw))
            x = q(self.conv(F.relu(h), p + "D.Block.%d.Conv2" % k, **kw) + x)
        feat = F.relu(x).mean(dim=(1, 2))
        wgan = self.lin(feat, p + "D.Output", **kw).reshape(-1)
        return feat, wgan

    def discriminator(self, x, update):
        if self.storage is not None:
            return self.discriminator_stored(x, update)
        p = "Discriminator/"
        kw = dict(sn=True, update=update)
        x = x.reshape(-1, 32, 32, 3)
        sc = self.conv(meanpool2(x), p + "D.Block.1.Shortcut", **kw)
        o = self.conv(x, p + "D.Block.1.Conv1", **kw)
        o = meanpool2(self.conv(F.relu(o), p + "D.Block.1.Conv2", **kw))
        x = sc + o
        sc = meanpool2(self.conv(x, p + "D.Block.2.Shortcut", **kw))
        o = self.conv(F.relu(x), p + "D.Block.2.Conv1", **kw)
        o = meanpool2(self.conv(F.relu(o), p + "D.Block.2.Conv2", **kw))
        x = sc + o
        for k in (3, 4, 5, 6):
            o = self.conv(F.relu(x), p + "D.Block.%d.Conv1" % k, **kw)
            o = self.conv(F.relu(o), p + "D.Block.%d.Conv2" % k, **kw)
            x = x + o
        feat = F.relu(x).mean(dim=(1, 2))
        wgan = self.lin(feat, p + "D.Output", **kw).reshape(-1)
        return feat, wgan

    def projection(self, labels, update=True):
        p = "Discriminator/"
        e = self.P[p + "Embedding.Label/embedding_map"][torch.as_tensor(labels, dtype=torch.long)]
        return self.lin(e, p + "D.Embedding_y", sn=True, update=update)

    def perm(self, x, perm_type="linear"):
        p = "Discriminator/"
        h = self.lin(x.reshape(-1, 3072), p + "D.d_perm_classifier_h1", sn=True)
        if perm_type == "2layer":
            h = self.lin(h, p + "D.d_perm_classifier_h2", sn=True)
        return h

    def C(self, cfg):
        if "confusion_logits" in self.P:
            return torch.softmax(self.P["confusion_logits"], dim=-1)
        return torch.tensor(cfg["C"], dtype=self.dtype)

    def disc_cost(self, cfg, b):
        alg = cfg["algorithm"]
        B = len(b["labels"])
        fake = self.generator(b["labels_random"], b["z"])
        real = torch.as_tensor(b["real"], dtype=self.dtype)
        if alg == "rcgan-u":
            feat, wgan = self.discriminator(real, True)
            dreal = wgan + (feat * self.projection(b["labels"])).sum(1)
            ff, wf = self.discriminator(fake, True)
            E = self.projection(np.arange(10))
            dfake = wf[:, None] + ff @ E.t()
            y = self.C(cfg)[torch.as_tensor(b["labels_random"], dtype=torch.long)]
            cost = (self.hinge("fake", 1 + dfake) * y).sum(1).mean() + self.hinge("real", 1 - dreal).mean()
        else:
            feat, wgan = self.discriminator(torch.cat([real, fake], 0), True)
            if alg in ("biased", "rcgan"):
                lab = np.concatenate([b["labels"], b["labels_random"] if alg == "biased" else b["labels_biased"]])
                d = wgan + (feat * self.projection(lab)).sum(1)
                cost = self.hinge("real", 1 - d[:B]).mean() + self.hinge("fake", 1 + d[B:]).mean()
            else:
                cols = []
                for j in range(10):
                    lab = np.concatenate([np.full(B, j), b["labels_random"]])
                    d = wgan + (feat * self.projection(lab)).sum(1)
                    cols.append(self.hinge("real%d" % j, 1 - d[:B]).reshape(B, 1))
                    fl = self.hinge("fake", 1 + d[B:]).mean()
                w = torch.as_tensor(b["inv_weights"], dtype=self.dtype)
                cost = (torch.cat(cols, 1) * w).sum(1).mean() + fl
        if cfg.get("perm_classifier"):
            tgt = F.one_hot(torch.as_tensor(b["labels"], dtype=torch.long), 10).to(self.dtype)
            cost = cost + F.binary_cross_entropy_with_logits(self.perm(real, cfg.get("perm_type", "linear")), tgt)
        return cost

    def gen_cost(self, cfg, b):
        alg = cfg["algorithm"]
        fake = self.generator(b["labels_random_G"], b["z"])
        feat, wgan = self.discriminator(fake, False)
        lab = b["labels_random_G"] if alg in ("biased", "unbiased") else b["labels_biased_G"]
        emb = self.projection(lab)
        if alg == "rcgan-u":
            E = self.projection(np.arange(10))
            dfake = wgan[:, None] + feat @ E.t()
            y = self.C(cfg)[torch.as_tensor(b["labels_random_G"], dtype=torch.long)]
            cost = (-dfake * y).sum(1).mean()
        else:
            cost = -(wgan + (feat * emb).sum(1)).mean()
        if cfg.get("perm_classifier"):
            tgt = F.one_hot(torch.as_tensor(b["labels_random_G"], dtype=torch.long), 10).to(self.dtype)
            cost = cost + cfg.get("perm_multiplier", 1.0) * F.binary_cross_entropy_with_logits(
                self.perm(fake, cfg.get("perm_type", "linear")), tgt)
        return cost


class CifarTorchTrainer:
    """One D step / G step of the CIFAR RCGAN on the CPU: autograd of disc_cost / gen_cost + TF-form Adam on the step's
    variable group, spectral-norm u update (gan_resnet.py:802-817, 928-947) -- the timed body of bench.py's cpu_baseline."""

    def __init__(self, P, U, cfg, lr=2e-4, dtype=torch.float32):
        self.net = CifarTorch(P, U, dtype)
        self.cfg, self.lr = cfg, lr
        self.m = {k: torch.zeros_like(v) for k, v in self.net.P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.net.P.items()}
        self.t = {"D": 0, "G": 0}

    def _step(self, cost, prefix, key):
        names = [k for k in self.net.P if k.startswith(prefix)]
        grads = torch.autograd.grad(cost, [self.net.P[k] for k in names], allow_unused=True)
        self.t[key] += 1
        with torch.no_grad():
            for k, g in zip(names, grads):
                if g is None:
                    continue
                w, self.m[k], self.v[k] = adam_tf_torch(self.net.P[k], g, self.m[k], self.v[k], self.t[key], self.lr, 0.0, 0.9)
                self.net.P[k].copy_(w)
            self.net.U.update(self.net.U_new)
            self.net.U_new = {}
        return float(cost.detach())

    def d_step(self, batch):
        return self._step(self.net.disc_cost(self.cfg, batch), "Discriminator/", "D")

    def g_step(self, batch):
        return self._step(self.net.gen_cost(self.cfg, batch), "Generator/", "G")


def adam_tf_torch(w, g, m, v, t, lr, b1, b2, eps=1e-8):
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    return w - lr_t * m / (v.sqrt() + eps), m, v


# ----------------------------------------------------------------------------------------------------
# MNIST (mnist/model.py) -- second implementation for the oracle cross-check
# ----------------------------------------------------------------------------------------------------
def bn_train(x, gamma, beta, eps=1e-5):
    dims = tuple(range(x.dim() - 1))
    mean = x.mean(dim=dims, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=dims, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


class MnistTorch:
    def __init__(self, P, U, cfg, dtype=torch.float64):
        self.P = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in P.items()}
        self.U = {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in U.items()}
        self.cfg, self.dtype = cfg, dtype

    def T(self, a):
        return torch.as_tensor(np.asarray(a), dtype=self.dtype)

    def lin(self, x, name):
        return x @ self.P[name + "/Matrix"] + self.P[name + "/bias"]

    def conv(self, x, name, sn):
        w = self.P[name + "/w"]
        if sn:
            w, _ = spectral_norm(w, self.U[name + "/spectral_norm/u"])
        return conv2d_same(x, w, 2) + self.P[name + "/biases"]

    def deconv(self, x, shape, name):
        return conv2d_transpose_same(x, self.P[name + "/w"], shape, 2) + self.P[name + "/biases"]

    def bn(self, x, name):
        return bn_train(x, self.P[name + "/gamma"], self.P[name + "/beta"])

    def cat(self, x, y):
        y = self.T(y)
        if x.dim() == 4:
            n, h, w, _ = x.shape
            return torch.cat([x, y[:, None, None, :].expand(n, h, w, y.shape[1])], 3)
        return torch.cat([x, y], 1)

    def generator(self, z, y):
        p = "generator/"
        B = len(z)
        h = self.cat(self.T(z), y)
        h0 = self.cat(F.relu(self.bn(self.lin(h, p + "g_h0_lin"), p + "g_bn0")), y)
        h1 = F.relu(self.bn(self.lin(h0, p + "g_h1_lin"), p + "g_bn1")).reshape(B, 7, 7, 128)
        h2 = F.relu(self.bn(self.deconv(self.cat(h1, y), (B, 14, 14, 128), p + "g_h2"), p + "g_bn2"))
        return torch.sigmoid(self.deconv(self.cat(h2, y), (B, 28, 28, 1), p + "g_h3"))

    def discriminator(self, image, y):
        p = "discriminator/"
        B = image.shape[0]
        lre = lambda v: torch.maximum(v, 0.2 * v)
        if self.cfg.get("disc_type", "projection") == "projection":
            sn = self.cfg.get("spectral_norm", True)
            layers = self.cfg.get("concat_y_layers", ()) if self.cfg.get("concat_y") else ()
            x = image
            for i in range(4):
                if (i + 1) in layers:
                    x = self.cat(x, y)
                x = self.conv(x, p + "d_h%d_conv" % i, sn)
                if i > 0:
                    x = self.bn(x, p + "d_bn%d" % i)
                x = lre(x)
            h3 = x.mean(dim=(1, 2))
            return self.lin(h3, p + "d_h4_lin") + (h3 * self.lin(self.T(y), p + "d_h5_y_lin")).sum(1, keepdim=True)
        h0 = self.cat(lre(self.conv(self.cat(image, y), p + "d_h0_conv", False)), y)
        h1 = lre(self.bn(self.conv(h0, p + "d_h1_conv", False), p + "d_bn1")).reshape(B, -1)
        h3 = lre(self.bn(self.lin(self.cat(h1, y), p + "d_h3_lin"), p + "d_bn2"))
        return self.lin(self.cat(h3, y), p + "d_h4_lin")

    def losses(self, b):
        cfg = self.cfg
        alg = cfg["algorithm"]
        hinge = cfg.get("loss_fn", "hinge") == "hinge"
        bce = lambda x, z: F.binary_cross_entropy_with_logits(x, torch.full_like(x, z), reduction="none")
        lr_ = (lambda x: F.relu(1 - x)) if hinge else (lambda x: bce(x, 1.0))
        lf_ = (lambda x: F.relu(1 + x)) if hinge else (lambda x: bce(x, 0.0))
        lg_ = (lambda x: -x) if hinge else (lambda x: bce(x, 1.0))
        B = len(b["z"])
        G = self.generator(b["z"], b["y_gen"])
        x = self.T(b["images"])
        eye = lambda i: np.eye(10)[np.full(B, i)]
        out = {}
        if alg in ("biased", "rcgan", "ambient"):
            out["d_loss_real"] = lr_(self.discriminator(x, b["y_real"])).mean()
        else:
            cols = torch.cat([lr_(self.discriminator(x, eye(i))) for i in range(10)], 1)
            out["d_loss_real"] = (cols * self.T(b["y_real_weights"])).sum(1).mean()
        if alg in ("rcgan", "ambient") and cfg.get("estimate_confuse"):
            los = [self.discriminator(G, eye(i)) for i in range(10)]
            C = torch.softmax(self.P["confusion_logits"], -1)
            yc = self.T(b["y_gen"]) @ C
            out["d_loss_fake"] = (torch.cat([lf_(l) for l in los], 1) * yc).sum(1).mean()
            out["g_loss"] = (torch.cat([lg_(l) for l in los], 1) * yc).sum(1).mean()
        else:
            lo = self.discriminator(G, b["y_fake"] if alg in ("rcgan", "ambient") else b["y_gen"])
            out["d_loss_fake"] = lf_(lo).mean()
            out["g_loss"] = lg_(lo).mean()
        if cfg.get("perm_regularizer", True):
            cl = lambda v: self.lin(v.reshape(B, -1), "classifier/d_classifier_h1")
            out["class_loss_real"] = F.binary_cross_entropy_with_logits(cl(x), self.T(b["y_real"]))
            out["class_loss_fake"] = F.binary_cross_entropy_with_logits(cl(G), self.T(b["y_gen"]))
        return out
