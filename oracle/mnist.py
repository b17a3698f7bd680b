"""numpy restatement of the MNIST DCGAN-style RCGAN step (oracle; test infrastructure).

Follows /root/reference/mnist/model.py (build_model :96-247, discriminator :644-703, generator :705-731,
gen_sampler :733-757, classifier :759-768, optimisers :250-262, step order :347-372) and mnist/ops.py,
mnist/sn.py.  Floating-point parity is UNPINNED by the reference (see oracle/__init__.py).
"""
import numpy as np
from . import nn
from .tape import Tape, Var

Y_DIM = 10
Z_DIM = 100
GF_DIM = 64
DF_DIM = 64
GFC_DIM = 1024
DFC_DIM = 1024


def _trunc_normal(rs, size, stddev):
    x = rs.normal(0.0, stddev, size=size)
    bad = np.abs(x) > 2 * stddev
    while bad.any():
        x[bad] = rs.normal(0.0, stddev, size=int(bad.sum()))
        bad = np.abs(x) > 2 * stddev
    return x.astype("float32")


def init_params(seed=0, disc_type="projection", estimate_confuse=False, perm_regularizer=True, spectral_norm=True,
                concat_y_layers=()):
    """Variables in the reference's creation order -> (P, S, U):
    trainable params, BN moving statistics, SN u vectors."""
    rs = np.random.RandomState(seed)
    P, S, U = {}, {}, {}

    def lin(name, cin, cout):
        P[name + "/Matrix"] = rs.normal(0.0, 0.02, size=(cin, cout)).astype("float32")      # ops.py:102-104
        P[name + "/bias"] = np.zeros((cout,), "float32")

    def bn(name, c):
        P[name + "/beta"] = np.zeros((c,), "float32")
        P[name + "/gamma"] = np.ones((c,), "float32")
        S[name + "/moving_mean"] = np.zeros((c,), "float32")
        S[name + "/moving_variance"] = np.ones((c,), "float32")

    def conv(name, cin, cout, sn):
        P[name + "/w"] = _trunc_normal(rs, (5, 5, cin, cout), 0.02)                          # ops.py:57-58
        if sn:
            U[name + "/spectral_norm/u"] = _trunc_normal(rs, (1, cout), 1.0)
        P[name + "/biases"] = np.zeros((cout,), "float32")

    def deconv(name, cout, cin):
        P[name + "/w"] = rs.normal(0.0, 0.02, size=(5, 5, cout, cin)).astype("float32")      # ops.py:74-75
        P[name + "/biases"] = np.zeros((cout,), "float32")

    if estimate_confuse:
        lim = np.sqrt(6.0 / (2 * Y_DIM))
        P["confusion_logits"] = rs.uniform(-lim, lim, size=(Y_DIM, Y_DIM)).astype("float32")
    g = "generator/"
    lin(g + "g_h0_lin", Z_DIM + Y_DIM, GFC_DIM)
    bn(g + "g_bn0", GFC_DIM)
    lin(g + "g_h1_lin", GFC_DIM + Y_DIM, GF_DIM * 2 * 7 * 7)
    bn(g + "g_bn1", GF_DIM * 2 * 7 * 7)
    deconv(g + "g_h2", GF_DIM * 2, GF_DIM * 2 + Y_DIM)
    bn(g + "g_bn2", GF_DIM * 2)
    deconv(g + "g_h3", 1, GF_DIM * 2 + Y_DIM)
    d = "discriminator/"
    if disc_type == "projection":
        cins = [1, DF_DIM, DF_DIM, DF_DIM]
        for i in range(4):
            cin = cins[i] + (Y_DIM if (i + 1) in concat_y_layers else 0)
            conv(d + "d_h%d_conv" % i, cin, DF_DIM, spectral_norm)
            if i > 0:
                bn(d + "d_bn%d" % i, DF_DIM)
        lin(d + "d_h4_lin", DF_DIM, 1)
        lin(d + "d_h5_y_lin", Y_DIM, DF_DIM)
    else:
        conv(d + "d_h0_conv", 1 + Y_DIM, 1 + Y_DIM, False)
        conv(d + "d_h1_conv", 1 + 2 * Y_DIM, DF_DIM + Y_DIM, False)
        bn(d + "d_bn1", DF_DIM + Y_DIM)
        lin(d + "d_h3_lin", 7 * 7 * (DF_DIM + Y_DIM) + Y_DIM, DFC_DIM)
        bn(d + "d_bn2", DFC_DIM)
        lin(d + "d_h4_lin", DFC_DIM + Y_DIM, 1)
    if perm_regularizer:
        lin("classifier/d_classifier_h1", 784, Y_DIM)
    return P, S, U


def is_d_var(name):
    return "d_" in name          # model.py:244 (includes classifier/d_classifier_h1)


def is_g_var(name):
    return "g_" in name          # model.py:245


class Net:
    def __init__(self, P, S, U, train, cfg, dtype=np.float32):
        """train: 'd' | 'g' (which variable set receives gradients)."""
        self.t = Tape()
        self.cfg, self.dtype = cfg, dtype
        self.S = S                       # BN moving stats, updated in place by training-mode BN
        self.U, self.U_read = U, dict(U)
        self.V = {}
        for k, v in P.items():
            req = (train == "d" and is_d_var(k)) or (train == "g" and (is_g_var(k) or k == "confusion_logits"))
            self.V[k] = Var(np.asarray(v, dtype=dtype), req=req, name=k)

    def const(self, a):
        return Var(np.asarray(a, dtype=self.dtype))

    # ---- mnist/ops.py wrappers
    def linear(self, x, scope):
        return self.t.linear(x, self.V[scope + "/Matrix"], self.V[scope + "/bias"])        # ops.py:97-116

    def conv2d(self, x, name, spectral_norm=False):
        w = self.V[name + "/w"]                                                            # ops.py:53-67
        if spectral_norm:
            w = self.t.spectral_norm(w, self.U_read, self.U, name + "/spectral_norm/u", True)
        return self.t.conv2d(x, w, self.V[name + "/biases"], 2)

    def deconv2d(self, x, out_shape, name):
        return self.t.conv2d_transpose(x, self.V[name + "/w"], self.V[name + "/biases"], out_shape, 2)   # ops.py:69-92

    def bn(self, x, name, train=True):
        g, b = self.V[name + "/gamma"], self.V[name + "/beta"]
        if train:
            st = {"moving_mean": self.S[name + "/moving_mean"].astype(self.dtype),
                  "moving_variance": self.S[name + "/moving_variance"].astype(self.dtype)}
            y = self.t.batch_norm_train(x, g, b, st)
            if self.cfg.get("update_moving", True):
                self.S[name + "/moving_mean"], self.S[name + "/moving_variance"] = st["moving_mean"], st["moving_variance"]
            return y
        return self.t.batch_norm_infer(x, g, b, self.S[name + "/moving_mean"].astype(self.dtype),
                                       self.S[name + "/moving_variance"].astype(self.dtype))

    def cond_concat(self, x, y):
        # conv_cond_concat (ops.py:46-51) / concat([h, y], 1)
        yv = np.asarray(y, dtype=self.dtype)
        if x.v.ndim == 4:
            n, h, w, _ = x.v.shape
            yb = np.broadcast_to(yv[:, None, None, :], (n, h, w, yv.shape[1]))
            return self.t.concat([x, self.const(yb)], 3)
        return self.t.concat([x, self.const(yv)], 1)

    # ---- model.py:705-757
    def generator(self, z, y, train=True):
        p = "generator/"
        B = z.shape[0]
        h = self.cond_concat(z if isinstance(z, Var) else self.const(z), y)
        h0 = self.t.relu(self.bn(self.linear(h, p + "g_h0_lin"), p + "g_bn0", train))
        h0 = self.cond_concat(h0, y)
        h1 = self.t.relu(self.bn(self.linear(h0, p + "g_h1_lin"), p + "g_bn1", train))
        h1 = self.t.reshape(h1, (B, 7, 7, GF_DIM * 2))
        h1 = self.cond_concat(h1, y)
        h2 = self.t.relu(self.bn(self.deconv2d(h1, (B, 14, 14, GF_DIM * 2), p + "g_h2"), p + "g_bn2", train))
        h2 = self.cond_concat(h2, y)
        return self.t.sigmoid(self.deconv2d(h2, (B, 28, 28, 1), p + "g_h3"))

    # ---- model.py:644-703
    def discriminator(self, image, y):
        p = "discriminator/"
        B = image.v.shape[0]
        cfg = self.cfg
        if cfg.get("disc_type", "projection") == "projection":
            sn = cfg.get("spectral_norm", True)
            layers = cfg.get("concat_y_layers", ()) if cfg.get("concat_y") else ()
            x = image
            for i in range(4):
                if (i + 1) in layers:
                    x = self.cond_concat(x, y)
                x = self.conv2d(x, p + "d_h%d_conv" % i, spectral_norm=sn)
                if i > 0:
                    x = self.bn(x, p + "d_bn%d" % i)
                x = self.t.lrelu(x)
            h3 = self.t.mean_hw(x)
            h4 = self.linear(h3, p + "d_h4_lin")
            h5 = self.linear(self.const(y), p + "d_h5_y_lin")
            return self.t.add(h4, self.t.sum_axis(self.t.mul(h3, h5), 1, keepdims=True))     # h6 [B,1]
        x = self.cond_concat(image, y)
        h0 = self.t.lrelu(self.conv2d(x, p + "d_h0_conv"))
        h0 = self.cond_concat(h0, y)
        h1 = self.t.lrelu(self.bn(self.conv2d(h0, p + "d_h1_conv"), p + "d_bn1"))
        h1 = self.t.reshape(h1, (B, -1))
        h1 = self.cond_concat(h1, y)
        h3 = self.t.lrelu(self.bn(self.linear(h1, p + "d_h3_lin"), p + "d_bn2"))
        h3 = self.cond_concat(h3, y)
        return self.linear(h3, p + "d_h4_lin")

    def classifier(self, x):
        return self.linear(self.t.reshape(x, (x.v.shape[0], -1)), "classifier/d_classifier_h1")   # model.py:759-768

    # ---- losses (model.py:133-148)
    def loss_real(self, x):
        t = self.t
        if self.cfg.get("loss_fn", "hinge") == "hinge":
            return t.relu(t.add(self.const(1.0), t.scale(x, -1.0)))
        return t.sigmoid_ce(x, np.ones_like(x.v))

    def loss_fake(self, x):
        t = self.t
        if self.cfg.get("loss_fn", "hinge") == "hinge":
            return t.relu(t.add(self.const(1.0), x))
        return t.sigmoid_ce(x, np.zeros_like(x.v))

    def loss_g(self, x):
        t = self.t
        if self.cfg.get("loss_fn", "hinge") == "hinge":
            return t.scale(x, -1.0)
        return t.sigmoid_ce(x, np.ones_like(x.v))

    def confusion(self):
        if "confusion_logits" in self.V:
            return self.t.softmax_rows(self.V["confusion_logits"])
        return self.const(self.cfg["C"])


def _eye_rows(i, B, dtype):
    y = np.zeros((B, Y_DIM), dtype)
    y[:, i] = 1
    return y


def losses(net, batch, real=True):
    """build_model (model.py:126-224) -> dict of scalar Vars: d_loss_real, d_loss_fake, g_loss,
    class_loss_real, class_loss_fake.  batch: images [B,28,28,1], z, y_real, y_fake, y_gen, y_real_weights.
    real=False skips the real-data branch (the G-step run fetches only g_optim/c_optim/g_sum, model.py:359-372)."""
    t, cfg = net.t, net.cfg
    alg = cfg["algorithm"]
    B = batch["images"].shape[0]
    dt = net.dtype
    G = net.generator(np.asarray(batch["z"], dt), np.asarray(batch["y_gen"], dt))
    inputs = net.const(batch["images"])
    out = {}
    if not real:
        pass
    elif alg in ("biased", "rcgan", "ambient"):
        out["d_loss_real"] = t.mean_all(net.loss_real(net.discriminator(inputs, batch["y_real"])))
    elif alg == "unbiased":
        cols = [net.loss_real(net.discriminator(inputs, _eye_rows(i, B, dt))) for i in range(Y_DIM)]
        allc = t.concat(cols, 1)
        out["d_loss_real"] = t.mean_all(t.sum_axis(t.mul(allc, net.const(batch["y_real_weights"])), 1))
    else:
        raise ValueError(alg)
    if alg in ("rcgan", "ambient") and cfg.get("estimate_confuse"):
        lf, lg = [], []
        for i in range(Y_DIM):                      # the reference re-runs the whole D for every label (model.py:187-197)
            lo = net.discriminator(G, _eye_rows(i, B, dt))
            lf.append(net.loss_fake(lo))
            lg.append(net.loss_g(lo))
        yc = t.linear(net.const(batch["y_gen"]), net.confusion())       # tensordot(y_gen, C)
        out["d_loss_fake"] = t.mean_all(t.sum_axis(t.mul(t.concat(lf, 1), yc), 1))
        out["g_loss"] = t.mean_all(t.sum_axis(t.mul(t.concat(lg, 1), yc), 1))
    else:
        ylab = batch["y_fake"] if alg in ("rcgan", "ambient") else batch["y_gen"]
        lo = net.discriminator(G, ylab)
        out["d_loss_fake"] = t.mean_all(net.loss_fake(lo))
        out["g_loss"] = t.mean_all(net.loss_g(lo))
    if cfg.get("perm_regularizer", True):
        if real:
            out["class_loss_real"] = t.sigmoid_ce_mean(net.classifier(inputs), np.asarray(batch["y_real"], dt))
        out["class_loss_fake"] = t.sigmoid_ce_mean(net.classifier(G), np.asarray(batch["y_gen"], dt))
    return out


def d_grads(P, S, U, cfg, batch, dtype=np.float32):
    """d_optim objective: d_loss + class_loss_real over d_vars (model.py:250-253)."""
    net = Net(P, S, U, "d", cfg, dtype)
    L = losses(net, batch)
    total = net.t.add(L["d_loss_real"], L["d_loss_fake"])
    if "class_loss_real" in L:
        total = net.t.add(total, L["class_loss_real"])
    net.t.backward(total)
    grads = {k: v.g for k, v in net.V.items() if is_d_var(k) and v.g is not None}
    return {k: float(v.v) for k, v in L.items()}, grads


def g_grads(P, S, U, cfg, batch, dtype=np.float32):
    """g_optim objective: g_loss + perm_multiplier*class_loss_fake over g_vars; c_optim: g_loss over
    confusion_logits (model.py:254-262)."""
    net = Net(P, S, U, "g", cfg, dtype)
    L = losses(net, batch, real=False)
    total = L["g_loss"]
    if "class_loss_fake" in L:
        total = net.t.add(total, net.t.scale(L["class_loss_fake"], cfg.get("perm_multiplier", 10.0)))
    net.t.backward(total)
    grads = {k: v.g for k, v in net.V.items() if (is_g_var(k) or k == "confusion_logits") and v.g is not None}
    return {k: float(v.v) for k, v in L.items()}, grads


class AdamState:
    def __init__(self):
        self.m, self.v, self.t = {}, {}, 0


def apply_adam(P, grads, st, lr, beta1, clip_names=()):
    st.t += 1
    for k, g in grads.items():
        if k not in st.m:
            st.m[k] = np.zeros_like(P[k])
            st.v[k] = np.zeros_like(P[k])
        clip = 1.0 if any(k.startswith(c) for c in clip_names) else None      # variable constraint, ops.py:102-111
        P[k], st.m[k], st.v[k] = nn.adam_tf(P[k], g.astype(P[k].dtype), st.m[k], st.v[k], st.t, lr, beta1, 0.999, clip=clip)


class Trainer:
    """One reference iteration = 1 D run + 2 G runs on the same batch (model.py:347-372)."""

    def __init__(self, P, S, U, cfg, lr=2e-4, beta1=0.5):
        self.P, self.S, self.U, self.cfg, self.lr, self.beta1 = P, S, U, cfg, lr, beta1
        self.ad, self.ag, self.ac = AdamState(), AdamState(), AdamState()
        self.clip = ("discriminator/d_h4_lin", "discriminator/d_h5_y_lin") if cfg.get("max_norm", True) and \
            cfg.get("disc_type", "projection") == "projection" else ()

    def d_step(self, batch):
        L, g = d_grads(self.P, self.S, self.U, self.cfg, batch)
        apply_adam(self.P, g, self.ad, self.lr, self.beta1, self.clip)
        return L, g

    def g_step(self, batch):
        L, g = g_grads(self.P, self.S, self.U, self.cfg, batch)
        gc = {k: g.pop(k) for k in list(g) if k == "confusion_logits"}
        apply_adam(self.P, g, self.ag, self.lr, self.beta1)
        if gc:
            apply_adam(self.P, gc, self.ac, self.lr * self.cfg.get("confuse_multiplier", 10.0), self.beta1)
        g.update(gc)
        return L, g

    def iteration(self, batch):
        out = {}
        out["d"], _ = self.d_step(batch)
        out["g1"], _ = self.g_step(batch)
        out["g2"], _ = self.g_step(batch)
        return out


def sampler(P, S, z, y, dtype=np.float32):
    """gen_sampler (model.py:733-757): inference-mode BN."""
    net = Net(P, S, {}, "none", {}, dtype)
    return net.generator(np.asarray(z, dtype), np.asarray(y, dtype), train=False).v


def recover_step(P, S, z, logits, actual, lr, dtype=np.float64):
    """One iteration of DCGAN.recover_labels (mnist/model.py:519-537, 606-630): z [R*10,100], logits [R,10], actual [R,784].
    loss = mean_r sum_y softmax(logits)[r,y] * mean_pix (actual[r] - sampler(z[r,y], onehot y))^2; plain SGD on z and logits.
    Returns (loss, new z, new logits, y_recover)."""
    net = Net(P, S, {}, "none", {}, dtype)
    t = net.t
    R = logits.shape[0]
    zv, lv = Var(np.asarray(z, dtype), req=True), Var(np.asarray(logits, dtype), req=True)
    hard_y = np.tile(np.eye(Y_DIM, dtype=dtype), (R, 1))
    G = net.generator(zv, hard_y, train=False)                                  # [R*10,28,28,1]
    yrec = t.softmax_rows(lv)
    diff = t.add(t.reshape(G, (R, Y_DIM, -1)), Var(-np.asarray(actual, dtype).reshape(R, 1, -1)))
    sq = t.scale(t.sum_axis(t.mul(diff, diff), 2), 1.0 / diff.v.shape[2])      # [R,10]
    loss = t.scale(t.sum_axis(t.sum_axis(t.mul(sq, yrec), 1), 0), 1.0 / R)
    t.backward(loss)
    return float(loss.v), zv.v - lr * zv.g, lv.v - lr * lv.g, yrec.v
