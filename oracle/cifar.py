"""numpy restatement of the CIFAR-10 SNGAN-projection RCGAN step (oracle; test infrastructure).

Follows /root/reference/cifar10/gan_resnet.py (model :199-421,458-483; losses :557-695,715-786;
optimisers :700-705,802-817; loop :919-947) and cifar10/common/ops/{conv2d,linear,normalization,
embedding,sn}.py.  Parity for this floating-point path is UNPINNED by the reference (no TF here, no
reference tests) -- see oracle/__init__.py.
"""
import numpy as np
from . import nn
from .tape import Tape, Var

Z_DIM = 128
DIM_G = 128
DIM_D = 128
VOCAB = 10
EMB_DIM = 300
IMG = 32
OUTPUT_DIM = 3072
N_CRITIC = 5
GEN_BS_MULTIPLE = 2


# ------------------------------------------------------------------------------------------
# parameter creation in the reference's variable-creation order
# ------------------------------------------------------------------------------------------
class _Streams(np.random.RandomState):
    """numpy's stream (the reference's numpy initialisers) + ``.tf``: a separate stream standing in for TensorFlow's generators."""

    def __init__(self, seed):
        super().__init__(seed)
        self.tf = np.random.RandomState((int(seed) + 0x7F4A7C15) % (1 << 32))


def _uniform(rs, stdev, size):
    # conv2d.py:83-88 / linear.py:54-61
    return rs.uniform(low=-stdev * np.sqrt(3), high=stdev * np.sqrt(3), size=size).astype("float32")


def _trunc_normal(rs, size, stddev=1.0):
    # tf.truncated_normal_initializer(): resample beyond 2 sigma (sn.py:36)
    x = rs.normal(0.0, stddev, size=size)
    bad = np.abs(x) > 2 * stddev
    while bad.any():
        x[bad] = rs.normal(0.0, stddev, size=int(bad.sum()))
        bad = np.abs(x) > 2 * stddev
    return x.astype("float32")


def _conv_params(P, U, rs, name, cin, cout, k, he_init, sn):
    """One Conv2D call (conv2d.py:31-218).  P None: the variables exist already (reuse=True) -- the numpy draw of the initial
    filter values still happens (conv2d.py:118-140 compute filter_values before tf.get_variable) and is thrown away."""
    fan_in = cin * k * k
    fan_out = cout * k * k
    stdev = np.sqrt(4. / (fan_in + fan_out)) if he_init else np.sqrt(2. / (fan_in + fan_out))
    w = _uniform(rs, stdev, (k, k, cin, cout))
    if P is None:
        return
    P[name + "/Filters"] = w
    if sn:
        U[name + "/filters/spectral_norm/u"] = _trunc_normal(rs.tf, (1, cout))
    P[name + "/Biases"] = np.zeros((cout,), "float32")


def _linear_params(P, U, rs, name, cin, cout, sn):
    w = _uniform(rs, np.sqrt(2. / (cin + cout)), (cin, cout))   # linear.py:76-80 (first matching branch); drawn on every call
    if P is None:
        return
    P[name + "/W"] = w
    if sn:
        U[name + "/spectral_norm/u"] = _trunc_normal(rs.tf, (1, cout))
    P[name + "/b"] = np.zeros((cout,), "float32")


def _condbn_params(P, name, c):
    if P is None:
        return
    P[name + "/CondBatchNorm/offset"] = np.zeros((VOCAB, c), "float32")
    P[name + "/CondBatchNorm/scale"] = np.ones((VOCAB, c), "float32")


def confusion_logits_init(confuse_init, confuse_init_diag=0.2, rs=None):
    """gan_resnet.py:499-520."""
    if not confuse_init:
        # TF default initializer for get_variable: glorot_uniform
        lim = np.sqrt(6.0 / (VOCAB + VOCAB))
        return (rs.tf if rs is not None else np.random.RandomState(0)).uniform(-lim, lim, size=(VOCAB, VOCAB)).astype("float32")
    if confuse_init_diag > 0.99 and VOCAB == 10.:
        aa = 7.0
    else:
        aa = np.log(VOCAB * confuse_init_diag / (1. - confuse_init_diag))
    aa = min(7.0, aa)
    c = (0 - aa / VOCAB) * np.ones([VOCAB, VOCAB], dtype=np.float32)
    np.fill_diagonal(c, (aa - (aa / VOCAB)))
    return c


def _generator_call(P, U, rs):                 # gan_resnet.py:356-371
    _linear_params(P, U, rs, "Generator/G.Input", Z_DIM, 4 * 4 * DIM_G * 8, sn=False)
    for k, cin in ((1, DIM_G * 8), (2, DIM_G * 2), (3, DIM_G * 2)):
        nm = "Generator/G.Block.%d" % k
        _conv_params(P, U, rs, nm + ".Shortcut", cin, DIM_G * 2, 1, he_init=False, sn=False)
        _condbn_params(P, nm + ".N1", cin)
        _conv_params(P, U, rs, nm + ".Conv1", cin, DIM_G * 2, 3, he_init=True, sn=False)
        _condbn_params(P, nm + ".N2", DIM_G * 2)
        _conv_params(P, U, rs, nm + ".Conv2", DIM_G * 2, DIM_G * 2, 3, he_init=True, sn=False)
    _condbn_params(P, "Generator/G.OutputNorm", DIM_G * 2)
    _conv_params(P, U, rs, "Generator/G.Output", DIM_G * 2, 3, 3, he_init=False, sn=False)


def _discriminator_call(P, U, rs):             # gan_resnet.py:331-353, 374-412
    d = "Discriminator/"
    _conv_params(P, U, rs, d + "D.Block.1.Shortcut", 3, DIM_D, 1, he_init=False, sn=True)
    _conv_params(P, U, rs, d + "D.Block.1.Conv1", 3, DIM_D, 3, he_init=True, sn=True)
    _conv_params(P, U, rs, d + "D.Block.1.Conv2", DIM_D, DIM_D, 3, he_init=True, sn=True)
    _conv_params(P, U, rs, d + "D.Block.2.Shortcut", DIM_D, DIM_D, 1, he_init=False, sn=True)
    _conv_params(P, U, rs, d + "D.Block.2.Conv1", DIM_D, DIM_D, 3, he_init=True, sn=True)
    _conv_params(P, U, rs, d + "D.Block.2.Conv2", DIM_D, DIM_D, 3, he_init=True, sn=True)
    for k in (3, 4, 5, 6):
        _conv_params(P, U, rs, d + "D.Block.%d.Conv1" % k, DIM_D, DIM_D, 3, he_init=True, sn=True)
        _conv_params(P, U, rs, d + "D.Block.%d.Conv2" % k, DIM_D, DIM_D, 3, he_init=True, sn=True)
    _linear_params(P, U, rs, d + "D.Output", DIM_D, 1, sn=True)


def _projection_call(P, U, rs):                # gan_resnet.py:414-421, embedding.py:27-40
    d = "Discriminator/"
    table = rs.uniform(-0.08, 0.08, size=(VOCAB, EMB_DIM)).astype("float32")
    if P is not None:
        P[d + "Embedding.Label/embedding_map"] = table
    _linear_params(P, U, rs, d + "D.Embedding_y", EMB_DIM, DIM_D, sn=True)


N_TOWERS = 2       # len(DEVICES), gan_resnet.py:186-188


def init_params(seed=0, algorithm="rcgan", perm_classifier=False, perm_type="linear",
                confuse_init=False, confuse_init_diag=0.2):
    """Returns (P, U): trainable params and non-trainable SN ``u`` vectors, dict name -> float32 array.
    The reference draws its filters / matrices / embedding table from numpy's global stream (conv2d.py:83-88, linear.py:54-61,
    embedding.py:29-34) while it BUILDS THE GRAPH, one draw per Conv2D / Linear / embed_y call -- also for calls that reuse
    existing variables (second tower's Generator :541-546, the ten extra projections of 'unbiased' :615-622, the second
    Discriminator + projection of 'rcgan-u' :654-657), whose values are thrown away but move the stream.  Following that call
    order reproduces the reference's initial values bit for bit from ``np.random.seed(seed)`` (tests/golden/ref_cifar_*.npz,
    generated by running the reference's own main()).  The spectral-norm ``u`` vectors (sn.py:36) and a default-initialised
    confusion matrix (gan_resnet.py:501-503) come from TensorFlow's own generators: a SECOND stream."""
    rs = _Streams(seed)
    P, U = {}, {}
    if algorithm == "rcgan-u":
        P["confusion_logits"] = confusion_logits_init(confuse_init, confuse_init_diag, rs)
    _generator_call(P, U, rs)
    for _ in range(N_TOWERS - 1):
        _generator_call(None, None, rs)
    _discriminator_call(P, U, rs)
    _projection_call(P, U, rs)
    if algorithm == "unbiased":
        for _ in range(VOCAB):
            _projection_call(None, None, rs)
    elif algorithm == "rcgan-u":
        _discriminator_call(None, None, rs)
        _projection_call(None, None, rs)
    d = "Discriminator/"
    if perm_classifier:
        if perm_type == "linear":
            _linear_params(P, U, rs, d + "D.d_perm_classifier_h1", OUTPUT_DIM, VOCAB, sn=True)
        elif perm_type == "2layer":
            _linear_params(P, U, rs, d + "D.d_perm_classifier_h1", OUTPUT_DIM, 128, sn=True)
            _linear_params(P, U, rs, d + "D.d_perm_classifier_h2", 128, VOCAB, sn=True)
        else:
            raise ValueError("Unknown perm_type {}".format(perm_type))
    return P, U


def c_alpha(alpha):
    """One-coin confusion matrix (gan_resnet.py:106)."""
    return ((1 - alpha) / 9.0) * np.ones((10, 10)) + (alpha - (1 - alpha) / 9.0) * np.eye(10)


# ------------------------------------------------------------------------------------------
# model (tape-level restatement; function names follow the reference)
# ------------------------------------------------------------------------------------------
class Net:
    """Holds the tape, parameter Vars and SN state for one graph execution."""

    def __init__(self, P, U, train_g, train_d, dtype=np.float32):
        self.t = Tape()
        self.U = U                 # written by SN updates
        self.U_read = dict(U)      # read by every SN evaluation of this step
        self.dtype = dtype
        self.V = {}
        for k, v in P.items():
            req = (k.startswith("Generator") and train_g) or (k.startswith("Discriminator") and train_d) \
                or (k == "confusion_logits" and train_g)
            self.V[k] = Var(np.asarray(v, dtype=dtype), req=req, name=k)

    def const(self, a):
        return Var(np.asarray(a, dtype=self.dtype))

    # lib.ops.conv2d.Conv2D (conv2d.py:31-218): stride 1 SAME, optional SN in scope 'filters/', bias_add
    def Conv2D(self, x, name, sn=False, update=True):
        w = self.V[name + "/Filters"]
        if sn:
            w = self.t.spectral_norm(w, self.U_read, self.U, name + "/filters/spectral_norm/u", update)
        return self.t.conv2d(x, w, self.V[name + "/Biases"], 1)

    # lib.ops.linear.Linear (linear.py:38-182)
    def Linear(self, x, name, sn=False, update=True):
        w = self.V[name + "/W"]
        if sn:
            w = self.t.spectral_norm(w, self.U_read, self.U, name + "/spectral_norm/u", update)
        return self.t.linear(x, w, self.V[name + "/b"])

    def Normalize(self, name, x, labels):
        # gan_resnet.py:207-228 with CONDITIONAL=True, NORMALIZATION_G=True, NORMALIZATION_D=False
        if "G." in name and labels is not None:
            return self.t.cond_batchnorm(x, labels, self.V[name + "/CondBatchNorm/scale"],
                                         self.V[name + "/CondBatchNorm/offset"])
        return x

    def ConvMeanPool(self, x, name, **kw):      # gan_resnet.py:231-241
        return self.t.meanpool2(self.Conv2D(x, name, **kw))

    def MeanPoolConv(self, x, name, **kw):      # gan_resnet.py:244-256
        return self.Conv2D(self.t.meanpool2(x), name, **kw)

    def UpsampleConv(self, x, name, **kw):      # gan_resnet.py:259-272
        return self.Conv2D(self.t.upsample2(x), name, **kw)

    def ResidualBlock(self, x, in_dim, out_dim, name, resample, labels, **kw):   # gan_resnet.py:275-328
        if resample == "down":
            conv_1, conv_2, conv_sc = self.Conv2D, self.ConvMeanPool, self.ConvMeanPool
        elif resample == "up":
            conv_1, conv_2, conv_sc = self.UpsampleConv, self.Conv2D, self.UpsampleConv
        elif resample is None:
            conv_1, conv_2, conv_sc = self.Conv2D, self.Conv2D, self.Conv2D
        else:
            raise Exception("invalid resample value")
        if out_dim == in_dim and resample is None:
            shortcut = x
        else:
            shortcut = conv_sc(x, name + ".Shortcut", **kw)
        out = self.Normalize(name + ".N1", x, labels)
        out = self.t.relu(out)
        out = conv_1(out, name + ".Conv1", **kw)
        out = self.Normalize(name + ".N2", out, labels)
        out = self.t.relu(out)
        out = conv_2(out, name + ".Conv2", **kw)
        return self.t.add(shortcut, out)

    def OptimizedResBlockDisc1(self, x, **kw):   # gan_resnet.py:331-353
        p = "Discriminator/"
        shortcut = self.MeanPoolConv(x, p + "D.Block.1.Shortcut", **kw)
        out = self.Conv2D(x, p + "D.Block.1.Conv1", **kw)
        out = self.t.relu(out)
        out = self.ConvMeanPool(out, p + "D.Block.1.Conv2", **kw)
        return self.t.add(shortcut, out)

    def Generator(self, labels, noise):          # gan_resnet.py:356-371
        p = "Generator/"
        out = self.Linear(noise, p + "G.Input")
        out = self.t.reshape(out, (-1, 4, 4, DIM_G * 8))
        out = self.ResidualBlock(out, DIM_G * 8, DIM_G * 2, p + "G.Block.1", "up", labels)
        out = self.ResidualBlock(out, DIM_G * 2, DIM_G * 2, p + "G.Block.2", "up", labels)
        out = self.ResidualBlock(out, DIM_G * 2, DIM_G * 2, p + "G.Block.3", "up", labels)
        out = self.Normalize(p + "G.OutputNorm", out, labels)
        out = self.t.relu(out)
        out = self.Conv2D(out, p + "G.Output")
        out = self.t.tanh(out)
        return self.t.reshape(out, (-1, OUTPUT_DIM))

    def Discriminator(self, inputs, update):     # gan_resnet.py:374-412  (labels unused: NORMALIZATION_D=False)
        p = "Discriminator/"
        kw = dict(sn=True, update=update)
        out = self.t.reshape(inputs, (-1, IMG, IMG, 3))
        out = self.OptimizedResBlockDisc1(out, **kw)
        out = self.ResidualBlock(out, DIM_D, DIM_D, p + "D.Block.2", "down", None, **kw)
        for k in (3, 4, 5, 6):
            out = self.ResidualBlock(out, DIM_D, DIM_D, p + "D.Block.%d" % k, None, None, **kw)
        out = self.t.relu(out)
        out = self.t.mean_hw(out)
        wgan = self.Linear(out, p + "D.Output", sn=True, update=update)
        return out, self.t.reshape(wgan, (-1,))

    def Discriminator_projection(self, labels, update=True):   # gan_resnet.py:414-421 (always update_collection=None)
        p = "Discriminator/"
        e = self.t.gather_rows(self.V[p + "Embedding.Label/embedding_map"], labels)     # embedding.py:12-51
        return self.Linear(e, p + "D.Embedding_y", sn=True, update=update)

    def perm_classifier(self, x, perm_type="linear", update=True):   # gan_resnet.py:458-483
        p = "Discriminator/"
        h = self.Linear(self.t.reshape(x, (-1, OUTPUT_DIM)), p + "D.d_perm_classifier_h1", sn=True, update=update)
        if perm_type == "2layer":
            h = self.Linear(h, p + "D.d_perm_classifier_h2", sn=True, update=update)
        return h

    def proj_logit(self, feat, wgan, emb):
        # output_wgan + reduce_sum(output*embedding_y, axis=1)   (gan_resnet.py:588)
        return self.t.add(wgan, self.t.sum_axis(self.t.mul(feat, emb), 1))

    def all_label_logits(self, feat, wgan):
        # disc_fake[n,10] = wgan[:,None] + sum(feat[:,None,:]*E[None,:,:], -1)   (gan_resnet.py:654-660)
        emb = self.Discriminator_projection(np.arange(VOCAB))
        f3 = self.t.reshape(feat, (feat.v.shape[0], 1, feat.v.shape[1]))
        e3 = self.t.reshape(emb, (1, VOCAB, emb.v.shape[1]))
        return self.t.add(self.t.reshape(wgan, (-1, 1)), self.t.sum_axis(self.t.mul(f3, e3), 2))

    def confusion_matrix(self, C_const):
        if "confusion_logits" in self.V:
            return self.t.softmax_rows(self.V["confusion_logits"])     # gan_resnet.py:522
        return self.const(C_const)                                   # gan_resnet.py:524


def preprocess_real(images_u8, noise):
    """gan_resnet.py:548-551: 2*(x/256-.5) + U[0,1/128) in CHW order, then CHW->HWC, flattened."""
    x = 2 * ((images_u8.astype(np.float32) / 256.) - .5)
    x = x + noise.astype(np.float32)
    n = x.shape[0]
    return x.reshape(n, 3, IMG, IMG).transpose(0, 2, 3, 1).reshape(n, OUTPUT_DIM)


def _onehot(idx, dtype):
    o = np.zeros((len(idx), VOCAB), dtype=dtype)
    o[np.arange(len(idx)), np.asarray(idx, dtype=np.int64)] = 1
    return o


def disc_cost_tower(net, cfg, real, labels, labels_random, labels_biased, inv_weights, z):
    """One device tower of the discriminator cost (gan_resnet.py:557-695).  HINGE, SOFT_PLUS=False."""
    t = net.t
    alg = cfg["algorithm"]
    B = real.shape[0]
    fake = net.Generator(labels_random, net.const(z))             # :540-546
    real_v = net.const(real)
    if alg == "rcgan-u":
        feat, wgan = net.Discriminator(real_v, update=True)
        emb = net.Discriminator_projection(labels)
        disc_real = net.proj_logit(feat, wgan, emb)
        feat_f, wgan_f = net.Discriminator(fake, update=True)
        disc_fake = net.all_label_logits(feat_f, wgan_f)          # [B,10]
        disc_fake_y = t.relu(t.add(net.const(1.0), disc_fake))
        disc_real_l = t.mean_all(t.relu(t.add(net.const(1.0), t.scale(disc_real, -1.0))))
        C = net.confusion_matrix(cfg.get("C"))
        y_conf = t.gather_rows(C, labels_random)                  # one_hot(labels_random) . C  (:682-683)
        abc = t.mean_all(t.sum_axis(t.mul(disc_fake_y, y_conf), 1))
        cost = t.add(abc, disc_real_l)
    else:
        x = t.concat([real_v, fake], 0)                           # :563-566
        feat, wgan = net.Discriminator(x, update=True)
        if alg in ("biased", "unbiased"):
            lab = np.concatenate([labels, labels_random])
        else:                                                     # rcgan: real noisy labels + labels_biased (:575-578)
            lab = np.concatenate([labels, labels_biased])
        if alg in ("biased", "rcgan"):
            emb = net.Discriminator_projection(lab)
            disc_all = net.proj_logit(feat, wgan, emb)
            disc_real = t.rows(disc_all, 0, B)
            disc_fake = t.rows(disc_all, B, 2 * B)
            disc_real_l = t.mean_all(t.relu(t.add(net.const(1.0), t.scale(disc_real, -1.0))))
            disc_fake_l = t.mean_all(t.relu(t.add(net.const(1.0), disc_fake)))
            cost = t.add(disc_real_l, disc_fake_l)                # :604-606
        elif alg == "unbiased":
            emb = net.Discriminator_projection(lab)               # :585 (created, result unused by the loss)
            cols = []
            disc_fake_l = None
            for j in range(VOCAB):                                # :615-646
                lab_j = np.concatenate([j * np.ones((B,), np.int64), labels_random])
                emb_j = net.Discriminator_projection(lab_j)
                disc_all = net.proj_logit(feat, wgan, emb_j)
                disc_real = t.rows(disc_all, 0, B)
                disc_fake = t.rows(disc_all, B, 2 * B)
                cols.append(t.reshape(t.relu(t.add(net.const(1.0), t.scale(disc_real, -1.0))), (B, 1)))
                disc_fake_l = t.mean_all(t.relu(t.add(net.const(1.0), disc_fake)))
            allc = t.concat(cols, 1)
            abc = t.mean_all(t.sum_axis(t.mul(allc, net.const(inv_weights)), 1))
            cost = t.add(abc, disc_fake_l)                        # :647-648 (only the LAST j's fake loss survives)
        else:
            raise ValueError(alg)
    if cfg.get("perm_classifier"):
        logits = net.perm_classifier(real_v, cfg.get("perm_type", "linear"))
        pl = t.sigmoid_ce_mean(logits, _onehot(labels, net.dtype))       # :692-695
        cost = t.add(cost, pl)
    return cost


def gen_cost_tower(net, cfg, labels_random_G, labels_biased_G, z):
    """One device tower of the generator cost (gan_resnet.py:715-786)."""
    t = net.t
    alg = cfg["algorithm"]
    fake = net.Generator(labels_random_G, net.const(z))
    feat, wgan = net.Discriminator(fake, update=False)            # update_collection="NO_OPS" (:723,729)
    lab = labels_random_G if alg in ("biased", "unbiased") else labels_biased_G
    emb = net.Discriminator_projection(lab, update=True)          # update_collection=None (:725,731)
    if alg == "rcgan-u":
        disc_fake = net.all_label_logits(feat, wgan)
        disc_fake_y = t.scale(disc_fake, -1.0)
        C = net.confusion_matrix(cfg.get("C"))
        y_conf = t.gather_rows(C, labels_random_G)
        cost = t.mean_all(t.sum_axis(t.mul(disc_fake_y, y_conf), 1))        # :757-760
    else:
        disc_fake = net.proj_logit(feat, wgan, emb)
        cost = t.scale(t.mean_all(disc_fake), -1.0)               # :773
    if cfg.get("perm_classifier"):
        logits = net.perm_classifier(fake, cfg.get("perm_type", "linear"))
        pl = t.sigmoid_ce_mean(logits, _onehot(labels_random_G, net.dtype))  # :781-784
        cost = t.add(cost, t.scale(pl, cfg.get("perm_multiplier", 1.0)))
    return cost


def _split(a, n):
    return np.split(np.asarray(a), n, axis=0)


def d_grads(P, U, cfg, batch, ntowers=1, dtype=np.float32):
    """disc_cost = mean over towers (:697) and its gradient w.r.t. Discriminator params.
    batch: dict(real[B,3072] float (already preprocessed), labels, labels_random, labels_biased,
    inv_weights[B,10], z[B,128]).  Mutates U (SN u update, update_collection=None)."""
    net = Net(P, U, train_g=False, train_d=True, dtype=dtype)
    costs = []
    keys = ("real", "labels", "labels_random", "labels_biased", "inv_weights", "z")
    parts = {k: _split(batch[k], ntowers) for k in keys}
    for i in range(ntowers):      # every tower reads the same u (shared variable, one logical update)
        costs.append(disc_cost_tower(net, cfg, *[parts[k][i] for k in keys]))
    total = costs[0]
    for c in costs[1:]:
        total = net.t.add(total, c)
    total = net.t.scale(total, 1.0 / ntowers)
    net.t.backward(total)
    grads = {k: v.g for k, v in net.V.items() if k.startswith("Discriminator") and v.g is not None}
    return float(total.v), grads


def g_grads(P, U, cfg, batch, ntowers=1, dtype=np.float32):
    """gen_cost (:786) and its gradient w.r.t. Generator params (+ confusion_logits for rcgan-u, :816-817).
    batch: dict(labels_random_G[2B], labels_biased_G[2B], z[2B,128])."""
    net = Net(P, U, train_g=True, train_d=False, dtype=dtype)
    keys = ("labels_random_G", "labels_biased_G", "z")
    parts = {k: _split(batch[k], ntowers) for k in keys}
    costs = []
    for i in range(ntowers):
        costs.append(gen_cost_tower(net, cfg, *[parts[k][i] for k in keys]))
    total = costs[0]
    for c in costs[1:]:
        total = net.t.add(total, c)
    total = net.t.scale(total, 1.0 / ntowers)
    net.t.backward(total)
    grads = {k: v.g for k, v in net.V.items()
             if (k.startswith("Generator") or k == "confusion_logits") and v.g is not None}
    return float(total.v), grads


def lr_decay(iteration):
    """gan_resnet.py:700-705 (DECAY=True)."""
    return max(0., 1. - iteration / 100000.) if iteration < 50000 else 0.5


class AdamState:
    """tf.train.AdamOptimizer slots for one optimiser (beta powers tracked through t)."""

    def __init__(self):
        self.m, self.v, self.t = {}, {}, 0


def apply_adam(P, grads, st, lr, beta1=0.0, beta2=0.9):
    st.t += 1
    for k, g in grads.items():
        if k not in st.m:
            st.m[k] = np.zeros_like(P[k])
            st.v[k] = np.zeros_like(P[k])
        P[k], st.m[k], st.v[k] = nn.adam_tf(P[k], g.astype(P[k].dtype), st.m[k], st.v[k], st.t, lr, beta1, beta2)


class Trainer:
    """Step order of gan_resnet.py:919-947: [G step (+C step) if it>0] then N_CRITIC D steps."""

    def __init__(self, P, U, cfg, lr=2e-4, ntowers=1):
        self.P, self.U, self.cfg, self.lr, self.ntowers = P, U, cfg, lr, ntowers
        self.adam_d, self.adam_g, self.adam_c = AdamState(), AdamState(), AdamState()

    def d_step(self, iteration, batch):
        cost, grads = d_grads(self.P, self.U, self.cfg, batch, self.ntowers)
        apply_adam(self.P, grads, self.adam_d, self.lr * lr_decay(iteration))
        return cost, grads

    def g_step(self, iteration, batch):
        cost, grads = g_grads(self.P, self.U, self.cfg, batch, self.ntowers)
        gc = {k: grads.pop(k) for k in list(grads) if k == "confusion_logits"}
        apply_adam(self.P, grads, self.adam_g, self.lr * lr_decay(iteration))
        if gc:
            clr = self.lr * self.cfg.get("confuse_multiplier", 1.0)
            if self.cfg.get("confuse_lr_decay"):
                clr *= lr_decay(iteration)
            apply_adam(self.P, gc, self.adam_c, clr)
        grads.update(gc)
        return cost, grads

    def iteration(self, it, g_batch, d_batches):
        out = {}
        if it > 0:
            out["g_cost"], _ = self.g_step(it, g_batch)
        for b in d_batches:
            out["d_cost"], _ = self.d_step(it, b)
        return out
