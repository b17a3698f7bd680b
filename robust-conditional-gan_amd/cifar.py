"""CIFAR-10 SNGAN-projection RCGAN / RCGAN-U on the gfx950 kernels: model, losses, step functions.

Host-side mirror of /root/reference cifar10/gan_resnet.py: the block/model functions keep the
reference's names and structure (:199-421, :458-483), the loss assembly follows :557-695 / :715-786,
the optimisers :700-705 / :802-817 and the step order :919-947.  What differs is *how* it runs:
one process per GPU, eager hand-written HIP kernels recorded on a tape, the whole D-step / G-step
captured into a hipGraph, gradients all-reduced over RCCL on flat slabs.
"""
import ctypes as C
import os

import warnings

import numpy as np
import torch

from . import _lib as L
from . import ops as O
from .ops_cifar import NO_OPS, Conv2D, Linear, cond_batchnorm, embed_y, spectral_normed_weight
from .runtime import DT, Context, ParamGroup
from .variables import Graph, scoped, variable_scope

Z_DIM = 128
DIM_G = 128
DIM_D = 128
VOCAB_SIZE = 10
EMBEDDING_DIM = 300
IMG_SIZE = 32
IMG_DIM = 3
OUTPUT_DIM = 3072
N_CRITIC = 5
GEN_BS_MULTIPLE = 2
ALGORITHMS = ("rcgan", "rcgan-u", "biased", "unbiased")
# The 8x8 discriminator stage -- D.Block.3 .. D.Block.6, eight 3x3 convolutions -- as one launch each way (ops.d_trunk /
# conv_trunk.hip: a workgroup carries one image through all eight layers, activations in LDS, each wavefront's filters of a whole
# layer in registers, filters re-laid fragment-major once per step).  Round 3, MI355X, n = 128: 49.6 us forward / 51.8 us data
# gradient against 68.2 / 71.4 us for the eight launches (n = 256: 54 vs 113 us).  RCGAN_FUSED_TRUNK=0 restores the layer-wise blocks.
FUSED_TRUNK = os.environ.get("RCGAN_FUSED_TRUNK", "1") == "1"
# the discriminator's relu + spatial mean comes out of the stage's launch, and its gradient goes back into the backward launch
# (rcgan_dtrunk_pooled): the projection head runs on [n, 128] features.  RCGAN_POOL_IN_TRUNK=0: the head pools by itself.
POOL_IN_TRUNK = os.environ.get("RCGAN_POOL_IN_TRUNK", "1") == "1"


# ------------------------------------------------------------------------------------------------------
# variable creation (reference initialisers, reference creation order -- SURVEY.md Appendix A)
# ------------------------------------------------------------------------------------------------------
class _Init:
    def __init__(self, seed):
        # rs: numpy's stream -- the reference draws its filters / matrices / embedding table from np.random in graph-construction
        # order (conv2d.py:83-88, linear.py:54-61, embedding.py:29-34), reproduced here bit for bit from np.random.seed(seed)
        # (tests/golden/ref_cifar_*.npz).  rs_tf: what TensorFlow's own generators initialise (the spectral-norm u vectors, sn.py:36;
        # a default-initialised confusion matrix, gan_resnet.py:501-503) comes from a second stream and leaves the first alone.
        self.rs = np.random.RandomState(seed)
        self.rs_tf = np.random.RandomState((int(seed) + 0x7F4A7C15) % (1 << 32))
        self.G, self.D, self.Cm, self.U = [], [], [], {}

    def uniform(self, stdev, shape):          # conv2d.py:83-88, linear.py:54-61
        return self.rs.uniform(low=-stdev * np.sqrt(3), high=stdev * np.sqrt(3), size=shape).astype("float32")

    def trunc_normal(self, shape):            # tf.truncated_normal_initializer() for u (sn.py:36)
        x = self.rs_tf.normal(0.0, 1.0, size=shape)
        while True:
            bad = np.abs(x) > 2.0
            if not bad.any():
                return x.astype("float32")
            x[bad] = self.rs_tf.normal(0.0, 1.0, size=int(bad.sum()))

    def conv(self, dst, name, cin, cout, k, he, sn):
        """dst None: a call of Conv2D on an EXISTING variable (a second tower, reuse=True) -- the reference draws the initial
        filter values from numpy before it asks for the variable (conv2d.py:118-140), so the draw happens and is discarded."""
        fan_in, fan_out = cin * k * k, cout * k * k
        sd = np.sqrt((4. if he else 2.) / (fan_in + fan_out))      # conv2d.py:103-106
        w = self.uniform(sd, (k, k, cin, cout))
        if dst is None:
            return
        dst.append((name + "/Filters", (k, k, cin, cout), w))
        if sn:
            self.U[name + "/filters/spectral_norm/u"] = self.trunc_normal((1, cout))
        dst.append((name + "/Biases", (cout,), np.zeros(cout, "float32")))

    def linear(self, dst, name, cin, cout, sn):
        w = self.uniform(np.sqrt(2. / (cin + cout)), (cin, cout))   # linear.py:76-80; drawn on every call (linear.py:54-80)
        if dst is None:
            return
        dst.append((name + "/W", (cin, cout), w))
        if sn:
            self.U[name + "/spectral_norm/u"] = self.trunc_normal((1, cout))
        dst.append((name + "/b", (cout,), np.zeros(cout, "float32")))

    def condbn(self, dst, name, c):           # normalization.py:49-52 (constant initialisers: no draw)
        if dst is None:
            return
        dst.append((name + "/CondBatchNorm/offset", (VOCAB_SIZE, c), np.zeros((VOCAB_SIZE, c), "float32")))
        dst.append((name + "/CondBatchNorm/scale", (VOCAB_SIZE, c), np.ones((VOCAB_SIZE, c), "float32")))

    # ---- one CALL of each model function of gan_resnet.py: dst = the list the variables go to, or None for a reuse=True call
    def generator(self, dst):                 # Generator, gan_resnet.py:356-371
        self.linear(dst, "Generator/G.Input", Z_DIM, 4 * 4 * DIM_G * 8, False)
        for blk, cin in ((1, DIM_G * 8), (2, DIM_G * 2), (3, DIM_G * 2)):
            nm = "Generator/G.Block.%d" % blk
            self.conv(dst, nm + ".Shortcut", cin, DIM_G * 2, 1, False, False)
            self.condbn(dst, nm + ".N1", cin)
            self.conv(dst, nm + ".Conv1", cin, DIM_G * 2, 3, True, False)
            self.condbn(dst, nm + ".N2", DIM_G * 2)
            self.conv(dst, nm + ".Conv2", DIM_G * 2, DIM_G * 2, 3, True, False)
        self.condbn(dst, "Generator/G.OutputNorm", DIM_G * 2)
        self.conv(dst, "Generator/G.Output", DIM_G * 2, IMG_DIM, 3, False, False)

    def discriminator(self, dst):             # Discriminator, gan_resnet.py:374-412
        d = "Discriminator/"
        self.conv(dst, d + "D.Block.1.Shortcut", IMG_DIM, DIM_D, 1, False, True)
        self.conv(dst, d + "D.Block.1.Conv1", IMG_DIM, DIM_D, 3, True, True)
        self.conv(dst, d + "D.Block.1.Conv2", DIM_D, DIM_D, 3, True, True)
        self.conv(dst, d + "D.Block.2.Shortcut", DIM_D, DIM_D, 1, False, True)
        self.conv(dst, d + "D.Block.2.Conv1", DIM_D, DIM_D, 3, True, True)
        self.conv(dst, d + "D.Block.2.Conv2", DIM_D, DIM_D, 3, True, True)
        for blk in (3, 4, 5, 6):
            self.conv(dst, d + "D.Block.%d.Conv1" % blk, DIM_D, DIM_D, 3, True, True)
            self.conv(dst, d + "D.Block.%d.Conv2" % blk, DIM_D, DIM_D, 3, True, True)
        self.linear(dst, d + "D.Output", DIM_D, 1, True)

    def projection(self, dst):                # Discriminator_projection, gan_resnet.py:414-421
        d = "Discriminator/"
        table = self.rs.uniform(-0.08, 0.08, size=(VOCAB_SIZE, EMBEDDING_DIM)).astype("float32")     # embedding.py:29-34
        if dst is not None:
            dst.append((d + "Embedding.Label/embedding_map", (VOCAB_SIZE, EMBEDDING_DIM), table))
        self.linear(dst, d + "D.Embedding_y", EMBEDDING_DIM, DIM_D, True)

    def perm(self, dst, perm_type):           # perm_classifier, gan_resnet.py:458-483
        d = "Discriminator/"
        if perm_type == "linear":
            self.linear(dst, d + "D.d_perm_classifier_h1", OUTPUT_DIM, VOCAB_SIZE, True)
        elif perm_type == "2layer":
            self.linear(dst, d + "D.d_perm_classifier_h1", OUTPUT_DIM, 128, True)
            self.linear(dst, d + "D.d_perm_classifier_h2", 128, VOCAB_SIZE, True)
        else:
            raise ValueError('Unknown perm_type {}'.format(perm_type))


def confusion_logits_initial(confuse_init, confuse_init_diag, rs):
    """gan_resnet.py:499-520."""
    if not confuse_init:
        lim = np.sqrt(6.0 / (2 * VOCAB_SIZE))      # TF default get_variable initialiser: glorot_uniform
        return rs.uniform(-lim, lim, size=(VOCAB_SIZE, VOCAB_SIZE)).astype("float32")
    if confuse_init_diag > 0.99 and VOCAB_SIZE == 10.:
        aa = 7.0
    else:
        aa = np.log(VOCAB_SIZE * confuse_init_diag / (1. - confuse_init_diag))
    aa = min(7.0, aa)
    m = (0 - aa / VOCAB_SIZE) * np.ones([VOCAB_SIZE, VOCAB_SIZE], dtype=np.float32)
    np.fill_diagonal(m, (aa - (aa / VOCAB_SIZE)))
    return m


N_TOWERS = 2      # len(DEVICES): two towers even on one device (gan_resnet.py:186-188)


def create_variables(seed=0, algorithm="rcgan", perm_classifier=False, perm_type="linear",
                     confuse_init=False, confuse_init_diag=0.2):
    """-> (G specs, D specs, C specs, U dict); specs are (name, shape, initial value).
    The numpy-initialised values equal the reference's under ``np.random.seed(seed)`` bit for bit (pinned by
    tests/golden/ref_cifar_*.npz, which scripts/make_golden_reference.py produces by running the reference's own main()).  That needs
    the reference's graph-construction ORDER including the calls that create nothing: every Conv2D / Linear / embed_y call draws
    its initial values before asking for the variable, so the second tower's Generator (reuse=True, gan_resnet.py:541-546), the
    ten extra projections of 'unbiased' (:615-622) and the second Discriminator + projection of 'rcgan-u' (:654-657) consume the
    stream in front of the variables created after them."""
    if algorithm not in ALGORITHMS:
        raise ValueError("Unknown algorithm %s" % algorithm)
    it = _Init(seed)
    if algorithm == "rcgan-u":
        it.Cm.append(("confusion_logits", (VOCAB_SIZE, VOCAB_SIZE),
                      confusion_logits_initial(confuse_init, confuse_init_diag, it.rs_tf)))
    it.generator(it.G)                        # tower 0 creates (gan_resnet.py:541-546) ...
    for _ in range(N_TOWERS - 1):
        it.generator(None)                    # ... the other towers reuse
    it.discriminator(it.D)                    # tower 0 of the critic cost (:584-586)
    it.projection(it.D)
    if algorithm == "unbiased":
        for _ in range(VOCAB_SIZE):
            it.projection(None)               # :615-622
    elif algorithm == "rcgan-u":
        it.discriminator(None)                # :654-657
        it.projection(None)
    if perm_classifier:
        it.perm(it.D, perm_type)              # :686-695, created at the end of tower 0
    return it.G, it.D, it.Cm, it.U


def C_ALPHA(alpha):
    """gan_resnet.py:106."""
    return ((1 - alpha) / 9.0) * np.ones((10, 10)) + (alpha - (1 - alpha) / 9.0) * np.eye(10)


FEED_COPY_KERNEL = os.environ.get("RCGAN_FEED_COPY_KERNEL", "1") == "1"


def lr_decay(iteration):
    """gan_resnet.py:700-705."""
    return max(0., 1. - iteration / 100000.) if iteration < 50000 else 0.5


# ------------------------------------------------------------------------------------------------------
# model blocks (gan_resnet.py:199-421).  Fusions relative to the reference graph, all algebraic identities:
#   * Normalize + nonlinearity -> one fused condBN+ReLU kernel;   * nonlinearity before a conv in D ->
#     folded into the conv operand load;   * UpsampleConv -> upsample folded into the conv's input indexing;
#   * shortcut + ConvMeanPool(...) with a pooled 1x1 shortcut -> both convs accumulate into one buffer that
#     is pooled once (mean-pool and 1x1 conv commute; pooling is linear).
# ------------------------------------------------------------------------------------------------------
def _ctx():
    return Graph.current.ctx


SHORTCUT_BEFORE_UPSAMPLE = os.environ.get("RCGAN_SHORTCUT_LOW", "1") == "1"


def UpsampleConv(inputs, output_dim, filter_size=3, name=None, spectral_normed=False, update_collection=None,
                 he_init=True, biases=True, _in_relu=False, _accumulate_into=None, _bn_next=False):
    return Conv2D(inputs, inputs.shape[-1], output_dim, filter_size, 1, name, spectral_normed=spectral_normed,
                  update_collection=update_collection, he_init=he_init, biases=biases,
                  _in_upsample=True, _in_relu=_in_relu, _accumulate_into=_accumulate_into, _bn_next=_bn_next)


def G_ResidualBlock(inputs, input_dim, output_dim, filter_size, name, labels, segments=1):
    """ResidualBlock(resample='up') with conditional batch norm (gan_resnet.py:275-328)."""
    # the 1x1 shortcut commutes with the nearest upsample in front of it (gan_resnet.py:258-272): evaluated on the block's
    # low-resolution input (a quarter of the multiply-adds and bytes) and added upsampled by Conv2's epilogue
    low = SHORTCUT_BEFORE_UPSAMPLE
    if low:
        shortcut = Conv2D(inputs, input_dim, output_dim, 1, 1, name + '.Shortcut', he_init=False)
    else:
        shortcut = UpsampleConv(inputs, output_dim, 1, name + '.Shortcut', he_init=False)
    with variable_scope(name + '.N1'):
        # (_defer_apply: in forward-only passes the affine + ReLU go into the consuming convolution's staged input where its kernel can
        # -- ops.BnPending: the halo-patch kernels and G.Output's; elsewhere, and under the tape, the norm is applied as before)
        out = cond_batchnorm(name + '.N1', [0, 1, 2], inputs, labels=labels, n_labels=10, _act=L.ACT_RELU, _segments=segments, _defer_apply=True)
    # (_bn_next: a batch norm follows -- on the big grids its statistics come out of this convolution's epilogue, ops.conv2d)
    out = UpsampleConv(out, output_dim, filter_size, name + '.Conv1', _bn_next=True)
    with variable_scope(name + '.N2'):
        out = cond_batchnorm(name + '.N2', [0, 1, 2], out, labels=labels, n_labels=10, _act=L.ACT_RELU, _segments=segments, _defer_apply=True)
    if low:
        return Conv2D(out, output_dim, output_dim, filter_size, 1, name + '.Conv2', _residual=shortcut, _residual_up=True, _bn_next=True)
    return Conv2D(out, output_dim, output_dim, filter_size, 1, name + '.Conv2', _accumulate_into=shortcut)


def Generator(n_samples, labels, noise, out=None, segments=1):
    """gan_resnet.py:356-371.  noise: [n,128] device tensor; returns [n, 3072] (NHWC flattened) in (-1,1).
    segments > 1 (forward only): the n samples are that many independent Generator() batches back to back, each with
    its own batch-norm statistics -- the convolutions run once on all of them."""
    ctx = _ctx()
    with variable_scope("Generator"):
        output = Linear(noise, 128, 4 * 4 * DIM_G * 8, 'G.Input')
        output = O.reshape(ctx, output, (-1, 4, 4, DIM_G * 8))
        output = G_ResidualBlock(output, DIM_G * 8, DIM_G * 2, 3, 'G.Block.1', labels, segments)
        if Graph.current.early_g is not None and output.req:
            ctx.record(Graph.current.early_g)       # backward: G.Block.2 .. G.Output are done here -> their bucket leaves early
        output = G_ResidualBlock(output, DIM_G * 2, DIM_G * 2, 3, 'G.Block.2', labels, segments)
        output = G_ResidualBlock(output, DIM_G * 2, DIM_G * 2, 3, 'G.Block.3', labels, segments)
        with variable_scope('G.OutputNorm'):
            # (forward-only passes -- the critic steps' generator forwards, sampling: the affine + ReLU are applied inside G.Output's
            # launch, the normalised tensor is never written: ops.BnPending)
            output = cond_batchnorm('G.OutputNorm', [0, 1, 2], output, labels=labels, n_labels=10, _act=L.ACT_RELU, _segments=segments,
                                    _defer_apply=True)
        output = Conv2D(output, DIM_G * 2, IMG_DIM, 3, 1, 'G.Output', he_init=False)
        output = O.act(ctx, output, L.ACT_TANH, out=out.reshape(output.shape) if out is not None else None)
        return O.reshape(ctx, output, (-1, OUTPUT_DIM))


def fused_pool_d(ctx, n):
    """The down blocks' ConvMeanPool layers run with the pool folded into the convolution: 16-bit activations and a batch that
    gives whole 64-pixel tiles of pooled pixels on both layers (n * 64 % 64 == 0 always; n * 256 for D.Block.1)."""
    if ctx.act_dtype == L.F32 or os.environ.get("RCGAN_FUSED_POOL", "1") != "1":
        return False
    d1 = L.ConvDesc(n, IMG_SIZE, IMG_SIZE, DIM_D, DIM_D, 3, 3, 1, ctx.act_dtype, L.CONV_OUT_MEANPOOL2)
    d2 = L.ConvDesc(n, IMG_SIZE // 2, IMG_SIZE // 2, DIM_D, DIM_D, 3, 3, 1, ctx.act_dtype, L.CONV_OUT_MEANPOOL2)
    return bool(ctx.lib.rcgan_conv_fused_pool_ok(C.byref(d1))) and bool(ctx.lib.rcgan_conv_fused_pool_ok(C.byref(d2)))


def Discriminator(inputs, labels, update_collection=None, _head=True):
    """gan_resnet.py:374-412 (+ OptimizedResBlockDisc1 :331-353, ResidualBlock :275-328).  No norm in D
    (NORMALIZATION_D=False), so ``labels`` is unused exactly as in the reference.  _head=False: return the pooled features
    only (D.Output is then applied inside Discriminator_head's fused launch)."""
    ctx = _ctx()
    kw = dict(spectral_normed=True, update_collection=update_collection)
    with variable_scope("Discriminator"):
        x = O.reshape(ctx, inputs, (-1, IMG_SIZE, IMG_SIZE, IMG_DIM))
        if fused_pool_d(ctx, x.shape[0]):
            # ConvMeanPool with the pool folded into the convolution (one 4x4 stride-2 convolution, ops.conv2d_meanpool); the
            # shortcuts as the reference writes them, MeanPoolConv: 1x1 convolution of the pooled input (gan_resnet.py:249-257, 346)
            # (the critic step's input rider has pooled the images already: x.pooled)
            xp = getattr(Graph.current, "image_pool", None)
            if xp is not None and (x.req or xp.shape != (x.shape[0], x.shape[1] // 2, x.shape[2] // 2, x.shape[3])):
                xp = None
            t = Conv2D(xp if xp is not None else O.meanpool2(ctx, x), IMG_DIM, DIM_D, 1, 1, 'D.Block.1.Shortcut', he_init=False, **kw)
            h = Conv2D(x, IMG_DIM, DIM_D, 3, 1, 'D.Block.1.Conv1', **kw)
            x = Conv2D(h, DIM_D, DIM_D, 3, 1, 'D.Block.1.Conv2', _in_relu=True, _accumulate_into=t, _out_meanpool=True, **kw)
            t = Conv2D(O.meanpool2(ctx, x), DIM_D, DIM_D, 1, 1, 'D.Block.2.Shortcut', he_init=False, **kw)
            h = Conv2D(x, DIM_D, DIM_D, 3, 1, 'D.Block.2.Conv1', _in_relu=True, **kw)
            x = Conv2D(h, DIM_D, DIM_D, 3, 1, 'D.Block.2.Conv2', _in_relu=True, _accumulate_into=t, _out_meanpool=True, **kw)
        else:
            # D.Block.1: shortcut = conv1x1(meanpool(x)) == meanpool(conv1x1(x)); pooled together with Conv2
            t = Conv2D(x, IMG_DIM, DIM_D, 1, 1, 'D.Block.1.Shortcut', he_init=False, **kw)
            h = Conv2D(x, IMG_DIM, DIM_D, 3, 1, 'D.Block.1.Conv1', **kw)
            t = Conv2D(h, DIM_D, DIM_D, 3, 1, 'D.Block.1.Conv2', _in_relu=True, _accumulate_into=t, **kw)
            x = O.meanpool2(ctx, t)
            # D.Block.2 (down)
            t = Conv2D(x, DIM_D, DIM_D, 1, 1, 'D.Block.2.Shortcut', he_init=False, **kw)
            h = Conv2D(x, DIM_D, DIM_D, 3, 1, 'D.Block.2.Conv1', _in_relu=True, **kw)
            t = Conv2D(h, DIM_D, DIM_D, 3, 1, 'D.Block.2.Conv2', _in_relu=True, _accumulate_into=t, **kw)
            x = O.meanpool2(ctx, t)
        if Graph.current.early_d is not None and x.req:
            ctx.record(Graph.current.early_d)       # backward: D.Block.3 .. the head are done here -> their bucket leaves early
        if FUSED_TRUNK and O.d_trunk_ok(ctx, x):
            # D.Block.3 .. D.Block.6 (identity shortcuts, 8 x 8 pixels): one launch for the eight convolutions (ops.d_trunk)
            g, blocks = Graph.current, []
            for blk in (3, 4, 5, 6):
                ws = []
                for cv in ('Conv1', 'Conv2'):
                    with variable_scope('D.Block.%d.%s' % (blk, cv)):
                        fname = scoped('Filters')
                        with variable_scope('filters'):
                            ws.append(spectral_normed_weight(fname, update_collection=update_collection))
                        ws.append(g.param(scoped('Biases')))
                blocks.append(tuple(ws))
            x = O.d_trunk(ctx, x, blocks, pool=(L.ACT_RELU if POOL_IN_TRUNK else None), frag=getattr(g, "trunk_frag", None))
        else:
            for blk in (3, 4, 5, 6):          # identity shortcut (in==out, no resample)
                h = Conv2D(x, DIM_D, DIM_D, 3, 1, 'D.Block.%d.Conv1' % blk, _in_relu=True, **kw)
                x = Conv2D(h, DIM_D, DIM_D, 3, 1, 'D.Block.%d.Conv2' % blk, _in_relu=True, _residual=x, **kw)   # shortcut + output
        if not _head:
            return O.act_meanhw_later(ctx, x, L.ACT_RELU)               # pooled inside Discriminator_head's launch
        output = O.act_meanhw(ctx, x, L.ACT_RELU)                       # relu + reduce_mean over (1,2)
        output_wgan = Linear(output, DIM_D, 1, 'D.Output', **kw)
        return output, O.reshape(ctx, output_wgan, (-1,))


def Discriminator_projection(labels, update_collection=None):
    """gan_resnet.py:414-421."""
    with variable_scope("Discriminator"):
        e = embed_y(labels, VOCAB_SIZE, EMBEDDING_DIM)
        return Linear(e, EMBEDDING_DIM, DIM_D, 'D.Embedding_y', spectral_normed=True,
                      update_collection=update_collection, biases=True)


def _head_weights(update_collection=None):
    """(w_out, b_out, table, w_e, b_e) of the projection head under the reference's variable names."""
    g = Graph.current
    with variable_scope("Discriminator"):
        with variable_scope('D.Output'):
            w_out = spectral_normed_weight(scoped('W'), update_collection=update_collection)
            b_out = g.param(scoped('b'))
        with variable_scope("Embedding.Label"):
            table = g.param(scoped('embedding_map'))
        with variable_scope('D.Embedding_y'):
            w_e = spectral_normed_weight(scoped('W'), update_collection=None)
            b_e = g.param(scoped('b'))
    return w_out, b_out, table, w_e, b_e


def Discriminator_head(features, parts, weight, loss_acc, update_collection=None, logits=None):
    """The tail of Discriminator (D.Output, gan_resnet.py:408-411), Discriminator_projection (:414-421), the projection
    logit (:588; every label's logit :654-660) and the loss terms built on it (:604-606, :647, :673-684, :751-773) as ONE
    launch with all their gradients (ops.proj_head).  update_collection: that of D.Output's spectral norm (the projection's
    is None at every call site of the reference).  parts: see ops.proj_head."""
    g = Graph.current
    w_out, b_out, table, w_e, b_e = _head_weights(update_collection)
    # (the label embeddings E = table @ W_e / sigma + b_e of this step, if _prepare_all let them ride in its launch)
    O.proj_head(g.ctx, features, w_out, b_out, table, w_e, b_e, parts, weight, loss_acc, logits=logits, E_pre=getattr(g, "head_E", None))


def perm_classifier(x, perm_type='linear'):
    """gan_resnet.py:458-483."""
    ctx = _ctx()
    with variable_scope("Discriminator"):
        x = O.cast(ctx, O.reshape(ctx, x, (-1, OUTPUT_DIM)), L.F32)
        if perm_type == 'linear':
            return Linear(x, OUTPUT_DIM, VOCAB_SIZE, 'D.d_perm_classifier_h1', spectral_normed=True, biases=True)
        elif perm_type == '2layer':
            h = Linear(x, OUTPUT_DIM, 128, 'D.d_perm_classifier_h1', spectral_normed=True, biases=True)
            return Linear(h, 128, VOCAB_SIZE, 'D.d_perm_classifier_h2', spectral_normed=True, biases=True)
        raise ValueError('Unknown perm_type {}'.format(perm_type))


# ------------------------------------------------------------------------------------------------------
# trainer
# ------------------------------------------------------------------------------------------------------
class CifarRCGAN:
    """One rank of the data-parallel RCGAN engine.  ``batch_size`` is the per-rank critic batch
    (the reference's BATCH_SIZE / len(DEVICES) tower batch, gan_resnet.py:190-192,544)."""

    def __init__(self, algorithm="rcgan", alpha=0.6, batch_size=64, lr=2e-4, dtype="bf16", seed=0,
                 perm_classifier=False, perm_multiplier=1.0, perm_type="linear",
                 confuse_init=False, confuse_init_diag=0.2, confuse_multiplier=1.0, confuse_lr_decay=False,
                 device=0, use_graphs=True, device_rng=True, arena_bytes=None, world_size=1, rank=0,
                 variables=None, loss_scale=None, dynamic_loss_scale=None, loss_scale_growth_interval=2000, comm=None,
                 grad_bucket_dtype=None, stub_model=None):
        if algorithm not in ALGORITHMS:
            raise ValueError("Unknown algorithm %s" % algorithm)
        self.alg, self.alpha, self.B, self.lr = algorithm, alpha, int(batch_size), lr
        self.perm, self.perm_mult, self.perm_type = perm_classifier, perm_multiplier, perm_type
        self.confuse_multiplier, self.confuse_lr_decay = confuse_multiplier, confuse_lr_decay
        self.world, self.rank = world_size, rank
        self.use_graphs, self.device_rng = use_graphs, device_rng
        # the projection head + loss terms as one launch (Discriminator_head); RCGAN_FUSED_HEAD=0 keeps the op-by-op form
        self.fused_head = os.environ.get("RCGAN_FUSED_HEAD", "1") == "1" and 2 * int(batch_size) <= O.HEAD_MAX_N
        self.ride_inputs = os.environ.get("RCGAN_RIDE_INPUTS", "1") == "1"
        if arena_bytes is None:
            # ~20 MB/sample with 16-bit activations (the five-step generator pass at 5B samples is the high-water mark), twice
            # that in fp32, + slack
            arena_bytes = int(2.5e6 * 4 * self.B * (4 if dtype == "f32" else 2)) + (1 << 30)
        self.ctx = Context(device, dtype, arena_bytes=arena_bytes)
        ctx = self.ctx
        # Static loss scaling for fp16 activations (5 exponent bits: activation gradients of ~1e-5 and below would go
        # subnormal): every loss term -- hence every activation and filter gradient -- is multiplied by a power of two,
        # the fp32 filter gradients are divided by it inside the Adam kernel (grad_scale).  1 for bf16 / fp32.
        self.loss_scale = float(loss_scale) if loss_scale is not None else (1024.0 if ctx.act_dtype == L.F16 else 1.0)
        # fp16: DYNAMIC loss scaling by default -- loss_scale is the initial value; after the all-reduce of an optimiser step the
        # gradient slab is checked for inf / nan on the device, an overflowed step is skipped and halves the scale, 2000 applied
        # steps in a row double it (rcgan_grad_finite_check / rcgan_adam_tf_dyn / rcgan_loss_scale_update; no host round trip).
        # The loss kernels read the scale from device memory (rcgan_set_grad_scale) and apply it to the gradients only: the loss
        # values are never scaled.  dynamic_loss_scale=False keeps the scale fixed (no overflow check).
        self.dynamic_ls = (ctx.act_dtype == L.F16) if dynamic_loss_scale is None else bool(dynamic_loss_scale)
        self.ls_growth_interval = float(loss_scale_growth_interval)
        if self.dynamic_ls:
            self.ls_state = torch.tensor([self.loss_scale, 0.0, 0.0, 0.0], dtype=torch.float32, device=ctx.device)
            ctx.check(ctx.lib.rcgan_set_grad_scale(ctx.h, 1.0, C.c_void_p(self.ls_state.data_ptr())))
        else:
            self.ls_state = None
            ctx.check(ctx.lib.rcgan_set_grad_scale(ctx.h, self.loss_scale, None))
        if variables is None:
            variables = create_variables(seed, algorithm, perm_classifier, perm_type, confuse_init, confuse_init_diag)
        gs, ds, cs, U = variables
        self.PG, self.PD = ParamGroup(ctx, gs), ParamGroup(ctx, ds, sn_scratch=True)
        self.PC = ParamGroup(ctx, cs) if cs else None
        self.groups = [self.PG, self.PD] + ([self.PC] if self.PC else [])
        self.state = {}
        for k, v in U.items():
            t = ctx.persistent((v.size,), L.F32)
            ctx.view(t).copy_(torch.from_numpy(np.ascontiguousarray(v.reshape(-1))))
            self.state[k] = t
        self.graph = Graph(ctx, self.groups, self.state)
        # Data parallel (world_size > 1; gan_resnet.py:529-546,697,786): the gradient slabs are all-reduced INSIDE the C ABI
        # (rcgan_allreduce_sum*, RCCL over xGMI) and inside the step's captured graph; the optimiser launch is part of the same graph.
        #   comm: None -> RCCL (one process per GPU, dp.init_comm);  "stub" -> the single-process test double (every rank holds what
        #   this rank holds), which lets ONE GPU run and verify the whole world > 1 schedule.
        #   "rccl-self" (world_size 1) -> a ONE-rank RCCL communicator under the same schedule: real ncclAllReduce calls, captured and
        #   replayed, on a single GPU.
        self.comm_kind = None
        self.dp_active = self.world > 1 or comm == "rccl-self"
        if self.dp_active:
            from . import dp
            self.comm_kind = comm or "rccl"
            if self.comm_kind not in ("rccl", "stub", "rccl-self") or (self.comm_kind == "rccl-self" and self.world != 1):
                raise ValueError("Unknown comm %s for world_size %d" % (comm, self.world))
            dp.init_comm(ctx, self.world, self.rank, stub=(self.comm_kind == "stub"))
            if stub_model is not None:
                # (bus GB/s, latency us): the test double then occupies its stream like an all-reduce under that link model
                ctx.check(ctx.lib.rcgan_comm_stub_model(ctx.h, float(stub_model[0]), float(stub_model[1])))
        # gradient buckets travel as fp32 (default: the reference sums fp32 tower gradients) or as bf16 (half the xGMI bytes; the
        # sum over the ranks then carries 8 mantissa bits -- an opt-in trade, RCGAN_DP_BUCKET_DTYPE=bf16)
        self.grad_bucket_dtype = grad_bucket_dtype or os.environ.get("RCGAN_DP_BUCKET_DTYPE", "f32")
        if self.grad_bucket_dtype not in ("f32", "bf16"):
            raise ValueError("Unknown gradient bucket dtype %s" % self.grad_bucket_dtype)
        self._bucket16 = None
        # One whole-slab bucket per optimiser group and step, on the step's own stream.  An overlapped schedule (the last layers' bucket
        # leaving on the communication stream in the middle of the backward pass) was built in round 3 and deleted in round 4: against
        # the test double it cost 0.6-0.73 ms per iteration of extra launches (second filter-gradient group, second spectral-norm
        # backward, fork / join) with nothing to hide, and under the link model of `bench.py --dp-stub 8 --dp-stub-gbps 200
        # --dp-stub-lat-us 40` it still lost: 6.94 ms against 6.70 ms (fp32 buckets) and 6.42 ms (bf16 buckets) -- DESIGN 5.
        # (round 6) ... and is back behind a switch, so that the first 8-GPU box can A/B the two schedules instead of trusting the link
        # model: RCGAN_DP_OVERLAP=1 = two buckets per step, the layers whose backward finishes first (D.Block.3 .. head | G.Block.2 ..
        # G.Output) leave on the communication stream while the rest of the backward pass runs, the remainder follows on the step's stream
        # (reverse-layer-order buckets overlapped with backward).  fp32 buckets only.
        self.dp_overlap = self.dp_active and self.grad_bucket_dtype == "f32" and os.environ.get("RCGAN_DP_OVERLAP", "0") == "1"
        # first parameter of the early bucket of each group (everything from there to the end of the slab)
        self._early_lo = {id(self.PD): self.PD.offsets["Discriminator/D.Block.3.Conv1/Filters"],
                          id(self.PG): self.PG.offsets["Generator/G.Block.2.Shortcut/Filters"]}
        self.dp_adam_in_graph = self.dp_active and not self.dynamic_ls and os.environ.get("RCGAN_DP_GRAPH_ADAM", "1") == "1"
        # (round 5, measured and left off) single rank: the optimiser launch at the end of the step's captured graph too ({lr, t} from
        # device memory, the rcgan_adam_tf launch the data-parallel steps capture).  As an eager launch BEHIND the graph it starts ~8 us
        # after the graph's last kernel (the only gap of a critic step in the launch sequence), six times per iteration -- but the
        # one-thread launch that writes {lr, t} in front of every graph costs more than the gap: same-box 5.50 / 5.52 / 5.52 ms eager
        # against 5.55 / 5.56 / 5.55 captured.  RCGAN_GRAPH_ADAM=1 turns it on.
        self.graph_adam = self.dp_adam_in_graph or (not self.dp_active and not self.dynamic_ls and
                                                    os.environ.get("RCGAN_GRAPH_ADAM", "0") == "1")
        # (round 6) single rank, static loss scale: the critic step's optimiser inside the second launch of the spectral-norm backward
        # (rcgan_sn_bwd_adam: TF-Adam on the rows of dW a workgroup has just formed, rider workgroups for the biases / embeddings
        # between them) -- the step's last launch updates the discriminator; no optimiser launch, no {lr, t} launch (the step count
        # lives on the device and moves on inside the call; the host rewrites {lr, t} only when lr changes: once per iteration).
        # RCGAN_SN_ADAM=0 restores the separate launch.
        self.fused_tail = (not self.dp_active and not self.dynamic_ls and not self.graph_adam
                           and os.environ.get("RCGAN_SN_ADAM", "1") == "1")
        self._tail_fused = {}
        B = self.B
        f32, i32, act = L.F32, "i32", ctx.act_dtype
        P = ctx.persistent
        # static step inputs (graph replays read these addresses)
        # The per-step batch inputs live in two contiguous 4-byte-word slabs ("feeds"), so a loader hands a whole D-step /
        # G-step batch over with ONE copy (set_feed); the named views below alias the slabs (set_inputs still works).
        self.feed_layout = {
            "d": [("images", (B, OUTPUT_DIM), i32), ("labels", (B,), i32), ("labels_random", (B,), i32), ("labels_biased", (B,), i32),
                  ("inv_weights", (B, VOCAB_SIZE), f32), ("labels_all", (2 * B,), i32)],
            "g": [("labels_random_G", (2 * B,), i32), ("labels_biased_G", (2 * B,), i32)],
            # generator labels of the N_CRITIC critic steps of one iteration (prepare_critic_fakes)
            "gf": [("labels_random_all", (N_CRITIC * B,), i32)]}
        self.feed, self.inp = {}, {}
        self._feed_ring = {}
        for key, fields in self.feed_layout.items():
            words = sum(int(np.prod(shp)) for _, shp, _ in fields)
            slab = torch.zeros(words, dtype=torch.int32, device=ctx.device)
            self.feed[key] = slab
            off = 0
            for name, shp, dt in fields:
                self.inp[name] = DT(slab.data_ptr() + 4 * off, shp, dt, slab, name)
                off += int(np.prod(shp))
        # (round 6) the N_CRITIC critic steps of an iteration as ONE captured graph (critic_steps): a slab of N_CRITIC "d" batches, and per
        # slot the views the step body reads its inputs through while that slot's step is recorded
        words_d = self.feed["d"].numel()
        self.feed["d5"] = torch.zeros(N_CRITIC * words_d, dtype=torch.int32, device=ctx.device)
        self._d_slot_views = []
        for k in range(N_CRITIC):
            off, views = k * words_d, {}
            for name, shp, dt in self.feed_layout["d"]:
                views[name] = DT(self.feed["d5"].data_ptr() + 4 * off, shp, dt, self.feed["d5"], name)
                off += int(np.prod(shp))
            self._d_slot_views.append(views)
        self.critic_graph = os.environ.get("RCGAN_CRITIC_GRAPH", "1") == "1"
        self.inp.update(noise=P((B, OUTPUT_DIM), f32), z=P((B, Z_DIM), act), z_G=P((2 * B, Z_DIM), act),
                        z_all=P((N_CRITIC * B, Z_DIM), act),
                        arange=P((VOCAB_SIZE,), i32), C_const=P((VOCAB_SIZE, VOCAB_SIZE), f32))
        ctx.view(self.inp["arange"]).copy_(torch.arange(VOCAB_SIZE, dtype=torch.int32))
        ctx.view(self.inp["C_const"]).copy_(torch.from_numpy(C_ALPHA(alpha).astype(np.float32)))
        # the critic steps' generator forwards evaluated as one batch (prepare_critic_fakes): fakes of all N_CRITIC steps,
        # and the discriminator input [real ; fake] of a step at a fixed address the fake rows are copied into
        self.fakes_all = P((N_CRITIC * B, OUTPUT_DIM), act)
        self.x_all = P((2 * B, OUTPUT_DIM), act)
        self._fakes_left = 0
        # loss accumulators: scalars behind the gradient slabs, cleared by the step's zero_grad()
        self.loss_d = self.PD.scalar(0)
        self.loss_g = self.PG.scalar(0)
        self.rng_state = torch.zeros(2, dtype=torch.int64, device=ctx.device)
        # (round 6) The critic steps' generator forwards on a SECOND stream.  The generator does not change during the N_CRITIC critic
        # updates of an iteration (gan_resnet.py:928-947), so step k + 1's Generator() call can run while critic step k does: a critic
        # step is a chain of ~26 small-grid launches that leaves half the chip idle most of the time, the generator forward is the
        # big-grid work that fills it.  prepare_critic_fakes() then enqueues N_CRITIC per-step passes (B samples each, the reference's
        # own call shape) on the second context's stream and every d_step() waits for its slice's event; a second context = its own
        # stream, arena, workspace, arrival counters and random-stream position, so nothing is shared but the (read-only) generator
        # parameters, their prepared filters and the step inputs.
        # MEASURED AND LEFT OFF (RCGAN_OVERLAP_GF=1 turns it on; tests/test_gpu_cifar_step.py runs both forms): same box, B = 64,
        # 5.26 ms per iteration with the one batched pass on the step stream against 5.47 ms overlapped.  The premise does not hold on
        # this chip (scripts/exp_overlap_streams.py, two engines with NO dependency between them): 5 critic steps 2.08 ms + the
        # batched generator pass 1.11 ms = 3.23 ms one after the other, 2.91 ms side by side -- only 0.3 ms of the 1.1 hides -- and
        # the per-step passes the dependency needs cost 1.72 ms instead of 1.11 (half-empty grids at n = 64): 3.11 ms pipelined.
        self.overlap_gf = os.environ.get("RCGAN_OVERLAP_GF", "0") == "1"
        self.ctx2 = None
        if self.overlap_gf:
            self.ctx2 = Context(device, dtype, arena_bytes=int(2.5e6 * self.B * (4 if dtype == "f32" else 2)) + (256 << 20), ws_bytes=256 << 20)
            ctx.also_close = [self.ctx2]
            ctx.check(self.ctx2.lib.rcgan_set_grad_scale(self.ctx2.h, 1.0, None))
            self.graph2 = Graph(self.ctx2, [ParamGroup.alias(self.ctx2, self.PG)], {})
            self.graph2.persist = self.graph.persist        # the generator's prepared filters (refresh_persistent on the step stream)
            self.rng_state_gf = torch.zeros(2, dtype=torch.int64, device=ctx.device)
            self._gf_done = [torch.cuda.Event() for _ in range(N_CRITIC)]
            self._gf_pending = [False] * N_CRITIC
            torch.cuda.set_device(ctx.device)
        self.slice_ctr = torch.zeros(4, dtype=torch.int32, device=ctx.device)     # which slice of fakes_all the next critic step takes
        self._slice_mirror = 0
        self.seed = seed
        self._graphs = {}
        # inspection hook: a persistent fp32 [2B, 10] buffer the fused projection head also writes its logits to (logit of the
        # sample's label, or of every label where the loss weights all of them); None in production.  Set before the first step.
        self.head_logits = None
        self.iteration = 0
        self._pg_prepared_version = -1
        torch.cuda.synchronize()

    # ---------------------------------------------------------------------------------- helpers
    def set_inputs(self, **arrays):
        """Copy host arrays into the static device inputs (int arrays -> int32, floats -> buffer dtype)."""
        ctx = self.ctx
        with torch.cuda.stream(ctx.stream):
            for k, a in arrays.items():
                dst = ctx.view(self.inp[k])
                src = torch.from_numpy(np.ascontiguousarray(np.asarray(a)))
                # pinned staging + asynchronous copy: a pageable (synchronous) copy would wait for every launch already
                # queued on the device -- one full step per array -- and serialise the host loop with the GPU
                dst.copy_(src.reshape(dst.shape).to(dst.dtype).pin_memory(), non_blocking=True)

    def pack_feed(self, key, **arrays):
        """Host-side packing of one step's batch into the feed layout (int32 words; float fields bit-cast): what a data
        loader prepares (pinned) so that set_feed is a single transfer."""
        parts = []
        for name, shp, dt in self.feed_layout[key]:
            a = np.ascontiguousarray(np.asarray(arrays[name]).reshape(shp))
            parts.append(a.astype(np.float32).view(np.int32).reshape(-1) if dt == L.F32 else a.astype(np.int32).reshape(-1))
        return np.concatenate(parts)

    def feed_host(self, key, **arrays):
        """pack_feed + set_feed for a host-side loader: the batch is packed straight into a pinned staging slot (a ring of
        8 per feed, guarded by events) and handed over with ONE asynchronous copy -- no per-array conversions, pinned
        allocations or synchronous copies in the training loop."""
        steps = arrays.pop("_steps", None)       # key "d5": a list of N_CRITIC "d" batches, packed slot by slot
        ring = self._feed_ring.setdefault(key, {"slots": [], "events": [], "next": 0})
        words = self.feed[key].numel()
        if not ring["slots"]:
            ring["slots"] = [torch.empty(words, dtype=torch.int32).pin_memory() for _ in range(8)]
            ring["events"] = [None] * 8
        i = ring["next"]
        ring["next"] = (i + 1) % 8
        if ring["events"][i] is not None:
            ring["events"][i].synchronize()              # the copy that last read this slot has finished
        dst = ring["slots"][i].numpy()
        off = 0
        for arrs in (steps if steps is not None else [arrays]):
            for name, shp, dt in self.feed_layout["d" if steps is not None else key]:
                n = int(np.prod(shp))
                a = np.asarray(arrs[name]).reshape(-1)
                if dt == L.F32:
                    dst[off:off + n].view(np.float32)[:] = a
                else:
                    dst[off:off + n] = a
                off += n
        assert off == words, (off, words)
        with torch.cuda.stream(self.ctx.stream):
            self.feed[key].copy_(ring["slots"][i], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.ctx.stream)
        ring["events"][i] = ev

    def set_feed(self, key, blob):
        """One copy of a packed batch (numpy int32 array or device / pinned int32 tensor) into the D-step ("d") or G-step
        ("g") inputs."""
        src = blob if isinstance(blob, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(blob, dtype=np.int32))
        if src.is_cuda and src.dtype == torch.int32 and src.is_contiguous() and FEED_COPY_KERNEL:
            # (round 6) a device-resident batch: an ordinary kernel on the step's stream instead of the runtime's device-to-device copy
            ctx = self.ctx
            assert src.numel() == self.feed[key].numel(), (src.numel(), self.feed[key].numel())
            ctx.check(ctx.lib.rcgan_copy_words(ctx.h, src.numel(), C.c_void_p(src.data_ptr()), C.c_void_p(self.feed[key].data_ptr())))
            return
        with torch.cuda.stream(self.ctx.stream):
            self.feed[key].copy_(src.reshape(-1), non_blocking=True)

    def _rng(self, t, kind, lo, hi):
        ctx = self.ctx
        ctx.check(ctx.lib.rcgan_rng_fill(ctx.h, t.size, t.dtype, kind, lo, hi, self.seed * 1000003 + self.rank,
                                         C.c_void_p(self.rng_state.data_ptr()), C.c_void_p(t.ptr)))

    def _sn_entries(self, conv_update, proj_update):
        ents = []
        for name in self.PD.names:
            if name.endswith("/Filters"):
                ents.append((name, name[:-len("Filters")] + "filters/spectral_norm/u", conv_update))
            elif name.endswith("/W"):
                base = name[:-2]
                upd = conv_update if base.endswith("D.Output") else proj_update
                ents.append((name, base + "/spectral_norm/u", upd))
        return ents

    def _filter_names(self, grp):
        names = []
        pool = fused_pool_d(self.ctx, GEN_BS_MULTIPLE * self.B)      # D always sees 2B images (real + fake, or 2B fakes)
        for n in grp.names:
            if n.endswith("/Filters"):
                shp = grp.shapes[n]
                # the 3-channel image-end convs (D.Block.1.*, G.Output) run at the image resolution
                # the up blocks' first convolutions read their input through the nearest 2x upsample (UpsampleConv, gan_resnet.py:259-272):
                # their prepared buffers also carry the summed filters of the sub-pixel form
                up = L.CONV_IN_UPSAMPLE2X if (n.startswith("Generator/G.Block.") and n.endswith(".Conv1/Filters")) else 0
                # ... and the down blocks' second convolutions those of the folded mean pool (ops.conv2d_meanpool)
                if pool and n in ("Discriminator/D.Block.1.Conv2/Filters", "Discriminator/D.Block.2.Conv2/Filters"):
                    up = L.CONV_OUT_MEANPOOL2
                names.append((n, shp[0], 1, IMG_SIZE if min(shp[2], shp[3]) <= 3 else 8, up))
            elif n == "Generator/G.Input/W" and O.LINEAR_MFMA:
                names.append((n, 1, 1, 1, 0))          # the dense input layer runs as a 1x1 convolution (ops.linear): its 16-bit filter
        return names

    def _prepare_all(self, which, head_update=False, inputs=None):
        """One launch prepares every spectrally normalised conv filter the step will use (W/sigma changes with
        every power iteration).  The generator's filters are NOT normalised: their kernel layouts are refreshed
        only when the parameters change (_refresh_generator_filters)."""
        names = []
        for grp in which:
            if grp is not self.PG:
                names += self._filter_names(grp)
        # the projection head's label embeddings depend on parameters only: they ride in the same launch (RCGAN_HEAD_EMBED_RIDE=0:
        # the head computes them in a launch of its own, in the middle of the step's dependency chain)
        g, embed = self.graph, None
        g.head_E = None
        if self.fused_head and self.PD in which and head_update is not False and os.environ.get("RCGAN_HEAD_EMBED_RIDE", "1") == "1":
            _, _, table, w_e, b_e = _head_weights(head_update)
            E = self.ctx.empty((table.shape[0], w_e.param.shape[-1]), L.F32)
            embed = (table, w_e, b_e, E)
        # every fragment-major filter copy of the critic: the 8x8 stage's (ops.d_trunk) and the register-filter layers' (ops.conv2d ->
        # rcgan_conv2d_rf: D.Block.2.Conv1, 16 x 16 x 128) -- one launch behind the preparation (rf_fragments_kernel); or, measured slower
        # and off by default (ops.FRAG_IN_PREPARE), fragment rows of the preparation launch itself.
        g.trunk_frag = None
        trunk, rf = None, []
        if self.PD in which and self.ctx.act_dtype != L.F32 and (FUSED_TRUNK or O.RF_CONV):
            sn = lambda n: g.sn["Discriminator/%s/Filters" % n][0]
            trunk = [sn("D.Block.%d.%s" % (b, c)) for b in (3, 4, 5, 6) for c in ("Conv1", "Conv2")] if FUSED_TRUNK else None
            if O.RF_CONV:
                d = L.ConvDesc(1, 16, 16, DIM_D, DIM_D, 3, 3, 1, self.ctx.act_dtype, L.CONV_IN_RELU)
                if self.ctx.lib.rcgan_conv_rf_ok(C.byref(d)):
                    rf.append((sn("D.Block.2.Conv1"), d))
        if (trunk or rf) and O.FRAG_IN_PREPARE:
            tf, reqs = O.fragment_requests(self.ctx, trunk, rf)
            rode, written = g.prepare_convs(names, self.ctx.act_dtype, embed=embed, inputs=inputs, frags=reqs)
            if len(written) == len(reqs):
                g.trunk_frag = tf
            else:
                # (some of these filters were prepared earlier in the step: their row-major copies exist, the separate launch re-lays them)
                g.trunk_frag = O.fragments_batch(self.ctx, trunk, rf)
        else:
            rode = g.prepare_convs(names, self.ctx.act_dtype, embed=embed, inputs=inputs)
            if trunk or rf:
                g.trunk_frag = O.fragments_batch(self.ctx, trunk, rf)
        if rode and embed is not None:
            g.head_E = embed[3]
        return rode

    def _refresh_generator_filters(self):
        if self._pg_prepared_version != self.PG.version:
            self.graph.refresh_persistent(self._filter_names(self.PG), self.ctx.act_dtype)
            self._pg_prepared_version = self.PG.version

    def confusion_matrix(self):
        ctx = self.ctx
        if self.PC is not None:
            return O.softmax_rows(ctx, self.graph.param("confusion_logits"))      # gan_resnet.py:522
        return self.inp["C_const"]                                               # gan_resnet.py:524

    # ---------------------------------------------------------------------------------- D step
    def _rides_inputs(self):
        return self.device_rng and self.ride_inputs and self.PD.gradbuf.numel() % 4 == 0

    def _d_body(self, fakes_ready=False):
        """Forward + backward of disc_cost (gan_resnet.py:557-697) on this rank's shard.  fakes_ready: the generator
        forward of this step was evaluated by prepare_critic_fakes; its images are in the fake rows of self.x_all."""
        ctx, g, B, inp = self.ctx, self.graph, self.B, self.inp
        self._refresh_generator_filters()       # no-op unless the generator changed behind d_step/g_step's back
        ctx.new_step()
        g.begin_step({1})
        ctx.sn_adam = (dict(group=self.PD, beta1=0.0, beta2=0.9, grad_scale=1.0 / (self.world * self.loss_scale))
                       if (self.fused_tail and ctx.recording and self.PD.hyper is not None) else None)
        g.early_d = self._dp_early(self.PD) if (self.dp_overlap and ctx.recording) else None
        # With the fakes ready and the noise drawn on the device, everything at the head of the step that depends on its inputs
        # only -- noise, preprocessing, the image pool of D.Block.1's shortcut, the zero-fill -- rides in the filter-preparation
        # launch (rcgan_conv_prepare_batch_riders) instead of five launches in front of the first convolution.
        ride = fakes_ready and self._rides_inputs()
        x_all = self.x_all if fakes_ready else ctx.empty((2 * B, OUTPUT_DIM))
        real, fake_dst = x_all.rows(0, B), x_all.rows(B, 2 * B)
        si = None
        if ride:
            fptr, fcount = self.PD.zero_grad(defer=True)
            g.image_pool = ctx.empty((2 * B, 16, 16, 3), x_all.dtype)
            si = L.StepInputsDesc(B, real.dtype, inp["images"].ptr, x_all.ptr, g.image_pool.ptr,
                                  0.0, 1.0 / 128, self.seed * 1000003 + self.rank, self.rng_state.data_ptr(), fptr, fcount,
                                  self.fakes_all.ptr, self.slice_ctr.data_ptr(), N_CRITIC)       # the launch also fetches its fake batch
        else:
            g.image_pool = None
            self.PD.zero_grad()
            if self.device_rng:
                self._rng(inp["noise"], 0, 0.0, 1.0 / 128)
                if not fakes_ready:
                    self._rng(inp["z"], 1, 0.0, 1.0)
        g.prefetch_sn(self._sn_entries(True, True))
        rode = self._prepare_all((self.PD,) if fakes_ready else (self.PG, self.PD), head_update=None, inputs=si)
        assert rode or si is None, "the step-input rider needs the batched filter preparation"
        # [real ; fake] is ONE discriminator pass for every algorithm: there is no norm layer in D and the spectral-norm
        # weights of a step are computed once (prefetch_sn above), so D(real) and D(fake) of the reference's rcgan-u graph
        # (:654-660) see the same filters and the trunk is evaluated on the 2B rows together
        if si is None:
            ctx.check(ctx.lib.rcgan_preprocess_cifar(ctx.h, B, inp["images"].ptr, inp["noise"].ptr, real.dtype, real.ptr))
        if fakes_ready:
            fake = self.x_all.rows(B, 2 * B)
        else:
            fake = Generator(B, inp["labels_random"], inp["z"], out=fake_dst)             # :540-546
        w = 1.0       # (the gradient scale of 16-bit activations is applied inside the loss kernels: rcgan_set_grad_scale)
        if self.fused_head:
            feat = Discriminator(x_all, None, update_collection=None, _head=False)        # :584
            lab_r, lab_f = inp["labels_all"].rows(0, B), inp["labels_all"].rows(B, 2 * B)
            if self.alg == "rcgan-u":          # fake logits for every label, weighted by the confusion rows (:654-684)
                y_conf = O.gather_rows(ctx, self.confusion_matrix(), inp["labels_random"], B)
                parts = [(B, L.LOSS_HINGE_REAL, inp["labels"], None), (B, L.LOSS_HINGE_FAKE, None, y_conf)]
            elif self.alg == "unbiased":       # real logits for every label, weighted by the C^-1 rows (:613-648)
                parts = [(B, L.LOSS_HINGE_REAL, None, inp["inv_weights"]), (B, L.LOSS_HINGE_FAKE, inp["labels_random"], None)]
            else:                              # biased / rcgan (:585-606): labels_all = [labels ; labels_random | labels_biased]
                parts = [(B, L.LOSS_HINGE_REAL, lab_r, None), (B, L.LOSS_HINGE_FAKE, lab_f, None)]
            Discriminator_head(feat, parts, w, self.loss_d, update_collection=None, logits=self.head_logits)
        elif self.alg == "rcgan-u":
            feat_a, wgan_a = Discriminator(x_all, None, update_collection=None)
            feat, wgan = O.rows(ctx, feat_a, 0, B), O.rows(ctx, wgan_a, 0, B)
            feat_f, wgan_f = O.rows(ctx, feat_a, B, 2 * B), O.rows(ctx, wgan_a, B, 2 * B)
            emb = Discriminator_projection(inp["labels"], update_collection=None)
            disc_real = O.proj_logit(ctx, feat, wgan, emb)
            E = Discriminator_projection(inp["arange"], update_collection=None)
            disc_fake = O.proj_logit_all(ctx, feat_f, wgan_f, E)                         # :654-660
            y_conf = O.gather_rows(ctx, self.confusion_matrix(), inp["labels_random"], B)   # :682-683
            O.loss_term(ctx, L.LOSS_HINGE_FAKE, disc_fake, w, self.loss_d, wts=y_conf)   # :673,684
            O.loss_term(ctx, L.LOSS_HINGE_REAL, disc_real, w, self.loss_d)               # :674
        else:
            feat, wgan = Discriminator(x_all, None, update_collection=None)              # :584
            if self.alg in ("biased", "rcgan"):
                emb = Discriminator_projection(inp["labels_all"], update_collection=None)    # :585
                disc_all = O.proj_logit(ctx, feat, wgan, emb)                            # :588
                O.loss_term(ctx, L.LOSS_HINGE_REAL, O.rows(ctx, disc_all, 0, B), w, self.loss_d)       # :604
                O.loss_term(ctx, L.LOSS_HINGE_FAKE, O.rows(ctx, disc_all, B, 2 * B), w, self.loss_d)   # :605
            else:   # unbiased (:613-648): every label's projection once, real loss weighted by C^-1 rows
                E = Discriminator_projection(inp["arange"], update_collection=None)
                feat_r, wgan_r = O.rows(ctx, feat, 0, B), O.rows(ctx, wgan, 0, B)
                feat_f, wgan_f = O.rows(ctx, feat, B, 2 * B), O.rows(ctx, wgan, B, 2 * B)
                disc_real_all = O.proj_logit_all(ctx, feat_r, wgan_r, E)                 # [B,10]
                O.loss_term(ctx, L.LOSS_HINGE_REAL, disc_real_all, w, self.loss_d, wts=inp["inv_weights"])   # :647
                emb_f = Discriminator_projection(inp["labels_random"], update_collection=None)
                disc_fake = O.proj_logit(ctx, feat_f, wgan_f, emb_f)
                O.loss_term(ctx, L.LOSS_HINGE_FAKE, disc_fake, w, self.loss_d)           # :639,648
        if self.perm:
            logits = perm_classifier(real, self.perm_type)                               # :692
            O.bce_onehot_term(ctx, logits, inp["labels"], 1.0, self.loss_d)                          # :693-695
        ctx.backward()
        self._tail_fused[bool(fakes_ready)] = bool(ctx.sn_adam and ctx.sn_adam.get("done"))
        ctx.sn_adam = None
        self._dp_finish([self.PD])

    # ---------------------------------------------------------------------------------- G step
    def _g_body(self):
        """Forward + backward of gen_cost (gan_resnet.py:715-786) on this rank's shard."""
        ctx, g, B, inp = self.ctx, self.graph, self.B, self.inp
        n = GEN_BS_MULTIPLE * B
        self._refresh_generator_filters()
        ctx.new_step()
        g.begin_step({0, 2} if self.PC is not None else {0})
        ctx.sn_adam = None
        g.early_g = self._dp_early(self.PG) if (self.dp_overlap and ctx.recording) else None
        self.PG.zero_grad()
        if self.PC is not None:
            self.PC.zero_grad()
        if self.device_rng:
            self._rng(inp["z_G"], 1, 0.0, 1.0)
        g.prefetch_sn(self._sn_entries(False, True))        # D convs + D.Output: NO_OPS; projection / perm: update
        self._prepare_all((self.PG, self.PD), head_update=NO_OPS)
        fake = Generator(n, inp["labels_random_G"], inp["z_G"])                                      # :719
        lab = inp["labels_random_G"] if self.alg in ("biased", "unbiased") else inp["labels_biased_G"]
        if self.fused_head:
            feat = Discriminator(fake, lab, update_collection=NO_OPS, _head=False)                   # :721-730
            if self.alg == "rcgan-u":
                y_conf = O.gather_rows(ctx, self.confusion_matrix(), inp["labels_random_G"], n)      # :757-758
                parts = [(n, L.LOSS_NEG_MEAN, None, y_conf)]                                         # :751,759
            else:
                parts = [(n, L.LOSS_NEG_MEAN, lab, None)]                                            # :763,773
            Discriminator_head(feat, parts, 1.0, self.loss_g, update_collection=NO_OPS, logits=self.head_logits)
        else:
            self._g_head_unfused(fake, lab, n)
        if self.perm:
            logits = perm_classifier(fake, self.perm_type)                                           # :781
            O.bce_onehot_term(ctx, logits, inp["labels_random_G"], self.perm_mult, self.loss_g)      # :782-784
        ctx.backward()
        self._dp_finish([self.PG] + ([self.PC] if self.PC is not None else []))

    def _g_head_unfused(self, fake, lab, n):
        ctx, inp = self.ctx, self.inp
        feat, wgan = Discriminator(fake, lab, update_collection=NO_OPS)                              # :721-730
        if self.alg == "rcgan-u":
            E = Discriminator_projection(inp["arange"], update_collection=None)                      # :736
            disc_fake = O.proj_logit_all(ctx, feat, wgan, E)
            y_conf = O.gather_rows(ctx, self.confusion_matrix(), inp["labels_random_G"], n)          # :757-758
            O.loss_term(ctx, L.LOSS_NEG_MEAN, disc_fake, 1.0, self.loss_g, wts=y_conf)               # :751,759
        else:
            emb = Discriminator_projection(lab, update_collection=None)                              # :725,731
            disc_fake = O.proj_logit(ctx, feat, wgan, emb)                                           # :763
            O.loss_term(ctx, L.LOSS_NEG_MEAN, disc_fake, 1.0, self.loss_g)                           # :773

    # ---------------------------------------------------------------------------------- stepping
    def _run(self, key, body, ctx=None):
        ctx = ctx or self.ctx
        if not self.use_graphs:
            body()
            return
        if key not in self._graphs:
            body()                      # eager warm-up (module loads, LDS attributes, self test)
            ctx.sync()
            ctx.graph_begin()
            try:
                body()
                gid = ctx.graph_end()
            except L.RcganError as e:
                ctx.graph_abort()
                if not (self.dp_active and e.code == L.RCGAN_ERCCL):
                    raise
                # a communicator whose collectives cannot be recorded: this step keeps running launch by launch (every rank runs
                # the same software and takes the same branch, so the ranks stay in step)
                warnings.warn("step %r: the all-reduce could not be captured (%s); running it uncaptured" % (key, e))
                gid = None
            except BaseException:
                ctx.graph_abort()       # never leave the stream in capture mode (any later launch or sync would fail)
                raise
            self._graphs[key] = gid
            return                      # the warm-up execution already did this step's work
        if self._graphs[key] is None:
            body()
            return
        ctx.graph_launch(self._graphs[key])

    # ---------------------------------------------------------------------------------- data parallel
    def _dp_early(self, grp):
        """Tape closure for the point of the backward pass where the group's LAST layers are done (RCGAN_DP_OVERLAP=1): flush their
        filter gradients, run their spectral-norm backward, and start the all-reduce of slab[lo:] on the communication stream."""
        ctx, lo = self.ctx, self._early_lo[id(grp)]
        state = {"sent": False}
        grp._dp_early_state = state

        def fire():
            ctx.flush_wgrads()
            for bw in ctx.sn_partial:
                bw(lambda name: name in grp.offsets and grp.offsets[name] >= lo)
            ctx.check(ctx.lib.rcgan_allreduce_sum_async(ctx.h, C.c_void_p(grp.grad.data_ptr() + 4 * lo), grp.count - lo))
            state["sent"] = True
        return fire

    def _dp_finish(self, groups):
        """End of a step's backward pass: all-reduce the gradient slabs of the step's optimiser groups (ONE RCCL group; what an early
        bucket has not taken yet) and -- static loss scale -- run the optimiser inside the same captured graph."""
        ctx = self.ctx
        if not ctx.recording:                         # (forward-only evaluations -- eval_d_cost -- exchange and update nothing)
            return
        if not self.dp_active:
            if self.graph_adam:
                for grp in groups:
                    grp.adam_captured(0.0, 0.9, grad_scale=1.0 / (self.world * self.loss_scale))
            return
        ptrs, counts = [], []
        for grp in groups:
            st = getattr(grp, "_dp_early_state", None)
            hi = self._early_lo[id(grp)] if (st is not None and st["sent"]) else grp.count
            grp._dp_early_state = None
            if hi > 0:
                ptrs.append(grp.grad.data_ptr())
                counts.append(hi)
        n = len(ptrs)
        if self.grad_bucket_dtype == "bf16":
            cnt = (C.c_size_t * n)(*counts)
            need = ctx.lib.rcgan_allreduce_bf16_scratch_bytes(n, cnt)
            if self._bucket16 is None:           # persistent (captured graphs keep its address): sized for all groups together
                allc = [g.count for g in self.groups]
                self._bucket16 = torch.empty(ctx.lib.rcgan_allreduce_bf16_scratch_bytes(len(allc), (C.c_size_t * len(allc))(*allc)),
                                             dtype=torch.uint8, device=ctx.device)
            assert need <= self._bucket16.numel()
            ctx.check(ctx.lib.rcgan_allreduce_sum_bf16_buckets(ctx.h, n, (C.c_void_p * n)(*ptrs), cnt, C.c_void_p(self._bucket16.data_ptr()),
                                                               self._bucket16.numel()))
        else:
            ctx.check(ctx.lib.rcgan_allreduce_sum_buckets(ctx.h, n, (C.c_void_p * n)(*ptrs), (C.c_size_t * n)(*counts)))
        if self.dp_overlap:
            ctx.check(ctx.lib.rcgan_allreduce_join(ctx.h))      # the step's stream waits for the early buckets
        if self.dp_adam_in_graph:
            for grp in groups:
                grp.adam_captured(0.0, 0.9, grad_scale=1.0 / (self.world * self.loss_scale))

    def _optimise_or_publish(self, steps):
        """After a step's launch: the optimiser (eager) -- or, when it ran inside the step's graph, only its host-side bookkeeping."""
        if self.graph_adam:
            for grp, _ in steps:
                grp.version += 1
            return
        self._optimise(steps)

    def _pre_step(self, steps):
        """In front of a step's launch: {lr, t} of the optimiser launches the step's graph contains."""
        if self.fused_tail:
            for grp, lr in steps:
                if grp is self.PD and (grp.hyper is None or getattr(grp, "_dev_hyper", None) != (float(lr), grp.t)):
                    grp.set_hyper_device(lr, grp.t)          # t = updates applied so far; the step's launches advance it themselves
                    grp._dev_hyper = (float(lr), grp.t)
        if self.graph_adam:
            for grp, lr in steps:
                grp.t += 1
                grp.set_hyper_device(lr, grp.t)

    def _gf_body(self):
        ctx, g, B, inp = self.ctx, self.graph, self.B, self.inp
        self._refresh_generator_filters()
        ctx.new_step()
        g.begin_step(set())
        rec, ctx.recording = ctx.recording, False
        try:
            if self.device_rng:
                self._rng(inp["z_all"], 1, 0.0, 1.0)
            self._prepare_all((self.PG,))
            Generator(N_CRITIC * B, inp["labels_random_all"], inp["z_all"], out=self.fakes_all, segments=N_CRITIC)
        finally:
            ctx.recording = rec

    def _gf_chunk_body(self, k):
        """Generator() of critic step k (gan_resnet.py:540-546) on the second context: its own z, labels and batch-norm statistics."""
        ctx2, g2, B, inp = self.ctx2, self.graph2, self.B, self.inp
        ctx2.new_step()
        g2.begin_step(set())
        rec, ctx2.recording = ctx2.recording, False
        try:
            z = inp["z_all"].rows(k * B, (k + 1) * B)
            if self.device_rng:
                # (a random stream of its own: the step stream draws the dequantisation noise from rng_state at the same time)
                ctx2.check(ctx2.lib.rcgan_rng_fill(ctx2.h, z.size, z.dtype, 1, 0.0, 1.0, self.seed * 1000003 + self.rank + 500009,
                                                   C.c_void_p(self.rng_state_gf.data_ptr()), C.c_void_p(z.ptr)))
            Generator(B, inp["labels_random_all"].rows(k * B, (k + 1) * B), z, out=self.fakes_all.rows(k * B, (k + 1) * B))
        finally:
            ctx2.recording = rec

    def _join_gf(self):
        """The step stream waits for every generator-forward pass still in flight: in front of anything that WRITES what they read
        (the generator's update, a checkpoint load)."""
        if self.overlap_gf:
            for k in range(N_CRITIC):
                if self._gf_pending[k]:
                    self.ctx.stream.wait_event(self._gf_done[k])
                    self._gf_pending[k] = False

    def _prepare_critic_fakes_overlapped(self):
        s1, s2 = self.ctx.stream, self.ctx2.stream
        self._refresh_generator_filters()
        # everything the passes read is final on the step stream here: the generator's parameters and prepared filters (the last
        # g_step), the "gf" feed; and the critic steps that read the previous fakes have been enqueued in front of this point
        ev = torch.cuda.Event()
        ev.record(s1)
        s2.wait_event(ev)
        for k in range(N_CRITIC):
            self._run("gf%d" % k, lambda k=k: self._gf_chunk_body(k), ctx=self.ctx2)
            self._gf_done[k].record(s2)
            self._gf_pending[k] = True

    def prepare_critic_fakes(self):
        """The generator does not change during the N_CRITIC critic updates of an iteration (gan_resnet.py:928-947), so their
        N_CRITIC Generator() calls (:540-546, one fresh z and label draw each) are evaluated here as ONE pass over
        N_CRITIC*B samples -- same values: the convolutions are per-sample, every conditional batch norm takes its
        statistics per step's batch (segments) -- and the next N_CRITIC d_step() calls consume one slice each.
        Inputs: labels_random_all [N_CRITIC*B] (feed "gf"; slice k must equal step k's labels_random), z_all (drawn on the
        device unless device_rng=False)."""
        if self.overlap_gf:
            self._prepare_critic_fakes_overlapped()
        else:
            self._run("gf", self._gf_body)
        self._fakes_left = N_CRITIC
        if self._slice_mirror != 0:          # the previous batch was not used up: the device's slice counter goes back to 0
            with torch.cuda.stream(self.ctx.stream):
                self.slice_ctr.zero_()
            self._slice_mirror = 0

    def d_step(self, iteration=None):
        """One critic update (disc_train_op, gan_resnet.py:802-804) on the current static inputs."""
        it = self.iteration if iteration is None else iteration
        self._refresh_generator_filters()
        steps = [(self.PD, self.lr * lr_decay(it))]
        self._pre_step(steps)
        if self._fakes_left > 0:
            k = N_CRITIC - self._fakes_left
            self._fakes_left -= 1
            ctx = self.ctx
            if self.overlap_gf and self._gf_pending[k]:
                ctx.stream.wait_event(self._gf_done[k])      # slice k of fakes_all is written by the generator-forward stream
                self._gf_pending[k] = False
            if self._rides_inputs():
                # the step's launch fetches slice slice_ctr (a device counter it moves on itself); the host mirrors it
                assert self._slice_mirror == k, (self._slice_mirror, k)
                self._slice_mirror = (k + 1) % N_CRITIC
            else:
                with torch.cuda.stream(ctx.stream):
                    ctx.view(self.x_all.rows(self.B, 2 * self.B)).copy_(ctx.view(self.fakes_all.rows(k * self.B, (k + 1) * self.B)), non_blocking=True)
            self._run("d_fakes", lambda: self._d_body(True))
            fused = self._tail_fused.get(True, False)
        else:
            self._run("d", self._d_body)
            fused = self._tail_fused.get(False, False)
        if fused:
            # the step's last launch applied the update (rcgan_sn_bwd_adam): host-side bookkeeping only
            lr = steps[0][1]
            self.PD.t += 1
            self.PD._dev_hyper = (float(lr), self.PD.t)
            self.PD.version += 1
            return
        self._optimise_or_publish(steps)

    def _critic_graph_ok(self):
        """The N_CRITIC critic steps can run as one graph: the optimiser lives in the step's last launch with its step count on the
        device (fused_tail), the fakes of all steps are ready, and the step's launch fetches its own inputs (noise, fake slice)."""
        return (self.critic_graph and self.fused_tail and self._fakes_left == N_CRITIC and self._rides_inputs() and self._slice_mirror == 0)

    def critic_steps(self, batches, iteration=None):
        """The N_CRITIC critic updates of one iteration (gan_resnet.py:928-947) after prepare_critic_fakes().  batches: N_CRITIC dicts
        of host arrays (the "d" feed's fields), or a device int32 tensor [N_CRITIC, words] of packed batches (pack_feed).
        Where _critic_graph_ok(): ONE hand-over of the five batches and ONE captured graph of the five steps -- the same launches in
        the same order as five d_step() calls (tests: bit-identical weights), without the four graph-to-graph turnarounds (~9 us from
        a graph's last kernel to the next packet) and the four feed copies between them.  Otherwise: five d_step() calls."""
        it = self.iteration if iteration is None else iteration
        packed = isinstance(batches, torch.Tensor)
        assert (batches.shape[0] if packed else len(batches)) == N_CRITIC
        if not self._critic_graph_ok():
            for k in range(N_CRITIC):
                if packed:
                    self.set_feed("d", batches[k])
                else:
                    self.feed_host("d", **batches[k])
                self.d_step(iteration=it)
            return
        ctx = self.ctx
        if packed:
            src = batches.reshape(-1)
            assert src.is_cuda and src.dtype == torch.int32 and src.is_contiguous() and src.numel() == self.feed["d5"].numel()
            ctx.check(ctx.lib.rcgan_copy_words(ctx.h, src.numel(), C.c_void_p(src.data_ptr()), C.c_void_p(self.feed["d5"].data_ptr())))
        else:
            self.feed_host("d5", _steps=batches)
        self._refresh_generator_filters()
        lr = self.lr * lr_decay(it)
        self._pre_step([(self.PD, lr)])
        self._join_gf()

        def body():
            for k in range(N_CRITIC):
                saved = {n: self.inp[n] for n in self._d_slot_views[k]}
                self.inp.update(self._d_slot_views[k])
                try:
                    self._d_body(True)
                finally:
                    self.inp.update(saved)
        self._run("d5", body)
        assert self._tail_fused.get(True, False), "critic_steps: the step's last launch did not apply the update"
        self._fakes_left = 0                    # (the device's slice counter has walked all N_CRITIC slices and is back at 0)
        self.PD.t += N_CRITIC
        self.PD._dev_hyper = (float(lr), self.PD.t)
        self.PD.version += 1

    def g_step(self, iteration=None):
        """One generator update (+ confusion-matrix update for rcgan-u): gen_train_op, confuse_train_op
        (gan_resnet.py:806-817)."""
        it = self.iteration if iteration is None else iteration
        self._fakes_left = 0                 # images prepared by prepare_critic_fakes belong to the generator before this update
        self._join_gf()
        self._refresh_generator_filters()
        steps = [(self.PG, self.lr * lr_decay(it))]
        if self.PC is not None:
            steps.append((self.PC, self.lr * self.confuse_multiplier * (lr_decay(it) if self.confuse_lr_decay else 1.0)))
        self._pre_step(steps)
        self._run("g", self._g_body)
        self._optimise_or_publish(steps)
        self._refresh_generator_filters()

    def _optimise(self, steps):
        """TF-form Adam (beta1 0, beta2 0.9: gan_resnet.py:802-817) on the groups of one optimiser step: [(group, lr)].  The gradient
        slabs hold world * scale times the mean gradient (all-reduce SUM over the ranks, loss scale of the 16-bit build)."""
        ctx = self.ctx
        if not self.dynamic_ls:
            for grp, lr in steps:
                grp.t += 1
                grp.set_hyper(lr, grp.t)
                grp.adam(0.0, 0.9, grad_scale=1.0 / (self.world * self.loss_scale))
            return
        # dynamic loss scale: one verdict for the whole step (an overflow anywhere skips every group's update)
        for grp, lr in steps:
            grp.t += 1                      # nominal count; the bias correction uses the device's count of APPLIED updates
            grp.finite_check(self.ls_state)
        for grp, lr in steps:
            grp.set_hyper(lr, grp.t)
            grp.adam_dyn(self.ls_state, 0.0, 0.9, grad_scale=1.0 / self.world)
        tds = [C.c_void_p(g.t_dev.data_ptr()) for g, _ in steps] + [None]
        ctx.check(ctx.lib.rcgan_loss_scale_update(ctx.h, C.c_void_p(self.ls_state.data_ptr()), tds[0], tds[1], self.ls_growth_interval,
                                                  1.0, float(1 << 24)))

    def loss_scale_state(self):
        """{scale, good_steps, skipped_steps} of the dynamic loss scale as it stands on the stream now (synchronises)."""
        if self.ls_state is None:
            return dict(scale=self.loss_scale, good_steps=None, skipped_steps=0)
        self.ctx.sync()
        s = self.ls_state.cpu().numpy()
        return dict(scale=float(s[0]), good_steps=int(s[1]), skipped_steps=int(s[3]))

    def losses(self):
        ctx = self.ctx
        return float(ctx.download(self.loss_d)[0]), float(ctx.download(self.loss_g)[0])

    def enqueue_losses(self):
        """Asynchronous read-back of (disc_cost, gen_cost) as they stand on the stream now: two device scalars -> one slot of a
        pinned ring (1024 slots), no host-device synchronisation.  Returns a ticket for fetch_losses; a ticket must be
        fetched before 1024 newer ones exist."""
        if not hasattr(self, "_loss_ring"):
            self._loss_ring = torch.zeros(1024, 2, dtype=torch.float32).pin_memory()
            self._loss_tickets = 0
        i = self._loss_tickets % 1024
        self._loss_tickets += 1
        ctx = self.ctx
        with torch.cuda.stream(ctx.stream):
            self._loss_ring[i, 0:1].copy_(ctx.view(self.loss_d), non_blocking=True)
            self._loss_ring[i, 1:2].copy_(ctx.view(self.loss_g), non_blocking=True)
        return self._loss_tickets - 1

    def fetch_losses(self, tickets):
        """[(disc_cost, gen_cost)] of the given tickets (one stream synchronisation for all of them)."""
        if not tickets:
            return []
        assert self._loss_tickets - min(tickets) <= 1024, "loss ring overrun: fetch at least every 1024 tickets"
        self.ctx.sync()
        return [(float(self._loss_ring[t % 1024, 0]), float(self._loss_ring[t % 1024, 1])) for t in tickets]

    def eval_d_cost(self):
        """disc_cost on the current D-step inputs, forward only -- the reference's dev-cost pass ``session.run([disc_cost])``
        (gan_resnet.py:977-990).  As there it is a training-mode evaluation of the graph: a fresh Generator() batch, batch
        statistics, and the spectral-norm ``u`` vectors take their power-iteration update (deterministic from W and u, so
        ranks stay identical as long as every rank makes the same calls).  No gradient, no optimiser step."""
        ctx = self.ctx
        left, self._fakes_left = self._fakes_left, 0
        rec, ctx.recording = ctx.recording, False
        try:
            self._d_body(False)
        finally:
            ctx.recording = rec
            self._fakes_left = left
        return float(ctx.download(self.loss_d)[0])

    # ---------------------------------------------------------------------------------- inspection
    def get_params(self):
        out = {}
        for grp in self.groups:
            for n in grp.names:
                out[n] = grp.get(n)
        return out

    def get_grads(self, group):
        """Gradients of the last step (the loss scale of the fp16 build divided out; a power of two, exact).  Under dynamic loss
        scaling: the scale the step RAN with -- if the step's verdict changed it afterwards (overflow -> halved, growth -> doubled)
        ask before the optimiser launch or accept the factor of two."""
        scale = self.loss_scale_state()["scale"] * (self.world if self.dp_active else 1)      # data parallel: the slab holds the SUM over the ranks
        return {n: group.get(n, "grad") / np.float32(scale) for n in group.names}

    def get_state(self):
        return {k: self.ctx.download(v).reshape(1, -1) for k, v in self.state.items()}

    # ---------------------------------------------------------------------------------- checkpoints
    def state_dict(self):
        """Everything tf.train.Saver would store (gan_resnet.py:906): variables, Adam slots (TF slot names
        "<var>/Adam", "<var>/Adam_1"), the optimisers' step counters and the SN ``u`` vectors."""
        out = {}
        for gname, grp in zip(("Generator", "Discriminator", "confusion"), self.groups):
            for n in grp.names:
                out[n] = grp.get(n)
                out[n + "/Adam"] = grp.get(n, "m")
                out[n + "/Adam_1"] = grp.get(n, "v")
            out["_opt/%s/step" % gname] = np.array([grp.steps_applied()], np.int64)
        for k, v in self.get_state().items():
            out[k] = v
        out["_iteration"] = np.array([self.iteration], np.int64)
        # AdamOptimizer's own step state, in the reference's creation order: disc_opt, gen_opt, confusion (:802-817)
        from .host import adam_power_tensors
        opts = [(self.PD.steps_applied(), 0.0, 0.9), (self.PG.steps_applied(), 0.0, 0.9)] + \
            ([(self.PC.steps_applied(), 0.0, 0.9)] if self.PC is not None else [])
        if self.ls_state is not None:
            out["_loss_scale"] = self.ls_state.cpu().numpy().astype(np.float32)
        out.update(adam_power_tensors(opts))
        return out

    def load_state_dict(self, sd):
        self._fakes_left = 0
        if self.overlap_gf:
            self.ctx2.sync()                 # (host-side writes to the parameters follow)
            self._gf_pending = [False] * N_CRITIC
        for gname, grp in zip(("Generator", "Discriminator", "confusion"), self.groups):
            for n in grp.names:
                if n not in sd:
                    raise KeyError("checkpoint is missing variable %s" % n)
                grp.set(n, sd[n])
                if n + "/Adam" in sd:
                    grp.set(n, sd[n + "/Adam"], "m")
                    grp.set(n, sd[n + "/Adam_1"], "v")
            key = "_opt/%s/step" % gname
            if key in sd:
                grp.t = int(np.asarray(sd[key]).reshape(-1)[0])
            else:                        # a bundle written by TensorFlow: recover the step from beta2_power
                from .host import steps_from_beta_power
                sfx = {"Discriminator": "", "Generator": "_1", "confusion": "_2"}[gname]
                if "beta2_power" + sfx in sd:
                    grp.t = steps_from_beta_power(float(np.asarray(sd["beta2_power" + sfx])), 0.9)
        for grp in self.groups:
            grp.t_dev = None                 # re-created from grp.t by the next dynamic-loss-scale update
        if self.ls_state is not None and "_loss_scale" in sd:
            with torch.cuda.stream(self.ctx.stream):
                self.ls_state.copy_(torch.from_numpy(np.asarray(sd["_loss_scale"], np.float32).reshape(4)))
        if "_iteration" in sd:
            self.iteration = int(np.asarray(sd["_iteration"]).reshape(-1)[0])
        ctx = self.ctx
        for k, t in self.state.items():
            if k in sd:
                with torch.cuda.stream(ctx.stream):
                    ctx.view(t).copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(sd[k], np.float32).reshape(-1))))
        ctx.sync()

    def confusion_matrix_value(self):
        """The confusion matrix the losses use, as a host array [10,10]: softmax of the learned logits (rcgan-u) or the
        fixed one-coin matrix (gan_resnet.py:499-524)."""
        ctx = self.ctx
        if self.PC is None:
            return ctx.download(self.inp["C_const"]).reshape(VOCAB_SIZE, VOCAB_SIZE)
        logits = self.PC.get("confusion_logits").astype(np.float64)
        e = np.exp(logits - logits.max(axis=1, keepdims=True))
        return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)

    def sample(self, labels, z, segments=1):
        """Generator forward only (fixed_noise_samples, gan_resnet.py:827); returns [n,3072] float32.
        segments: evaluate that many independent batches (own batch-norm statistics each) in one pass."""
        ctx, g = self.ctx, self.graph
        self._refresh_generator_filters()
        ctx.new_step()
        g.begin_step(set())
        rec, ctx.recording = ctx.recording, False
        try:
            lab = ctx.upload(np.asarray(labels, np.int32))
            zz = ctx.upload(np.asarray(z, np.float32))
            out = Generator(len(labels), lab, zz, segments=segments)
            res = ctx.download(out)
        finally:
            ctx.recording = rec
        return res
