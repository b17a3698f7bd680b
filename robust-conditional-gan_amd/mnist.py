"""MNIST DCGAN-style RCGAN / RCGAN-U / RCGAN+y on the gfx950 kernels.

Host-side mirror of /root/reference mnist/model.py: generator :705-731, gen_sampler :733-757, discriminator
:644-703, classifier :759-768, loss assembly :126-224, optimisers :250-262, step order :347-372
(one D run, then TWO G runs on the same batch).  The reference's five logging-only ``.eval()`` passes per
iteration (:374-398), which also mutate BN moving averages and SN ``u``, are not executed.
"""
import os
import time

import numpy as np
import torch

from . import _lib as L
from . import ops as O
from .ops_mnist import batch_norm, conv2d, conv_cond_concat, deconv2d, linear, lrelu
from .runtime import DT, Context, ParamGroup
from .variables import Graph, variable_scope

Y_DIM, Z_DIM, GF_DIM, DF_DIM, GFC_DIM, DFC_DIM = 10, 100, 64, 64, 1024, 1024


def _trunc_normal(rs, size, stddev):
    x = rs.normal(0.0, stddev, size=size)
    while True:
        bad = np.abs(x) > 2 * stddev
        if not bad.any():
            return x.astype("float32")
        x[bad] = rs.normal(0.0, stddev, size=int(bad.sum()))


def create_variables(seed=0, disc_type="projection", estimate_confuse=False, perm_regularizer=True, spectral_norm=True,
                     concat_y_layers=()):
    """(G specs, D specs, C specs, S moving stats, U sn vectors) in the reference's creation order
     with the reference's initialisers (ops.py:57-58, 74-75, 102-108; sn.py:36)."""
    rs = np.random.RandomState(seed)
    G, D, Cm, S, U = [], [], [], {}, {}

    def lin(dst, name, cin, cout):
        dst.append((name + "/Matrix", (cin, cout), rs.normal(0.0, 0.02, size=(cin, cout)).astype("float32")))
        dst.append((name + "/bias", (cout,), np.zeros(cout, "float32")))

    def bn(dst, name, c):
        dst.append((name + "/beta", (c,), np.zeros(c, "float32")))
        dst.append((name + "/gamma", (c,), np.ones(c, "float32")))
        S[name + "/moving_mean"] = np.zeros(c, "float32")
        S[name + "/moving_variance"] = np.ones(c, "float32")

    def conv(dst, name, cin, cout, sn):
        dst.append((name + "/w", (5, 5, cin, cout), _trunc_normal(rs, (5, 5, cin, cout), 0.02)))
        if sn:
            U[name + "/spectral_norm/u"] = _trunc_normal(rs, (1, cout), 1.0)
        dst.append((name + "/biases", (cout,), np.zeros(cout, "float32")))

    def deconv(dst, name, cout, cin):
        dst.append((name + "/w", (5, 5, cout, cin), rs.normal(0.0, 0.02, size=(5, 5, cout, cin)).astype("float32")))
        dst.append((name + "/biases", (cout,), np.zeros(cout, "float32")))

    if estimate_confuse:
        lim = np.sqrt(6.0 / (2 * Y_DIM))
        Cm.append(("confusion_logits", (Y_DIM, Y_DIM), rs.uniform(-lim, lim, size=(Y_DIM, Y_DIM)).astype("float32")))
    g = "generator/"
    lin(G, g + "g_h0_lin", Z_DIM + Y_DIM, GFC_DIM)
    bn(G, g + "g_bn0", GFC_DIM)
    lin(G, g + "g_h1_lin", GFC_DIM + Y_DIM, GF_DIM * 2 * 49)
    bn(G, g + "g_bn1", GF_DIM * 2 * 49)
    deconv(G, g + "g_h2", GF_DIM * 2, GF_DIM * 2 + Y_DIM)
    bn(G, g + "g_bn2", GF_DIM * 2)
    deconv(G, g + "g_h3", 1, GF_DIM * 2 + Y_DIM)
    d = "discriminator/"
    if disc_type == "projection":
        cins = [1, DF_DIM, DF_DIM, DF_DIM]
        for i in range(4):
            conv(D, d + "d_h%d_conv" % i, cins[i] + (Y_DIM if (i + 1) in concat_y_layers else 0), DF_DIM, spectral_norm)
            if i > 0:
                bn(D, d + "d_bn%d" % i, DF_DIM)
        lin(D, d + "d_h4_lin", DF_DIM, 1)
        lin(D, d + "d_h5_y_lin", Y_DIM, DF_DIM)
    else:
        conv(D, d + "d_h0_conv", 1 + Y_DIM, 1 + Y_DIM, False)
        conv(D, d + "d_h1_conv", 1 + 2 * Y_DIM, DF_DIM + Y_DIM, False)
        bn(D, d + "d_bn1", DF_DIM + Y_DIM)
        lin(D, d + "d_h3_lin", 49 * (DF_DIM + Y_DIM) + Y_DIM, DFC_DIM)
        bn(D, d + "d_bn2", DFC_DIM)
        lin(D, d + "d_h4_lin", DFC_DIM + Y_DIM, 1)
    if perm_regularizer:
        lin(D, "classifier/d_classifier_h1", 784, Y_DIM)      # lands in d_vars: its name contains 'd_' (model.py:244)
    return G, D, Cm, S, U


class MnistRCGAN:
    """DCGAN(sess, ...) + train() of the reference as a per-rank engine."""

    def __init__(self, algorithm="rcgan", alpha=0.3, batch_size=100, learning_rate=2e-4, beta1=0.5, dtype="f32", seed=0,
                 disc_type="projection", loss_fn="hinge", estimate_confuse=False, confuse_multiplier=10.0,
                 perm_regularizer=True, perm_multiplier=10.0, spectral_norm=True, max_norm=True,
                 concat_y=False, concat_y_layers=(1,), device=0, use_graphs=True, world_size=1, rank=0, variables=None,
                 confusion_matrix=None):
        if loss_fn not in ("hinge", "ce"):
            raise ValueError('Unknown self.config.loss_fn: {}!'.format(loss_fn))          # model.py:147
        if algorithm not in ("biased", "unbiased", "rcgan", "ambient"):
            raise ValueError("Unknown algorithm %s" % algorithm)
        self.alg, self.B, self.lr, self.beta1 = algorithm, int(batch_size), learning_rate, beta1
        self.disc_type, self.loss_fn, self.est = disc_type, loss_fn, bool(estimate_confuse)
        self.confuse_multiplier, self.perm, self.perm_mult = confuse_multiplier, perm_regularizer, perm_multiplier
        self.sn = spectral_norm and disc_type == "projection"
        self.max_norm = max_norm and disc_type == "projection"
        self.layers = tuple(int(v) for v in concat_y_layers) if (concat_y and disc_type == "projection") else ()
        self.world, self.rank, self.use_graphs = world_size, rank, use_graphs
        B = self.B
        self.ctx = ctx = Context(device, dtype, arena_bytes=(1 << 29) + B * (24 << 20) // 8, ws_bytes=1 << 29)
        if variables is None:
            variables = create_variables(seed, disc_type, self.est, perm_regularizer, spectral_norm, self.layers)
        gs, ds, cs, S, U = variables
        self.PG, self.PD = ParamGroup(ctx, gs), ParamGroup(ctx, ds)
        self.PC = ParamGroup(ctx, cs) if cs else None
        self.groups = [self.PG, self.PD] + ([self.PC] if self.PC else [])
        self.state = {}
        for k, v in list(S.items()) + list(U.items()):
            t = ctx.persistent((v.size,), L.F32)
            ctx.view(t).copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(v, np.float32).reshape(-1))))
            self.state[k] = t
        self.state_shapes = {k: np.shape(v) for k, v in list(S.items()) + list(U.items())}
        self.graph = Graph(ctx, self.groups, self.state)
        P = ctx.persistent
        f32, i32, act = L.F32, "i32", ctx.act_dtype
        self.inp = dict(images=P((B, 28, 28, 1), act), z=P((B, Z_DIM), act), y_real=P((B, Y_DIM), f32), y_fake=P((B, Y_DIM), f32),
                        y_gen=P((B, Y_DIM), f32), y_real_weights=P((B, Y_DIM), f32), lab_real=P((B,), i32), lab_gen=P((B,), i32),
                        eye=P((Y_DIM, Y_DIM), f32), C_const=P((Y_DIM, Y_DIM), f32))
        # the critic step's one-pass input [images ; G(z)] and its label rows [y_real ; y_fake] / [y_real ; y_gen] (set_inputs fills the
        # real halves; the generator writes the fake half of x_all)
        self.inp.update(x_all=P((2 * B, 28, 28, 1), act, fill=0.0), y_all_fake=P((2 * B, Y_DIM), f32, fill=0.0), y_all_gen=P((2 * B, Y_DIM), f32, fill=0.0))
        self.merge_critic = os.environ.get("RCGAN_MNIST_MERGE_CRITIC", "1") == "1"
        ctx.view(self.inp["eye"]).copy_(torch.eye(Y_DIM))
        if (self.alg == "unbiased" or self.est) and (self.disc_type != "projection" or self.layers):
            # one-hot(l) for every sample of copy l of the batch: the label rows of discriminator_all_labels' single pass
            self.inp["y_all_labels"] = P((Y_DIM * B, Y_DIM), f32, fill=0.0)
            ctx.view(self.inp["y_all_labels"]).copy_(torch.eye(Y_DIM).repeat_interleave(B, dim=0))
        if confusion_matrix is None:
            confusion_matrix = ((1 - alpha) / 9.0) * np.ones((10, 10)) + (alpha - (1 - alpha) / 9.0) * np.eye(10)
        self.confusion_matrix_actual = np.asarray(confusion_matrix)
        ctx.view(self.inp["C_const"]).copy_(torch.from_numpy(self.confusion_matrix_actual.astype(np.float32)))
        # one block, a 256-byte slot per term, the critic step's three in front of the generator step's two: a step clears ITS terms with one fill
        self._loss_block = P((5 * 64,), f32, fill=0.0)
        self._loss_order = ("d_loss_real", "d_loss_fake", "class_loss_real", "g_loss", "class_loss_fake")
        self.loss = {k: DT(self._loss_block.ptr + 256 * i, (1,), f32, self._loss_block.base) for i, k in enumerate(self._loss_order)}
        # bn objects as in DCGAN.__init__ (model.py:68-81)
        self.d_bn1, self.d_bn2, self.d_bn3 = batch_norm(name='d_bn1'), batch_norm(name='d_bn2'), batch_norm(name='d_bn3')
        self.g_bn0, self.g_bn1, self.g_bn2 = batch_norm(name='g_bn0'), batch_norm(name='g_bn1'), batch_norm(name='g_bn2')
        self._graphs = {}
        self.fuse_first_g = os.environ.get("RCGAN_MNIST_FUSE_G", "1") == "1"       # see _d_body_keep
        self._fused_ready = False
        # clip range of the max_norm constraint inside the D slab
        self.clip_range = None
        if self.max_norm:
            lo = self.PD.offsets["discriminator/d_h4_lin/Matrix"]
            hi = self.PD.offsets["discriminator/d_h5_y_lin/bias"] + DF_DIM
            self.clip_range = (lo, (hi + 63) // 64 * 64)
        torch.cuda.synchronize()

    # ------------------------------------------------------------------------------------ model
    def generator(self, z, y, train=True, out=None):
        ctx = self.ctx
        B = z.shape[0]
        with variable_scope("generator"):
            zz = O.concat_channels(ctx, z, y)
            h0 = self.g_bn0(linear(zz, GFC_DIM, 'g_h0_lin'), train=train, _act=L.ACT_RELU)
            h0 = O.concat_channels(ctx, h0, y)
            h1 = self.g_bn1(linear(h0, GF_DIM * 2 * 49, 'g_h1_lin'), train=train, _act=L.ACT_RELU)
            h1 = O.reshape(ctx, h1, (B, 7, 7, GF_DIM * 2))
            h1 = conv_cond_concat(h1, y)
            h2 = self.g_bn2(deconv2d(h1, [B, 14, 14, GF_DIM * 2], name='g_h2'), train=train, _act=L.ACT_RELU)
            h2 = conv_cond_concat(h2, y)
            return O.act(ctx, deconv2d(h2, [B, 28, 28, 1], name='g_h3'), L.ACT_SIGMOID, out=out)

    def _features(self, image, y, segments=1):
        """projection D up to the pooled features h3 [B,64] and h4 [B] (model.py:649-681).  segments: that many batches back to
        back, each with its own batch-norm statistics (consecutive discriminator() calls of the reference as one pass)."""
        ctx = self.ctx
        x = image
        bns = (None, self.d_bn1, self.d_bn2, self.d_bn3)
        for i in range(4):
            if (i + 1) in self.layers:
                x = conv_cond_concat(x, y)
            x = conv2d(x, DF_DIM, spectral_norm=self.sn, name='d_h%d_conv' % i)
            x = lrelu(x) if i == 0 else bns[i](x, _act=L.ACT_LRELU, _segments=segments)
        h3 = O.act_meanhw(ctx, x, L.ACT_NONE)
        h4 = O.reshape(ctx, linear(h3, 1, 'd_h4_lin', max_norm=self.max_norm), (-1,))
        return h3, h4

    def discriminator(self, image, y, segments=1):
        """-> logits [B] (h6 / h4 of model.py:685,701)."""
        ctx = self.ctx
        B = image.shape[0]
        with variable_scope("discriminator"):
            if self.disc_type == "projection":
                h3, h4 = self._features(image, y, segments)
                h5 = linear(y, DF_DIM, 'd_h5_y_lin', max_norm=self.max_norm)
                return O.proj_logit(ctx, h3, h4, h5)
            x = conv_cond_concat(image, y)
            h0 = lrelu(conv2d(x, 1 + Y_DIM, name='d_h0_conv'))
            h0 = conv_cond_concat(h0, y)
            h1 = self.d_bn1(conv2d(h0, DF_DIM + Y_DIM, name='d_h1_conv'), _act=L.ACT_LRELU, _segments=segments)
            h1 = O.concat_channels(ctx, O.reshape(ctx, h1, (B, -1)), y)
            h3 = self.d_bn2(linear(h1, DFC_DIM, 'd_h3_lin'), _act=L.ACT_LRELU, _segments=segments)
            h3 = O.concat_channels(ctx, h3, y)
            return O.cast(ctx, O.reshape(ctx, linear(h3, 1, 'd_h4_lin'), (-1,)), L.F32)

    def discriminator_all_labels(self, image):
        """logits [B,10] for every label.  Projection D without label concat: the conv stack does not depend
        on the label, so the features are computed once (the reference re-runs the whole D ten times,
        model.py:152-163,187-197)."""
        ctx = self.ctx
        if self.disc_type == "projection" and not self.layers:
            with variable_scope("discriminator"):
                h3, h4 = self._features(image, None)
                E = linear(self.inp["eye"], DF_DIM, 'd_h5_y_lin', max_norm=self.max_norm)
                return O.proj_logit_all(ctx, h3, h4, E)
        # A discriminator whose convolutions see the label (disc_type=vanilla, --concat_y): the reference's ten discriminator() calls
        # (model.py:152-163 `unbiased`, :187-197 `estimate_confuse`) as ONE pass over 10 x B samples, label-major -- copy l of the batch
        # carries one-hot(l) -- with the batch norms taking their statistics and moving-average updates per copy, in label order,
        # as the ten consecutive calls do; the [10][B] logits come back as tf.concat(D_logits_all, 1) = [B][10].
        B = image.shape[0]
        x10 = O.tile_rows(ctx, image, Y_DIM)
        logits = self.discriminator(x10, self.inp["y_all_labels"], segments=Y_DIM)
        return O.transpose2d(ctx, O.reshape(ctx, logits, (Y_DIM, B)))

    def classifier(self, x):
        ctx = self.ctx
        with variable_scope("classifier"):
            flat = O.cast(ctx, O.reshape(ctx, x, (x.shape[0], -1)), L.F32)
            return linear(flat, Y_DIM, 'd_classifier_h1')

    def confusion(self):
        if self.PC is not None:
            return O.softmax_rows(self.ctx, self.graph.param("confusion_logits"))
        return self.inp["C_const"]

    # ------------------------------------------------------------------------------------ steps
    def _kinds(self):
        if self.loss_fn == "hinge":
            return L.LOSS_HINGE_REAL, L.LOSS_HINGE_FAKE, L.LOSS_NEG_MEAN
        return L.LOSS_CE_ONES, L.LOSS_CE_ZEROS, L.LOSS_CE_ONES

    def _zero_losses(self, keys):
        ctx = self.ctx
        idx = sorted(self._loss_order.index(k) for k in keys)
        assert idx == list(range(idx[0], idx[-1] + 1)), keys            # (a step's terms are neighbours in the block)
        ctx.check(ctx.lib.rcgan_fill_f32(ctx.h, 64 * (idx[-1] - idx[0]) + 1, self._loss_block.ptr + 256 * idx[0], 0.0))

    def _sn_prefetch(self):
        if not self.sn:
            return
        ents = [(n, n[:-2] + "/spectral_norm/u", True) for n in self.PD.names if n.endswith("_conv/w")]
        self.graph.prefetch_sn(ents)
        # ... and W / sigma of all of them in ONE launch (every critic convolution is 5x5 stride 2: ops.conv2d; the layers find their filter
        # prepared) instead of one launch in front of each convolution: 12 -> 3 launches per iteration
        if os.environ.get("RCGAN_MNIST_PREPARE_BATCH", "1") == "1":
            self.graph.prepare_convs([(n, 5, 2, 8) for n, _, _ in ents], self.ctx.act_dtype)

    def _fake_branch(self, G, train_d):
        """d_loss_fake / g_loss on G(z) (model.py:179-212).  Returns nothing: loss terms record their gradients."""
        ctx, inp = self.ctx, self.inp
        kr, kf, kg = self._kinds()
        if self.alg in ("rcgan", "ambient") and self.est:
            logits = self.discriminator_all_labels(G)
            yc = O.gather_rows(ctx, self.confusion(), inp["lab_gen"], self.B)       # tensordot(y_gen, C)
            if train_d:
                O.loss_term(ctx, kf, logits, 1.0, self.loss["d_loss_fake"], wts=yc)
            else:
                O.loss_term(ctx, kg, logits, 1.0, self.loss["g_loss"], wts=yc)
        else:
            y = inp["y_fake"] if self.alg in ("rcgan", "ambient") else inp["y_gen"]
            logits = self.discriminator(G, y)
            if train_d:
                O.loss_term(ctx, kf, logits, 1.0, self.loss["d_loss_fake"])
            else:
                O.loss_term(ctx, kg, logits, 1.0, self.loss["g_loss"])

    def _merged_critic(self):
        """The critic step's D(real) and D(fake) as ONE pass over [images ; G(z)] (2B samples): the discriminator's convolutions and
        dense layers are per-sample, its batch norms take their statistics (and moving-average updates, real first) per half -- the
        values of two consecutive discriminator() calls (model.py:131-135,179-183).  Taken where both halves go through the plain
        discriminator(): projection D without label concatenation, one label per sample on both sides."""
        return (self.merge_critic and self.disc_type == "projection" and not self.layers and self.alg in ("biased", "rcgan", "ambient")
                and not (self.alg in ("rcgan", "ambient") and self.est))

    def _critic_losses_merged(self, G):
        """d_loss_real + d_loss_fake from the one pass; G = the fake half of inp['x_all'] (the generator wrote it there)."""
        ctx, inp, B = self.ctx, self.inp, self.B
        kr, kf, kg = self._kinds()
        logits = self.discriminator(inp["x_all"], inp["y_all_fake" if self.alg in ("rcgan", "ambient") else "y_all_gen"], segments=2)
        O.loss_term(ctx, kr, O.rows(ctx, logits, 0, B), 1.0, self.loss["d_loss_real"])
        O.loss_term(ctx, kf, O.rows(ctx, logits, B, 2 * B), 1.0, self.loss["d_loss_fake"])

    def _d_body(self):
        ctx, g, inp = self.ctx, self.graph, self.inp
        ctx.new_step()
        g.begin_step({1})
        self.PD.zero_grad()
        self._zero_losses(("d_loss_real", "d_loss_fake", "class_loss_real"))
        self._sn_prefetch()
        kr, kf, kg = self._kinds()
        merged = self._merged_critic()
        G = self.generator(inp["z"], inp["y_gen"], out=inp["x_all"].rows(self.B, 2 * self.B) if merged else None)      # model.py:126
        if merged:
            self._critic_losses_merged(G)
        elif self.alg in ("biased", "rcgan", "ambient"):
            O.loss_term(ctx, kr, self.discriminator(inp["images"], inp["y_real"]), 1.0, self.loss["d_loss_real"])
        else:                                                                        # unbiased (:152-174)
            logits = self.discriminator_all_labels(inp["images"])
            O.loss_term(ctx, kr, logits, 1.0, self.loss["d_loss_real"], wts=inp["y_real_weights"])
        if not merged:
            self._fake_branch(G, True)
        if self.perm:
            O.bce_onehot_term(ctx, self.classifier(inp["images"]), inp["lab_real"], 1.0, self.loss["class_loss_real"])
        ctx.backward()

    def _g_body(self):
        ctx, g, inp = self.ctx, self.graph, self.inp
        ctx.new_step()
        g.begin_step({0, 2} if self.PC is not None else {0})
        self.PG.zero_grad()
        if self.PC is not None:
            self.PC.zero_grad()
        self._zero_losses(("g_loss", "class_loss_fake"))
        self._sn_prefetch()
        G = self.generator(inp["z"], inp["y_gen"])
        self._fake_branch(G, False)
        if self.perm:
            O.bce_onehot_term(ctx, self.classifier(G), inp["lab_gen"], self.perm_mult, self.loss["class_loss_fake"])
        ctx.backward()

    def _run(self, key, body):
        ctx = self.ctx
        if not self.use_graphs:
            body()
            return
        if key not in self._graphs:
            body()
            ctx.sync()
            ctx.graph_begin()
            try:
                body()
            except BaseException:
                ctx.graph_abort()       # never leave the stream in capture mode
                raise
            self._graphs[key] = ctx.graph_end()
            return
        ctx.graph_launch(self._graphs[key])

    def _allreduce(self, group):
        if self.world > 1:
            from .dp import allreduce_sum_
            allreduce_sum_(group.grad, self.ctx.stream)

    def _d_update(self):
        self._allreduce(self.PD)
        self.PD.t += 1
        self.PD.set_hyper(self.lr, self.PD.t)
        gs = 1.0 / self.world
        if self.clip_range is None:
            self.PD.adam(self.beta1, 0.999, grad_scale=gs)
        else:                                       # variable constraint on d_h4_lin / d_h5_y_lin (ops.py:102-111)
            lo, hi = self.clip_range
            self.PD.adam(self.beta1, 0.999, grad_scale=gs, lo=0, hi=lo)
            self.PD.adam(self.beta1, 0.999, grad_scale=gs, clip=1.0, lo=lo, hi=hi)
            self.PD.adam(self.beta1, 0.999, grad_scale=gs, lo=hi, hi=self.PD.count)

    def _g_update(self):
        self._allreduce(self.PG)
        self.PG.t += 1
        self.PG.set_hyper(self.lr, self.PG.t)
        self.PG.adam(self.beta1, 0.999, grad_scale=1.0 / self.world)
        if self.PC is not None:
            self._allreduce(self.PC)
            self.PC.t += 1
            self.PC.set_hyper(self.lr * self.confuse_multiplier, self.PC.t)
            self.PC.adam(self.beta1, 0.999, grad_scale=1.0 / self.world)

    def d_step(self):
        self._run("d", self._d_body)
        self._d_update()

    def g_step(self):
        self._run("g", self._g_body)
        self._g_update()

    # The D update does not touch the generator, so the first generator step of an iteration (model.py:347-372: D once, then G
    # twice on the SAME z / y_gen) would recompute exactly the forward pass the D step just ran.  The fused iteration runs
    # that forward once: the D step records it on the tape and keeps its activations (the arena is not reset in between),
    # the first G step only runs the (updated) discriminator on the kept images and back-propagates through the kept tape.
    # The generator's batch-norm moving averages receive the two identical updates the reference applies as one update
    # with the decay squared.
    def _d_body_keep(self):
        ctx, g, inp = self.ctx, self.graph, self.inp
        ctx.new_step()
        g.begin_step({0, 1})
        bns = (self.g_bn0, self.g_bn1, self.g_bn2)
        keep = [b.momentum for b in bns]
        for b in bns:
            b.momentum = b.momentum * b.momentum
        merged = self._merged_critic()
        try:
            G = self.generator(inp["z"], inp["y_gen"], out=inp["x_all"].rows(self.B, 2 * self.B) if merged else None)   # taped: generator parameters are trainable
        finally:
            for b, m0 in zip(bns, keep):
                b.momentum = m0
        n_gen = len(ctx.tape)
        G.req = False                                                               # this step's backward stops at the images
        self.PD.zero_grad()
        self._zero_losses(("d_loss_real", "d_loss_fake", "class_loss_real"))
        self._sn_prefetch()
        kr, kf, kg = self._kinds()
        if merged:
            self._critic_losses_merged(G)
        elif self.alg in ("biased", "rcgan", "ambient"):
            O.loss_term(ctx, kr, self.discriminator(inp["images"], inp["y_real"]), 1.0, self.loss["d_loss_real"])
        else:
            logits = self.discriminator_all_labels(inp["images"])
            O.loss_term(ctx, kr, logits, 1.0, self.loss["d_loss_real"], wts=inp["y_real_weights"])
        if not merged:
            self._fake_branch(G, True)
        if self.perm:
            O.bce_onehot_term(ctx, self.classifier(inp["images"]), inp["lab_real"], 1.0, self.loss["class_loss_real"])
        self._kept = (G, list(ctx.tape[:n_gen]))
        ctx.tape = ctx.tape[n_gen:]
        ctx.backward()

    def _g_body_reuse(self):
        ctx, g, inp = self.ctx, self.graph, self.inp
        ctx.tape, ctx.pending_wgrads = [], []                                       # NOT new_step(): the arena keeps the D step's tensors
        g.begin_step({0, 2} if self.PC is not None else {0})
        self.PG.zero_grad()
        if self.PC is not None:
            self.PC.zero_grad()
        self._zero_losses(("g_loss", "class_loss_fake"))
        self._sn_prefetch()
        G, gen_tape = self._kept
        G.req, G.grad = True, None
        self._fake_branch(G, False)
        if self.perm:
            O.bce_onehot_term(ctx, self.classifier(G), inp["lab_gen"], self.perm_mult, self.loss["class_loss_fake"])
        ctx.tape = gen_tape + ctx.tape                                              # backward: discriminator first, then the kept generator tape
        ctx.backward()

    def iteration(self):
        """model.py:347-372: D once, G twice on the same fed batch."""
        if not self.fuse_first_g:
            self.d_step()
            self.g_step()
            self.g_step()
            return
        if not self._fused_ready:
            # first iteration: the plain steps run eagerly once (module loads, LDS attributes) and capture their graphs;
            # the fused bodies below are then captured WITHOUT an eager rehearsal -- they must execute exactly once per
            # capture, because the first G step works on tensor objects the D step left behind.  (Same rule without
            # graphs, so that graph replay and eager execution stay bit-identical.)
            self.d_step()
            self.g_step()
            self.g_step()
            self._fused_ready = True
            return
        self._run_once("d_keep", self._d_body_keep)
        self._d_update()
        self._run_once("g_reuse", self._g_body_reuse)
        self._g_update()
        self.g_step()

    def _run_once(self, key, body):
        ctx = self.ctx
        if not self.use_graphs:
            body()
            return
        if key not in self._graphs:
            # no eager rehearsal here: the hidden scratch of the fp32 entry points is sized up front (the capture contract of
            # include/rcgan_hip.h) -- split reductions <= 16 MiB, narrow data gradient 4 B x batch x 14 x 14 x 25 x channels
            ctx.reserve_scratch(max(16 << 20, 4 * self.B * 14 * 14 * 25 * 4))
            ctx.graph_begin()
            try:
                body()
            except L.RcganError:
                # a layer shape this bound does not cover wanted more hidden scratch than was reserved (hipErrorStreamCaptureUnsupported:
                # nothing of the capture has run): drop the capture, run the body eagerly -- which grows the scratch -- as THIS step, and
                # capture on the next call.  A genuine error raises again from the eager run.
                ctx.graph_abort()
                body()
                return
            except BaseException:
                ctx.graph_abort()
                raise
            self._graphs[key] = ctx.graph_end()
        ctx.graph_launch(self._graphs[key])

    # ------------------------------------------------------------------------------------ io
    def set_inputs(self, images=None, z=None, y_real=None, y_fake=None, y_gen=None, y_real_weights=None):
        ctx = self.ctx
        arrs = dict(images=images, z=z, y_real=y_real, y_fake=y_fake, y_gen=y_gen, y_real_weights=y_real_weights)
        if y_real is not None:
            arrs["lab_real"] = np.argmax(y_real, axis=1)
        if y_gen is not None:
            arrs["lab_gen"] = np.argmax(y_gen, axis=1)
        with torch.cuda.stream(ctx.stream):
            for k, a in arrs.items():
                if a is None:
                    continue
                dst = ctx.view(self.inp[k])
                src = torch.from_numpy(np.ascontiguousarray(np.asarray(a)))
                # pinned staging + asynchronous copy (a pageable copy waits for all queued device work)
                dst.copy_(src.reshape(dst.shape).to(dst.dtype).pin_memory(), non_blocking=True)
                # the halves of the critic step's one-pass inputs this array belongs to (device-to-device, same stream)
                B = self.B
                if k == "images":
                    ctx.view(self.inp["x_all"])[:B].copy_(dst, non_blocking=True)
                elif k == "y_real":
                    ctx.view(self.inp["y_all_fake"])[:B].copy_(dst, non_blocking=True)
                    ctx.view(self.inp["y_all_gen"])[:B].copy_(dst, non_blocking=True)
                elif k == "y_fake":
                    ctx.view(self.inp["y_all_fake"])[B:].copy_(dst, non_blocking=True)
                elif k == "y_gen":
                    ctx.view(self.inp["y_all_gen"])[B:].copy_(dst, non_blocking=True)

    def losses(self):
        out = {k: float(self.ctx.download(v)[0]) for k, v in self.loss.items()}
        if self.perm_mult:
            out["class_loss_fake"] /= self.perm_mult      # the accumulator holds the weighted term of the G objective
        return out

    def evaluate(self):
        """Forward pass only, on the current static inputs: the console metrics of the reference's loop
        (errD_real / errD_fake / errG / prob_real / prob_fake, model.py:374-399).  Like every ``sess.run`` of the
        reference it runs D and G in training mode (batch statistics, power-iteration update).  The losses are
        formed on the host from the downloaded logits -- logging only, nothing here feeds the optimisers."""
        ctx, g, inp = self.ctx, self.graph, self.inp
        ctx.new_step()
        g.begin_step(set())
        rec, ctx.recording = ctx.recording, False
        try:
            self._sn_prefetch()
            G = self.generator(inp["z"], inp["y_gen"])
            if self.alg in ("biased", "rcgan", "ambient"):
                lr_, wr = ctx.download(self.discriminator(inp["images"], inp["y_real"])).astype(np.float64), None
            else:
                lr_ = ctx.download(self.discriminator_all_labels(inp["images"])).astype(np.float64)
                wr = ctx.download(inp["y_real_weights"]).astype(np.float64)
            if self.alg in ("rcgan", "ambient") and self.est:
                lf = ctx.download(self.discriminator_all_labels(G)).astype(np.float64)
                C = ctx.download(self.confusion()).astype(np.float64).reshape(Y_DIM, Y_DIM)
                wf = ctx.download(inp["y_gen"]).astype(np.float64) @ C
            else:
                y = inp["y_fake"] if self.alg in ("rcgan", "ambient") else inp["y_gen"]
                lf, wf = ctx.download(self.discriminator(G, y)).astype(np.float64), None
        finally:
            ctx.recording = rec
        softplus = lambda x: np.logaddexp(0.0, x)
        if self.loss_fn == "hinge":
            f_real, f_fake, f_g = (lambda x: np.maximum(1 - x, 0)), (lambda x: np.maximum(1 + x, 0)), (lambda x: -x)
        else:
            f_real, f_fake, f_g = (lambda x: softplus(-x)), softplus, (lambda x: softplus(-x))
        sig = lambda x: 1.0 / (1.0 + np.exp(-x))
        red = lambda a, w: a.reshape(len(a), -1).mean(1) if w is None else (a * w).sum(1)
        return dict(d_loss_real=float(red(f_real(lr_), wr).mean()), d_loss_fake=float(red(f_fake(lf), wf).mean()),
                    g_loss=float(red(f_g(lf), wf).mean()), prob_real=red(sig(lr_), wr), prob_fake=red(sig(lf), wf))

    # ------------------------------------------------------------------------------------ recover_labels
    def recover_labels(self, images, y_actual, epochs=1000, learning_rate=5.e+2, seed=0, log_every=100, log=print):
        """DCGAN.recover_labels (mnist/model.py:494-640): for each of the R given real images, gradient descent
        (plain SGD, lr 500) on one latent z per (image, label) and on R x 10 label logits through the FROZEN sampler
        (inference-mode batch norm), minimising  mean_r sum_y softmax(logits)[r,y] * mse(image_r, G(z[r,y], onehot y)).
        Needs batch_size == R * 10 (the sampler batch, model.py:523).  Returns dict(y_recover, z_recover, mse_loss,
        zero_one_loss, history); variables start from TF's default glorot-uniform initialiser (RandomState(seed))."""
        ctx, g = self.ctx, self.graph
        images = np.asarray(images, np.float32).reshape(len(images), -1)
        y_actual = np.asarray(y_actual, np.float64)
        R = len(images)
        if self.B != R * Y_DIM:
            raise ValueError("recover_labels on %d images needs an engine built with batch_size=%d (got %d)" % (R, R * Y_DIM, self.B))
        rs = np.random.RandomState(seed)
        glorot = lambda shape: rs.uniform(-np.sqrt(6.0 / sum(shape)), np.sqrt(6.0 / sum(shape)), size=shape).astype(np.float32)
        P = ctx.persistent
        z = P((R * Y_DIM, Z_DIM), L.F32)
        logits = P((R, Y_DIM), L.F32)
        ctx.view(logits).copy_(torch.from_numpy(glorot((R, Y_DIM))))        # creation order: y_logit_recover, then z_recover
        ctx.view(z).copy_(torch.from_numpy(glorot((R * Y_DIM, Z_DIM))))
        hard_y = P((R * Y_DIM, Y_DIM), L.F32)
        ctx.view(hard_y).copy_(torch.from_numpy(np.tile(np.eye(Y_DIM, dtype=np.float32), (R, 1))))
        actual = P((R, images.shape[1]), ctx.act_dtype)
        ctx.view(actual).copy_(torch.from_numpy(images).to(ctx.view(actual).dtype))
        loss = P((1,), L.F32, fill=0.0)
        history = []
        t0 = time.time()
        res = {}
        for epoch in range(epochs):
            ctx.new_step()
            g.begin_step(set())
            z.grad = logits.grad = None
            z.req = logits.req = True
            yrec = O.softmax_rows(ctx, logits)
            G = self.generator(O.cast(ctx, z, ctx.act_dtype), hard_y, train=False)
            O.recover_mse(ctx, O.reshape(ctx, G, (R * Y_DIM, -1)), actual, yrec, loss)
            ctx.backward()
            # tf.train.GradientDescentOptimizer(lr).minimize(mse, var_list=[z_recover, y_logit_recover]); metrics of THIS run
            if (epoch + 1) % log_every == 0 or epoch == epochs - 1:
                yr = ctx.download(yrec).astype(np.float64)
                onehot = np.eye(Y_DIM)[np.argmax(yr, 1)]
                res = dict(mse_loss=float(ctx.download(loss)[0]), zero_one_loss=float(np.mean(1.0 - (y_actual * onehot).sum(1))))
                history.append((epoch, res["mse_loss"], res["zero_one_loss"]))
                log("Recover Epoch: [%2d] time: %4.2f, mse_loss: %.5g, zeroone_loss: %.5g" % (epoch, time.time() - t0, res["mse_loss"], res["zero_one_loss"]))
            for var in (z, logits):
                ctx.check(ctx.lib.rcgan_axpby(ctx.h, var.size, L.F32, -float(learning_rate), var.grad.ptr, 1.0, var.ptr))
        res.update(y_recover=ctx.download(O.softmax_rows(ctx, logits)), z_recover=ctx.download(z), history=history)
        return res

    # ------------------------------------------------------------------------------------ checkpoints
    def state_dict(self):
        """TF variable names -> arrays: parameters, Adam slots ("<var>/Adam", "<var>/Adam_1"), optimiser steps,
        BN moving statistics and power-iteration vectors (what tf.train.Saver stores for DCGAN, model.py:845-854)."""
        sd = {}
        for gname, grp in zip(("generator", "discriminator", "confusion"), self.groups):
            for n in grp.names:
                sd[n] = grp.get(n)
                sd[n + "/Adam"] = grp.get(n, "m")
                sd[n + "/Adam_1"] = grp.get(n, "v")
            sd["_opt/%s/step" % gname] = np.array([grp.t], np.int64)
        sd.update(self.get_state())
        # AdamOptimizer's own step state, in the reference's creation order: d_optim, g_optim, c_optim (model.py:250-262)
        from .host import adam_power_tensors
        opts = [(self.PD.t, self.beta1, 0.999), (self.PG.t, self.beta1, 0.999)]
        if getattr(self, "PC", None) is not None:
            opts.append((self.PC.t, self.beta1, 0.999))
        sd.update(adam_power_tensors(opts))
        return sd

    def load_state_dict(self, sd):
        for gname, grp in zip(("generator", "discriminator", "confusion"), self.groups):
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
                sfx = {"discriminator": "", "generator": "_1", "confusion": "_2"}[gname]
                if "beta2_power" + sfx in sd:
                    grp.t = steps_from_beta_power(float(np.asarray(sd["beta2_power" + sfx])), 0.999)
        ctx = self.ctx
        with torch.cuda.stream(ctx.stream):
            for k, t in self.state.items():
                if k in sd:
                    ctx.view(t).copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(sd[k], np.float32).reshape(-1))))
        ctx.sync()

    def get_params(self):
        out = {}
        for grp in self.groups:
            for n in grp.names:
                out[n] = grp.get(n)
        return out

    def get_grads(self, group):
        return {n: group.get(n, "grad") for n in group.names}

    def get_state(self):
        return {k: self.ctx.download(v).reshape(self.state_shapes[k]) for k, v in self.state.items()}

    def sampler(self, z, y):
        """gen_sampler (model.py:733-757): same weights, BN in inference mode."""
        ctx, g = self.ctx, self.graph
        ctx.new_step()
        g.begin_step(set())
        rec, ctx.recording = ctx.recording, False
        try:
            out = self.generator(ctx.upload(np.asarray(z, np.float32)), ctx.upload(np.asarray(y, np.float32), L.F32), train=False)
            res = ctx.download(out)
        finally:
            ctx.recording = rec
        return res
