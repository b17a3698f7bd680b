"""Generated-label accuracy (gan_resnet.py:424-455, 847-861, 995-1005): how often a frozen CIFAR-10 classifier agrees
with the label the generator was conditioned on.

The reference imports ``resnet-110/graph_optimized.pb`` into a TensorFlow session.  Here the same network -- decoded from
that file without TensorFlow by scripts/extract_label_classifier.py into assets/cifar_label_classifier.npz -- runs on the
engine's own kernels in fp32: a pre-activation ResNet-32 (conv0 3->16, three stages of five basic blocks at 16/32/64
channels, stride-2 first conv in stages 2 and 3 with the option-A shortcut = 2x2 average pool + zero channel padding),
every batch norm on the moments of the evaluated batch itself (eps 1e-3, no moving statistics), ReLU, global average
pool, 64->10 dense layer, softmax.  Inputs are the raw integer pixels 0..255 in NHWC, all 1000 samples in ONE batch
(the batch statistics depend on it), exactly as ``generated_label_accuracy`` feeds them.
"""
import os

import numpy as np

from . import _lib as L
from . import ops as O
from .runtime import Context

ASSET = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "cifar_label_classifier.npz")
BN_EPS = 1e-3
STAGES, BLOCKS = 3, 5


class LabelClassifier:
    def __init__(self, device=0, arena_bytes=5 << 30, asset=ASSET):
        if not os.path.exists(asset):
            raise RuntimeError("label-classifier weights missing: %s (scripts/extract_label_classifier.py writes them)" % asset)
        self.ctx = ctx = Context(device, "f32", arena_bytes=arena_bytes, ws_bytes=1 << 28)
        z = np.load(asset)
        self.w = {}
        for k in z.files:
            a = z[k]
            if a.dtype.kind != "f" or a.ndim == 0:
                continue                                   # reduction indices, paddings, eps scalars
            t = ctx.persistent(a.shape, L.F32)
            ctx.view(t).copy_(__import__("torch").from_numpy(np.ascontiguousarray(a, np.float32)))
            t.name = k.replace("|", "/")
            self.w[t.name] = t
        ctx.sync()

    def _conv(self, x, name, stride=1):
        return O.conv2d(self.ctx, x, O.Weight(self.ctx, self.w[name + "/conv"]), None, 3, stride=stride)

    def _bn_relu(self, x, name):
        return O.batch_norm_act(self.ctx, x, self.w[name + "/gamma"], self.w[name + "/beta"], act=L.ACT_RELU, eps=BN_EPS)

    def softmax(self, images):
        """images: [n,32,32,3] raw pixel values 0..255 (any numeric dtype).  -> softmax [n,10] float32."""
        ctx = self.ctx
        x = np.ascontiguousarray(np.asarray(images, np.float32))
        if x.ndim != 4 or x.shape[1:] != (32, 32, 3):
            raise ValueError("expected [n,32,32,3] images, got %s" % (x.shape,))
        ctx.new_step()
        rec, ctx.recording = ctx.recording, False
        try:
            h = self._bn_relu(self._conv(ctx.upload(x, L.F32), "conv0"), "conv0")
            for s in range(1, STAGES + 1):
                for b in range(BLOCKS):
                    p = "conv%d_%d" % (s, b)
                    down = b == 0 and s > 1
                    t = h if (s == 1 and b == 0) else self._bn_relu(h, p + "/conv1_in_block")
                    c1 = self._conv(t, p + "/conv1_in_block", stride=2 if down else 1)
                    c2 = self._conv(self._bn_relu(c1, p + "/conv2_in_block"), p + "/conv2_in_block")
                    sc = h
                    if down:                               # option-A shortcut: AvgPool 2x2 + Pad channels (C/2 each side)
                        pooled = O.meanpool2(ctx, h)
                        n, hh, ww, c = pooled.shape
                        sc = ctx.empty((n, hh, ww, 2 * c), L.F32)
                        ctx.check(ctx.lib.rcgan_pad_channels(ctx.h, n * hh * ww, c, c // 2, c // 2, L.F32, pooled.ptr, sc.ptr))
                    h = O.add(ctx, c2, sc)
            feat = O.act_meanhw(ctx, self._bn_relu(h, "fc"), L.ACT_NONE)
            logits = O.linear(ctx, feat, O.Weight(ctx, self.w["fc/fc_weights"]), self.w["fc/fc_bias"])
            return ctx.download(O.softmax_rows(ctx, logits))
        finally:
            ctx.recording = rec

    def close(self):
        self.ctx.close()


class TemplateClassifier:
    """Stand-in label classifier for the "templates" synthetic images (data.synthetic_cifar(kind="templates")): the class of an
    image is the nearest of the ten class patterns in the pre-squash domain, atanh(pixel) vs 0.6 * template -- the maximum-
    likelihood rule of that generative model up to the noise covariance; 99.9 % correct on the synthetic real images
    (tests/test_host_cpu.py).  Host numpy: evaluation of a synthetic stand-in, not part of the training step.  Same interface as
    ``LabelClassifier`` (``softmax`` returns a one-hot row per image) so ``generated_label_accuracy`` takes either."""

    def __init__(self, device=0):
        from . import data as D
        self.t = (0.6 * D.class_templates()).transpose(0, 2, 3, 1).reshape(10, -1).astype(np.float32)     # HWC rows

    def softmax(self, images):
        x = np.asarray(images, np.float32)
        if x.ndim != 4 or x.shape[1:] != (32, 32, 3):
            raise ValueError("expected [n,32,32,3] images, got %s" % (x.shape,))
        a = np.arctanh(np.clip((x + 0.5) / 128.0 - 1.0, -0.999, 0.999)).reshape(len(x), -1)
        d = (a * a).sum(1, keepdims=True) - 2.0 * a.dot(self.t.T) + (self.t * self.t).sum(1)[None]
        out = np.zeros((len(x), 10), np.float32)
        out[np.arange(len(x)), d.argmin(1)] = 1.0
        return out

    def close(self):
        pass


def generated_label_accuracy(samples, labels, confusion_matrix=None, classifier=None, device=0):
    """gan_resnet.py:424-455.  samples int [n,32,32,3] in 0..255; labels int [n]; confusion_matrix (rcgan-u): labels are
    first mapped through the arg-max permutation of the learned matrix."""
    labels = np.asarray(labels)
    if confusion_matrix is not None:
        cm = np.asarray(confusion_matrix)
        perm = np.zeros_like(cm, dtype=int)
        perm[np.arange(cm.shape[0]), np.argmax(cm, axis=-1)] = 1
        onehot = np.zeros([labels.shape[0], cm.shape[0]], dtype=float)
        onehot[np.arange(labels.shape[0]), labels] = 1
        labels = np.argmax(onehot.dot(perm), axis=-1)
    own = classifier is None
    clf = LabelClassifier(device) if own else classifier
    try:
        softmax = clf.softmax(samples)
    finally:
        if own:
            clf.close()
    return float((labels == np.argmax(softmax, axis=-1)).astype(float).mean())
