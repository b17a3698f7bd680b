"""Drop-in mirror of the reference's MNIST L1 op API (mnist/ops.py:30-116, mnist/sn.py): same names and
arguments, executed eagerly on device tensors by the gfx950 kernels.  ``_act`` on batch_norm is this
build's fusion hook (the reference applies relu / lrelu as a separate op right after)."""
from . import _lib as L
from . import ops as O
from .variables import Graph, scoped, variable_scope


def _graph():
    g = Graph.current
    if g is None:
        raise RuntimeError("no active Graph: call Graph.begin_step() first")
    return g


class batch_norm(object):
    """mnist/ops.py:30-44: tf.contrib.layers.batch_norm(decay=momentum, epsilon, scale=True,
    updates_collections=None, is_training=train)."""

    def __init__(self, epsilon=1e-5, momentum=0.9, name="batch_norm"):
        self.epsilon, self.momentum, self.name = epsilon, momentum, name

    def __call__(self, x, train=True, _act=L.ACT_NONE, _segments=1):
        """_segments > 1: x holds that many batches back to back (consecutive calls of the reference on each of them, in that order)."""
        g = _graph()
        with variable_scope(self.name):
            gamma, beta = g.param(scoped("gamma")), g.param(scoped("beta"))
            mm, mv = g.state[scoped("moving_mean")], g.state[scoped("moving_variance")]
        if train:
            return O.batch_norm_act(g.ctx, x, gamma, beta, act=_act, moving=(mm, mv), decay=self.momentum, eps=self.epsilon, segments=_segments)
        assert _segments == 1
        return O.batch_norm_infer(g.ctx, x, gamma, beta, mm, mv, act=_act, eps=self.epsilon)


def conv_cond_concat(x, y):
    """mnist/ops.py:46-51.  y: fp32 [n, y_dim] one-hot rows (the reference passes it reshaped to [n,1,1,y_dim])."""
    return O.concat_channels(_graph().ctx, x, y)


def conv2d(input_, output_dim, k_h=5, k_w=5, d_h=2, d_w=2, stddev=0.02, spectral_norm=False, name="conv2d"):
    """mnist/ops.py:53-67."""
    if k_h != k_w or d_h != d_w:
        raise NotImplementedError("only square kernels / strides are used by the reference")
    g = _graph()
    with variable_scope(name):
        wname = scoped('w')
        if spectral_norm:
            with variable_scope('spectral_norm'):
                w = g.sn_weight(wname, scoped('u'), True)          # update_collection=None: u updated every execution
        else:
            w = g.weight(wname)
        b = g.param(scoped('biases'))
    return O.conv2d(g.ctx, input_, w, b, k_h, d_h)


def deconv2d(input_, output_shape, k_h=5, k_w=5, d_h=2, d_w=2, stddev=0.02, name="deconv2d", with_w=False):
    """mnist/ops.py:69-92."""
    g = _graph()
    with variable_scope(name):
        w, b = g.param(scoped('w')), g.param(scoped('biases'))
    out = O.deconv2d(g.ctx, input_, w, b, tuple(output_shape), k_h, d_h)
    return (out, w, b) if with_w else out


def lrelu(x, leak=0.2, name="lrelu"):
    """mnist/ops.py:94-95."""
    if leak != 0.2:
        raise NotImplementedError("leak != 0.2 is never used by the reference")
    return O.act(_graph().ctx, x, L.ACT_LRELU)


def linear(input_, output_size, scope=None, stddev=0.02, bias_start=0.0, with_w=False, max_norm=False):
    """mnist/ops.py:97-116.  max_norm is a variable *constraint* (clip to [-1,1] after each optimiser update);
    it is applied by the optimiser step of the trainer, exactly where TF applies it."""
    g = _graph()
    with variable_scope(scope or "Linear"):
        w = g.weight(scoped("Matrix"))
        b = g.param(scoped("bias"))
    out = O.linear(g.ctx, input_, w, b)
    return (out, w.param, b) if with_w else out
