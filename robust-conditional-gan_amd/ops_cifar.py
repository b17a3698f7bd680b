"""Drop-in mirror of the reference's CIFAR L1 op API (cifar10/common/ops/{conv2d,linear,normalization,
embedding,sn}.py): same function names, argument names and error behaviour, but each call runs the
gfx950 kernels eagerly on device tensors instead of adding TF graph nodes.

Only the code paths the reference actually exercises are implemented (``conv_type='conv2d'``, no
weightnorm, no mask, no ``inputs_norm`` -- SURVEY.md 2.1 #7); the others raise NotImplementedError.
"""
from . import _lib as L
from . import ops as O
from .variables import Graph, scoped, variable_scope

NO_OPS = 'NO_OPS'     # cifar10/common/ops/sn.py:10


def _graph():
    g = Graph.current
    if g is None:
        raise RuntimeError("no active Graph: call Graph.begin_step() first")
    return g


def spectral_normed_weight(W, u=None, num_iters=1, update_collection=None, with_sigma=False, reuse=False):
    """cifar10/common/ops/sn.py:17-75.  ``W``: the weight tensor as in the reference (a parameter handle from the variable
    store, Graph.param) -- or, as a convenience, its full variable name.  update_collection None -> u is updated by this
    execution; NO_OPS -> u untouched.  Returns the Weight handle (W_bar = W / sigma, never materialised: the division is fused
    into the consumers' operand load)."""
    if num_iters != 1:
        raise NotImplementedError("num_iters != 1 is never used by the reference")
    if update_collection not in (None, NO_OPS):
        raise NotImplementedError("update collections other than None / NO_OPS are never used by the reference")
    g = _graph()
    with variable_scope('spectral_norm'):
        uname = u if u is not None else scoped('u')
    W_name = W if isinstance(W, str) else W.name
    if not isinstance(W_name, str) or not g.has(W_name):
        raise ValueError("spectral_normed_weight needs a variable of the store (got %r)" % (W,))
    w = g.sn_weight(W_name, uname, update_collection is None)
    if with_sigma:
        return w, w.sigma
    return w


def Conv2D(inputs, input_dim, output_dim, filter_size=3, stride=1, name='Conv2D',
           conv_type='conv2d', channel_multiplier=0, padding='SAME',
           spectral_normed=False, update_collection=None, inputs_norm=False, he_init=True,
           mask_type=None, weightnorm=None, biases=True, gain=1.,
           _in_upsample=False, _in_relu=False, _accumulate_into=None, _residual=None, _out_meanpool=False, _residual_up=False, _bn_next=False):
    """cifar10/common/ops/conv2d.py:31-218 (conv2d path).  The underscore arguments are this build's
    fusion hooks (nearest-2x upsample / ReLU folded into the operand load, residual accumulate / residual add)."""
    if conv_type != 'conv2d':
        raise NotImplementedError('{0} is not supported!'.format(conv_type))
    if mask_type is not None or weightnorm or inputs_norm or padding != 'SAME' or gain != 1.:
        raise NotImplementedError('mask/weightnorm/inputs_norm/gain are never used by the reference')
    g = _graph()
    with variable_scope(name):
        fname = scoped('Filters')
        if spectral_normed:
            with variable_scope('filters'):
                w = spectral_normed_weight(g.param(fname), update_collection=update_collection)      # conv2d.py:169-171
        else:
            w = g.weight(fname)
        b = g.param(scoped('Biases')) if biases else None
    if inputs.shape[-1] != input_dim:
        raise ValueError("input_dim %d does not match inputs %s" % (input_dim, inputs.shape))
    if _out_meanpool:        # ConvMeanPool with the pool folded into the convolution (O.conv2d_meanpool)
        assert filter_size == 3 and stride == 1 and not _in_upsample and _residual is None
        return O.conv2d_meanpool(g.ctx, inputs, w, b, in_relu=_in_relu, accumulate_into=_accumulate_into)
    return O.conv2d(g.ctx, inputs, w, b, filter_size, stride, in_up=_in_upsample, in_relu=_in_relu,
                    accumulate_into=_accumulate_into, residual=_residual, residual_up=_residual_up, want_stats=_bn_next)


def Linear(inputs, input_dim, output_dim, name,
           spectral_normed=False, update_collection=None, reuse=False, inputs_norm=False,
           biases=True, initialization=None, weightnorm=None, gain=1.):
    """cifar10/common/ops/linear.py:38-182."""
    if weightnorm or inputs_norm or gain != 1.:
        raise NotImplementedError('weightnorm/inputs_norm/gain are never used by the reference')
    g = _graph()
    with variable_scope(name):
        wname = scoped('W')
        if spectral_normed:
            w = spectral_normed_weight(g.param(wname), update_collection=update_collection)          # linear.py:163-168
        else:
            w = g.weight(wname)
        b = g.param(scoped('b')) if biases else None
    x = inputs
    if len(x.shape) != 2:
        x = O.reshape(g.ctx, x, (-1, input_dim))
    return O.linear(g.ctx, x, w, b)


def cond_batchnorm(name, axes, inputs, is_training=None, stats_iter=None, update_moving_stats=True, fused=True,
                   labels=None, n_labels=None, _act=L.ACT_NONE, _segments=1, _defer_apply=False):
    """cifar10/common/ops/normalization.py:27-59: batch moments over (N,H,W), per-class scale/offset
    gathered by label, no moving averages.  ``_act`` fuses the following nonlinearity."""
    if axes != [0, 1, 2]:
        raise Exception('Axes is not supported in Conditional BatchNorm!')
    g = _graph()
    with variable_scope('CondBatchNorm'):
        offset_m = g.param(scoped('offset'))
        scale_m = g.param(scoped('scale'))
    return O.batch_norm_act(g.ctx, inputs, scale_m, offset_m, act=_act, labels=labels, n_labels=n_labels, segments=_segments,
                            defer_apply=_defer_apply)


def embed_y(inputs, vocab_size=1000, embedding_dim=300, word2vec_file=None,
            spectral_normed=False, update_collection=None, reuse=False):
    """cifar10/common/ops/embedding.py:12-51: trainable table lookup.  inputs: int32 device labels [n]."""
    if word2vec_file is not None:
        raise NotImplementedError("word2vec embeddings are never used by the reference (WORD2VEC_FILE = None)")
    g = _graph()
    with variable_scope("Embedding.Label"):
        table = g.param(scoped('embedding_map'))
    if table.shape != (vocab_size, embedding_dim):
        raise ValueError("embedding_map shape %s != (%d, %d)" % (table.shape, vocab_size, embedding_dim))
    return O.gather_rows(g.ctx, table, inputs, inputs.shape[0])
