"""numpy forward/backward primitives with TensorFlow-1.5 semantics (oracle; test infrastructure).

Layouts: activations NHWC, conv filters HWIO, transposed-conv filters HW-O-I
(reference ``mnist/ops.py:74``).  Every function keeps the dtype of its inputs
so the same code serves as an fp32 oracle and as an fp64 gradient checker.
Citations are relative to /root/reference.
"""
import numpy as np


# ----------------------------------------------------------------------------
# SAME padding  (tf.nn.conv2d padding='SAME'; mnist/ops.py:62, cifar10/common/ops/conv2d.py:181-187)
# ----------------------------------------------------------------------------
def same_pad(in_size, k, s):
    """TF rule: out=ceil(in/s); total=max((out-1)s+k-in,0); before=total//2 (extra goes after)."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2, total - total // 2


def _pad_input(x, kh, kw, s):
    n, h, w, c = x.shape
    oh, pt, pb = same_pad(h, kh, s)
    ow, pl, pr = same_pad(w, kw, s)
    xp = np.zeros((n, h + pt + pb, w + pl + pr, c), dtype=x.dtype)
    xp[:, pt:pt + h, pl:pl + w, :] = x
    return xp, oh, ow, pt, pl


def conv2d_fwd(x, w, stride=1):
    """y[n,oh,ow,co] = sum x[n,oh*s+kh-pt,ow*s+kw-pl,ci] w[kh,kw,ci,co]  (tf.nn.conv2d NHWC SAME)."""
    kh, kw, ci, co = w.shape
    xp, oh, ow, _, _ = _pad_input(x, kh, kw, stride)
    n = x.shape[0]
    y = np.zeros((n * oh * ow, co), dtype=x.dtype)
    for a in range(kh):
        for b in range(kw):
            xs = xp[:, a:a + (oh - 1) * stride + 1:stride, b:b + (ow - 1) * stride + 1:stride, :]
            y += xs.reshape(-1, ci) @ w[a, b]
    return y.reshape(n, oh, ow, co)


def conv2d_bwd_input(dy, w, x_shape, stride=1):
    """Gradient of conv2d_fwd w.r.t. x (== tf.nn.conv2d_backprop_input)."""
    kh, kw, ci, co = w.shape
    n, h, wd, _ = x_shape
    oh, pt, pb = same_pad(h, kh, stride)
    ow, pl, pr = same_pad(wd, kw, stride)
    assert dy.shape == (n, oh, ow, co), (dy.shape, (n, oh, ow, co))
    dxp = np.zeros((n, h + pt + pb, wd + pl + pr, ci), dtype=dy.dtype)
    dy2 = dy.reshape(-1, co)
    for a in range(kh):
        for b in range(kw):
            g = (dy2 @ w[a, b].T).reshape(n, oh, ow, ci)
            dxp[:, a:a + (oh - 1) * stride + 1:stride, b:b + (ow - 1) * stride + 1:stride, :] += g
    return dxp[:, pt:pt + h, pl:pl + wd, :]


def conv2d_bwd_filter(x, dy, w_shape, stride=1):
    """Gradient of conv2d_fwd w.r.t. w (== tf.nn.conv2d_backprop_filter)."""
    kh, kw, ci, co = w_shape
    xp, oh, ow, _, _ = _pad_input(x, kh, kw, stride)
    dy2 = dy.reshape(-1, co)
    dw = np.zeros(w_shape, dtype=x.dtype)
    for a in range(kh):
        for b in range(kw):
            xs = xp[:, a:a + (oh - 1) * stride + 1:stride, b:b + (ow - 1) * stride + 1:stride, :]
            dw[a, b] = xs.reshape(-1, ci).T @ dy2
    return dw


# tf.nn.conv2d_transpose(value, filter[kh,kw,Cout,Cin], output_shape, strides) == conv2d_backprop_input
# (mnist/ops.py:78-79).  Its "forward conv" has input channels = Cout(of the deconv) and output channels = Cin.
def conv2d_transpose_fwd(x, w, out_shape, stride=2):
    return conv2d_bwd_input(x, w, out_shape, stride)


def conv2d_transpose_bwd_input(dy, w, stride=2):
    return conv2d_fwd(dy, w, stride)


def conv2d_transpose_bwd_filter(x, dy, w_shape, stride=2):
    # forward-conv input is dy's tensor (the deconv output), forward-conv output grad is x
    return conv2d_bwd_filter(dy, x, w_shape, stride)


# ----------------------------------------------------------------------------
# resampling used by the CIFAR ResNet blocks (cifar10/gan_resnet.py:231-272)
# ----------------------------------------------------------------------------
def meanpool2(x):
    """add_n of the four strided slices / 4  (gan_resnet.py:239-240, 248-249)."""
    return (x[:, ::2, ::2, :] + x[:, 1::2, ::2, :] + x[:, ::2, 1::2, :] + x[:, 1::2, 1::2, :]) / 4.


def meanpool2_bwd(dy):
    n, h, w, c = dy.shape
    dx = np.empty((n, 2 * h, 2 * w, c), dtype=dy.dtype)
    q = dy / 4.
    dx[:, ::2, ::2, :] = q
    dx[:, 1::2, ::2, :] = q
    dx[:, ::2, 1::2, :] = q
    dx[:, 1::2, 1::2, :] = q
    return dx


def upsample2(x):
    """concat([x]*4, axis=3) + depth_to_space(2) == nearest-neighbour 2x (gan_resnet.py:263-264)."""
    return x.repeat(2, axis=1).repeat(2, axis=2)


def upsample2_bwd(dy):
    return dy[:, ::2, ::2, :] + dy[:, 1::2, ::2, :] + dy[:, ::2, 1::2, :] + dy[:, 1::2, 1::2, :]


# ----------------------------------------------------------------------------
# normalisation
# ----------------------------------------------------------------------------
def cond_batchnorm_fwd(x, labels, scale_m, offset_m, eps=1e-5):
    """cifar10/common/ops/normalization.py:47-57: tf.nn.moments over (N,H,W) (biased variance),
    per-sample gamma/beta rows gathered by label, tf.nn.batch_normalization:
    inv = rsqrt(var+eps)*scale ; y = x*inv + (offset - mean*inv)."""
    mean = x.mean(axis=(0, 1, 2))
    var = ((x - mean) ** 2).mean(axis=(0, 1, 2))
    rstd = 1.0 / np.sqrt(var + x.dtype.type(eps))
    g = scale_m[labels][:, None, None, :]
    b = offset_m[labels][:, None, None, :]
    inv = rstd * g
    y = x * inv + (b - mean * inv)
    return y, (mean, rstd)


def cond_batchnorm_bwd(dy, x, labels, scale_m, stats):
    mean, rstd = stats
    xhat = (x - mean) * rstd
    g = scale_m[labels][:, None, None, :]
    dscale = np.zeros_like(scale_m)
    doffset = np.zeros_like(scale_m)
    np.add.at(doffset, labels, dy.sum(axis=(1, 2)))
    np.add.at(dscale, labels, (dy * xhat).sum(axis=(1, 2)))
    dxhat = dy * g
    m1 = dxhat.mean(axis=(0, 1, 2))
    m2 = (dxhat * xhat).mean(axis=(0, 1, 2))
    dx = rstd * (dxhat - m1 - xhat * m2)
    return dx, dscale, doffset


def batch_norm_train_fwd(x, gamma, beta, moving_mean, moving_var, decay=0.9, eps=1e-5):
    """tf.contrib.layers.batch_norm(decay, epsilon, scale=True, updates_collections=None, is_training=True)
    (mnist/ops.py:38-44).  TF 1.5 routes rank-2/rank-4 inputs to the fused kernel: normalise with the
    biased batch variance; moving_var is updated with the UNBIASED (N/(N-1)) batch variance;
    moving <- moving - (moving - batch)*(1-decay)   (confidence medium)."""
    axes = tuple(range(x.ndim - 1))
    cnt = int(np.prod([x.shape[a] for a in axes]))
    mean = x.mean(axis=axes)
    var = ((x - mean) ** 2).mean(axis=axes)
    rstd = 1.0 / np.sqrt(var + x.dtype.type(eps))
    y = (x - mean) * rstd * gamma + beta
    one_m = x.dtype.type(1.0 - decay)
    uvar = var * x.dtype.type(cnt / max(cnt - 1, 1))
    new_mm = moving_mean - (moving_mean - mean) * one_m
    new_mv = moving_var - (moving_var - uvar) * one_m
    return y, (mean, rstd), new_mm, new_mv


def batch_norm_train_bwd(dy, x, gamma, stats):
    mean, rstd = stats
    axes = tuple(range(x.ndim - 1))
    xhat = (x - mean) * rstd
    dgamma = (dy * xhat).sum(axis=axes)
    dbeta = dy.sum(axis=axes)
    dxhat = dy * gamma
    m1 = dxhat.mean(axis=axes)
    m2 = (dxhat * xhat).mean(axis=axes)
    dx = rstd * (dxhat - m1 - xhat * m2)
    return dx, dgamma, dbeta


def batch_norm_infer(x, gamma, beta, moving_mean, moving_var, eps=1e-5):
    """is_training=False path used by gen_sampler (mnist/model.py:745-754)."""
    return (x - moving_mean) / np.sqrt(moving_var + x.dtype.type(eps)) * gamma + beta


# ----------------------------------------------------------------------------
# spectral normalisation (mnist/sn.py:13-75 == cifar10/common/ops/sn.py:13-75)
# ----------------------------------------------------------------------------
SN_EPS = 1e-12


def _l2n(v, eps=SN_EPS):
    return v / (np.sqrt((v ** 2).sum()) + v.dtype.type(eps))


def spectral_norm_fwd(w, u):
    """One power iteration (num_iters=1): v=l2n(u W^T); u'=l2n(v W); sigma=v W u'^T; W_bar=W/sigma.
    w: any shape with last dim C; u: [1,C].  Returns W_bar, sigma, u', cache."""
    shape = w.shape
    wr = w.reshape(-1, shape[-1])
    a = u @ wr.T                      # [1,K]
    v = _l2n(a)
    b = v @ wr                        # [1,C]
    u2 = _l2n(b)
    sigma = (b @ u2.T)[0, 0]          # == (v W) u'^T
    wbar = (wr / sigma).reshape(shape)
    return wbar, sigma, u2, (a, v, b, u2, sigma)


def spectral_norm_bwd(dwbar, w, u, cache):
    """Gradient of W_bar w.r.t. W *through* the power iteration: the reference puts no
    stop_gradient on v / u' (sn.py:37-60), so TF differentiates the while_loop body too."""
    a, v, b, u2, sigma = cache
    shape = w.shape
    wr = w.reshape(-1, shape[-1])
    g = dwbar.reshape(wr.shape)
    eps = w.dtype.type(SN_EPS)
    dsigma = -(g * wr).sum() / (sigma * sigma)
    # sigma = b . u2,  u2 = b/(|b|+eps)
    nb = np.sqrt((b ** 2).sum())
    du2 = dsigma * b                                   # [1,C]
    db = dsigma * u2 + du2 / (nb + eps) - b * ((du2 * b).sum() / (nb * (nb + eps) ** 2))
    # b = v W
    dw = g / sigma + v.T @ db
    dv = db @ wr.T                                     # [1,K]
    na = np.sqrt((a ** 2).sum())
    da = dv / (na + eps) - a * ((dv * a).sum() / (na * (na + eps) ** 2))
    # a = u W^T  -> dW[k,c] += da[k] u[c]
    dw = dw + da.T @ u
    return dw.reshape(shape)


# ----------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------
def sigmoid_ce_logits(x, z):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log1p(exp(-|x|))."""
    return np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))


def sigmoid_ce_logits_bwd(x, z):
    return 1.0 / (1.0 + np.exp(-x)) - z


def softmax_rows(l):
    e = np.exp(l - l.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def softmax_rows_bwd(dp, p):
    return p * (dp - (dp * p).sum(axis=-1, keepdims=True))


# ----------------------------------------------------------------------------
# optimiser: tf.train.AdamOptimizer (mnist/model.py:250-262, cifar10/gan_resnet.py:802-817)
# ----------------------------------------------------------------------------
def adam_tf(w, g, m, v, t, lr, beta1, beta2, eps=1e-8, clip=None):
    """lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; w -= lr_t*m/(sqrt(v)+eps)  (eps outside the
    bias correction).  ``clip``: variable constraint applied right after the update
    (mnist/ops.py:102-111, tf.clip_by_value(x,-1,1))."""
    dt = w.dtype.type
    # arithmetic form of TF's ApplyAdam kernel (tensorflow/core/kernels/training_ops.cc):
    #   alpha = lr*sqrt(1-beta2_power)/(1-beta1_power); m += (g-m)*(1-b1); v += (g^2-v)*(1-b2);
    #   var -= (m*alpha)/(sqrt(v)+eps); beta powers are variables multiplied by beta once per step.
    b1p, b2p = dt(1.0), dt(1.0)
    for _ in range(int(t)):
        b1p, b2p = dt(b1p * dt(beta1)), dt(b2p * dt(beta2))
    alpha = dt(lr) * np.sqrt(dt(1.0) - b2p) / (dt(1.0) - b1p)
    m = m + (g - m) * (dt(1.0) - dt(beta1))
    v = v + (g * g - v) * (dt(1.0) - dt(beta2))
    w = w - (m * alpha) / (np.sqrt(v) + dt(eps))
    if clip is not None:
        w = np.clip(w, -clip, clip)
    return w, m, v
