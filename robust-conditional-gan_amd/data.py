"""Host-side data + label-noise pipeline of the CIFAR engine (product code; numpy legacy RNG so the
label streams are bit-identical to the reference's for the same seed).

Mirrors /root/reference cifar10/common/data/cifar10.py:12-52 (unpickle, cifar_generator, load) and the
two infinite generators of cifar10/gan_resnet.py:864-885.
"""
import os
import pickle

import numpy as np


def C_ALPHA(alpha):
    """One-coin confusion matrix (gan_resnet.py:106)."""
    return ((1 - alpha) / 9.0) * np.ones((10, 10)) + (alpha - (1 - alpha) / 9.0) * np.eye(10)


def unpickle(file):
    with open(file, 'rb') as fo:
        d = pickle.load(fo, encoding='bytes')
    return d[b'data'], d[b'labels']


def corrupt_labels(labels, C, rng=np.random):
    """cifar10.py:29-41: the noisy channel applied to the real labels, the uniformly random generator labels,
    their channel-corrupted version and the C^-1 rows for the unbiased loss.  Draw order is the reference's:
    randint(10, 50000) first, then per sample multinomial(C[label]) and multinomial(C[random])."""
    labels = np.array(labels)
    labels_random = rng.randint(10, size=50000)
    labels_biased = np.zeros((50000,))
    labels_inv_weights = np.zeros((50000, 10))
    C_inv = np.linalg.inv(C)
    for i in range(len(labels)):
        labels[i] = np.flatnonzero(rng.multinomial(1, C[labels[i], :]))[0]
        labels_inv_weights[i] = C_inv[labels[i], :]
        labels_biased[i] = np.flatnonzero(rng.multinomial(1, C[labels_random[i], :]))[0]
    return labels, labels_random, labels_biased, labels_inv_weights


def cifar_generator(images, labels, batch_size, C, rng=np.random):
    """cifar10.py:19-45 on in-memory arrays: fixed order, no shuffling, tail dropped."""
    labels, labels_random, labels_biased, labels_inv_weights = corrupt_labels(labels, C, rng)

    def get_epoch():
        for i in range(int(len(images) / batch_size)):
            s = slice(i * batch_size, (i + 1) * batch_size)
            yield (images[s], labels[s], labels_random[s], labels_biased[s], labels_inv_weights[s])
    return get_epoch


def load(batch_size, data_dir, C, rng=np.random):
    """cifar10.py:48-52."""
    def read(files):
        xs, ys = [], []
        for f in files:
            x, y = unpickle(os.path.join(data_dir, f))
            xs.append(x)
            ys.append(y)
        return np.concatenate(xs, axis=0), np.concatenate(ys, axis=0)
    tx, ty = read(['data_batch_%d' % i for i in range(1, 6)])
    vx, vy = read(['test_batch'])
    return cifar_generator(tx, ty, batch_size, C, rng), cifar_generator(vx, vy, batch_size, C, rng)


def class_templates():
    """Ten fixed low-frequency colour patterns [10,3,32,32] (sums of three separable cosines per channel, frequencies 0..2):
    the class-carrying part of the "templates" synthetic images.  Not reference data -- a stand-in for CIFAR-10 (no dataset and
    no network here) on which a generated image's class can be read off exactly (eval_cifar.TemplateClassifier)."""
    trs = np.random.RandomState(4242)
    yy, xx = np.mgrid[0:32, 0:32] / 32.0
    tmpl = np.zeros((10, 3, 32, 32))
    for c in range(10):
        for ch in range(3):
            for _ in range(3):
                fx, fy = trs.randint(0, 3, size=2)
                px, py = trs.uniform(0, 2 * np.pi, size=2)
                tmpl[c, ch] += trs.uniform(0.3, 1.0) * np.cos(2 * np.pi * fx * xx + px) * np.cos(2 * np.pi * fy * yy + py)
    return tmpl


def template_images(rs, labels):
    """uint8-valued CHW rows [n,3072]: tanh(0.6 template[label] + 0.6 low-pass noise) mapped to 0..255 -- natural-image-like
    second-order statistics that carry the label (per-image noise sigma ~ 1.4 pixels, as strong as the class pattern)."""
    n = len(labels)
    noise = rs.randn(n, 3, 32, 32)
    for _ in range(4):          # separable [1 2 1]/4 blur, wrap-around
        noise = (np.roll(noise, 1, 2) + 2 * noise + np.roll(noise, -1, 2)) / 4
        noise = (np.roll(noise, 1, 3) + 2 * noise + np.roll(noise, -1, 3)) / 4
    noise /= noise.std()
    img = np.tanh(0.6 * class_templates()[np.asarray(labels)] + 0.6 * noise)
    return np.clip(np.floor((img * 0.5 + 0.5) * 256.0), 0, 255).astype(np.int64).reshape(n, 3072)


def synthetic_cifar(n=50000, seed=1234, kind="uniform"):
    """Synthetic stand-ins for the CIFAR-10 arrays.  kind "uniform" (SURVEY 8(d)): uint8 images U{0..255} [n,3072] (CHW) and
    clean labels U{0..9} -- timing / plumbing only, the images carry no label.  kind "templates": ``template_images`` of the
    clean labels -- a learnable conditional distribution for end-to-end training runs."""
    rs = np.random.RandomState(seed)
    if kind == "uniform":
        return rs.randint(0, 256, size=(n, 3072), dtype=np.uint8), rs.randint(10, size=n)
    if kind != "templates":
        raise ValueError("unknown synthetic kind %r" % (kind,))
    labels = rs.randint(10, size=n)
    images = np.concatenate([template_images(rs, labels[i:i + 5000]) for i in range(0, n, 5000)]).astype(np.uint8)
    return images, labels


def inf_train_gen(train_gen):
    """gan_resnet.py:865-868."""
    while True:
        for batch in train_gen():
            yield batch


def inf_train_gen_G(train_gen, gen_bs_multiple=2):
    """gan_resnet.py:869-882: generator labels = consecutive (random, biased) batches of a second pass."""
    it = train_gen()
    while True:
        rnd, bia = [], []
        for _ in range(gen_bs_multiple):
            try:
                _, _, r, b, _ = next(it)
            except StopIteration:
                it = train_gen()
                _, _, r, b, _ = next(it)
            rnd.append(r)
            bia.append(b)
        yield np.concatenate(rnd, axis=0), np.concatenate(bia, axis=0)
