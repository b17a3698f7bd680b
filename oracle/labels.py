"""Label-corruption restatement (oracle; test infrastructure).  PINNED: bit-exact against vectors
captured from the reference's own numpy code (tests/golden/labels_*.npz, made by
scripts/make_golden_labels.py).

``rng`` is a ``numpy.random.RandomState`` (or the ``numpy.random`` module after ``seed``): the
reference draws from the legacy global MT19937 stream, whose output is version-stable.
"""
import numpy as np


def one_coin(alpha):
    """cifar10/gan_resnet.py:106, mnist/model.py:809."""
    return ((1 - alpha) / 9.0) * np.ones((10, 10)) + (alpha - (1 - alpha) / 9.0) * np.eye(10)


def class_dependent(alpha):
    """mnist/model.py:812-816 (np.linspace defaults to 50 points; only the first 10 are used)."""
    C = np.zeros((10, 10))
    mean_diag = np.linspace(0.15, -0.15 + 2 * alpha)
    for i in range(10):
        C[i, :] = (1. - mean_diag[i]) / 9.
        C[i, i] = mean_diag[i]
    return C


def cifar_corrupt(labels_clean, C, rng):
    """cifar10/common/data/cifar10.py:29-41.  Returns (labels_noisy, labels_random[50000],
    labels_biased[50000], labels_inv_weights[50000,10]); rows past len(labels_clean) stay zero,
    as in the reference (the arrays are always sized 50000, also for the 10000-row test split)."""
    labels = np.array(labels_clean).copy()
    labels_random = rng.randint(10, size=50000)
    labels_biased = np.zeros((50000,))
    labels_inv_weights = np.zeros((50000, 10))
    C_inv = np.linalg.inv(C)
    for i in range(len(labels)):
        labels[i] = np.nonzero(rng.multinomial(1, C[labels[i], :], size=1))[1][0]
        labels_inv_weights[i] = C_inv[labels[i], :]
        labels_biased[i] = np.nonzero(rng.multinomial(1, C[labels_random[i], :], size=1))[1][0]
    return labels, labels_random, labels_biased, labels_inv_weights


def mnist_corrupt(X, y, alpha, confusion_class_depend=False, real_match=False, seed=547):
    """mnist/model.py:795-832.  X [N,...], y int[N] in file order.  Returns
    (X_shuffled, y_actual, y_real, y_gen, y_fake, y_real_weights, C); label arrays are one-hot float64."""
    X = np.array(X)
    y = np.array(y).astype(int)
    rng = np.random.RandomState(seed)
    rng.shuffle(X)
    rng = np.random.RandomState(seed)
    rng.shuffle(y)
    n = len(y)
    y_actual = np.zeros((n, 10))
    y_real = np.zeros((n, 10))
    y_fake = np.zeros((n, 10))
    y_gen = np.zeros((n, 10))
    y_real_weights = np.zeros((n, 10))
    C = class_dependent(alpha) if confusion_class_depend else one_coin(alpha)
    C_inv = np.linalg.inv(C)
    # the reference keeps drawing from the global stream left by the second shuffle
    for i, label in enumerate(y):
        y_actual[i, label] = 1
        y_real[i] = rng.multinomial(1, C[y[i], :], size=1)
        y_real_weights[i] = C_inv[np.where(y_real[i] == 1)[0], :]
        y_gen_label = rng.randint(10, size=1)
        y_gen[i, int(y_gen_label[0])] = 1
        if real_match:
            y_gen[i] = y_real[i]
            y_gen_label = np.argmax(y_gen[i])
        else:
            y_gen_label = int(y_gen_label[0])
        y_fake[i] = rng.multinomial(1, C[int(y_gen_label), :], size=1)
    return X, y_actual, y_real, y_gen, y_fake, y_real_weights, C


def mnist_noise_schedule(epoch, alpha, noise_alpha, noise_start, noise_end, y_dim=10):
    """mnist/model.py:293-319 (--add_noise annealing of the extra label noise).  Not importable from
    the reference (lives inside DCGAN.train): restated, unpinned."""
    alpha_start = ((noise_alpha - (1. - alpha) / (y_dim - 1)) / (alpha - (1. - alpha) / (y_dim - 1)))
    alpha_start = min(1.0, alpha_start)
    if noise_alpha > 0.9:
        raise ValueError('same rate activated, but effective noise alpha {} > 0.9!'.format(noise_alpha))
    if alpha_start == 1.:
        end_epoch = noise_start
    else:
        end_epoch = noise_start + ((noise_end - noise_start) / (0.9 - noise_alpha) * (alpha - noise_alpha))
        end_epoch = min(noise_end, end_epoch)
    if epoch < noise_start:
        na = alpha_start
    elif epoch < end_epoch:
        na = alpha_start + (1. - alpha_start) * (epoch - noise_start) / (end_epoch - noise_start)
    else:
        na = 1.0
    return min(1.0, na)


def mnist_add_noise(y_real_orig, y_fake_orig, noise_alpha_eff, rng, y_dim=10):
    """mnist/model.py:321-333: per row, one multinomial draw for y_real then one for y_fake."""
    noise_C = ((1 - noise_alpha_eff) / (y_dim - 1)) * np.ones((y_dim, y_dim)) + \
        (noise_alpha_eff - (1 - noise_alpha_eff) / (y_dim - 1)) * np.eye(y_dim)
    y_real = np.zeros_like(y_real_orig)
    y_fake = np.zeros_like(y_fake_orig)
    for ii in range(len(y_real)):
        y_real[ii] = rng.multinomial(1, noise_C[np.argmax(y_real_orig[ii]), :], size=1)
        y_fake[ii] = rng.multinomial(1, noise_C[np.argmax(y_fake_orig[ii]), :], size=1)
    return y_real, y_fake
