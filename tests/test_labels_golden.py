"""Label-corruption indexing: oracle restatement vs vectors captured from the reference's own numpy
code (scripts/make_golden_labels.py).  Bit-exact (integer index work)."""
import glob
import os

import numpy as np
import pytest

from oracle import labels as L


def _synth(n, seed):
    return np.random.RandomState(seed).randint(10, size=n)


@pytest.mark.parametrize("fn", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "labels_cifar_*.npz"))))
def test_cifar_corruption_bit_exact(fn):
    g = np.load(fn)
    C = L.one_coin(float(g["alpha"]))
    rng = np.random.RandomState(int(g["seed"]))
    bs = int(g["batch_size"])
    for name, n, lseed in (("train", 50000, int(g["clean_train_seed"])), ("dev", 10000, int(g["clean_test_seed"]))):
        noisy, rnd, bia, inv = L.cifar_corrupt(_synth(n, lseed), C, rng)
        k = (n // bs) * bs
        assert np.array_equal(noisy[:k], g[name + "_noisy"])
        assert np.array_equal(rnd[:k], g[name + "_random"])
        assert np.array_equal(bia[:k].astype(np.int64), g[name + "_biased"])
        assert np.array_equal(np.argmax(inv[:k], 1), g[name + "_invrow"])
        assert np.array_equal(inv[:8], g[name + "_inv_first8"])


@pytest.mark.parametrize("fn", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "labels_mnist_*.npz"))))
def test_mnist_corruption_bit_exact(fn):
    g = np.load(fn)
    y = np.concatenate([_synth(60000, int(g["train_label_seed"])), _synth(10000, int(g["test_label_seed"]))])
    X = (np.arange(70000) % 251).astype(np.float64)
    Xs, y_actual, y_real, y_gen, y_fake, y_w, C = L.mnist_corrupt(
        X, y, float(g["alpha"]), confusion_class_depend=bool(g["depend"]), real_match=bool(g["match"]))
    assert np.array_equal(np.argmax(y_actual, 1), g["y_actual"])
    assert np.array_equal(np.argmax(y_real, 1), g["y_real"])
    assert np.array_equal(np.argmax(y_gen, 1), g["y_gen"])
    assert np.array_equal(np.argmax(y_fake, 1), g["y_fake"])
    assert np.array_equal(Xs.astype(np.uint8), g["x_first_pixel"])
    assert np.array_equal(C, g["C"])
    assert np.array_equal(y_w[:8], g["w_first8"])
    for a in (y_actual, y_real, y_gen, y_fake):
        assert np.array_equal(a.sum(1), np.ones(70000))
