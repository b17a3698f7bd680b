#!/usr/bin/env python3
"""Generate tests/golden/labels_*.npz by IMPORTING the reference's own numpy label-corruption code.

Runs only in the build container (needs /root/reference); the fixtures it writes are data
(seeded synthetic inputs + the reference's outputs) and travel with the repo.  The reference source
itself is never copied.

  * CIFAR: ``cifar10/common/data/cifar10.py:19-45`` ``cifar_generator`` imported as-is, fed synthetic
    pickles (50 000 train rows / 10 000 test rows of seeded labels) after ``np.random.seed(s)``.
  * MNIST: ``mnist/model.py:770-834`` ``DCGAN.load_mnist`` called as an unbound function on a stub
    ``self``; ``tensorflow``/``scipy.misc`` are stubbed in ``sys.modules`` because model.py imports them
    at module scope (they are not used by load_mnist), and the numpy aliases ``np.float``/``np.int``
    removed in numpy>=1.24 are restored.
"""
import os
import pickle
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def synth_labels(n, seed):
    return np.random.RandomState(seed).randint(10, size=n)


def make_cifar():
    # load the module by file path: the package __init__ (common/__init__.py:3) imports tensorflow,
    # the data module itself is pure numpy
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_cifar10_data", os.path.join(REF, "cifar10/common/data/cifar10.py"))
    ref_cifar = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_cifar)
    tmp = tempfile.mkdtemp()
    clean = synth_labels(50000, 2024)
    for i in range(5):
        with open(os.path.join(tmp, "data_batch_%d" % (i + 1)), "wb") as f:
            pickle.dump({b"data": np.zeros((10000, 1), np.uint8), b"labels": [int(v) for v in clean[i * 10000:(i + 1) * 10000]]}, f)
    clean_test = synth_labels(10000, 2025)
    with open(os.path.join(tmp, "test_batch"), "wb") as f:
        pickle.dump({b"data": np.zeros((10000, 1), np.uint8), b"labels": [int(v) for v in clean_test]}, f)
    for seed, alpha in ((1234, 0.6), (7, 0.8)):
        C = ((1 - alpha) / 9.0) * np.ones((10, 10)) + (alpha - (1 - alpha) / 9.0) * np.eye(10)
        np.random.seed(seed)
        bs = 64
        train_gen, dev_gen = ref_cifar.load(bs, tmp, C)      # train generator built first, then dev (same stream)
        out = {}
        for name, gen, n in (("train", train_gen, 50000), ("dev", dev_gen, 10000)):
            labs, rnd, bia, inv = [], [], [], []
            for _, l, r, b, w in gen():
                labs.append(l); rnd.append(r); bia.append(b); inv.append(w)
            labs, rnd, bia, inv = (np.concatenate(a) for a in (labs, rnd, bia, inv))
            k = (n // bs) * bs
            assert len(labs) == k
            out[name + "_noisy"] = labs.astype(np.int8)
            out[name + "_random"] = rnd.astype(np.int8)
            out[name + "_biased"] = bia.astype(np.int8)
            # inv_weights rows are rows of inv(C): store which row (argmax is unique: diagonal dominates)
            out[name + "_invrow"] = np.argmax(inv, axis=1).astype(np.int8)
            out[name + "_inv_first8"] = inv[:8]
        np.savez_compressed(os.path.join(OUT, "labels_cifar_seed%d_alpha%s.npz" % (seed, alpha)),
                            clean_train_seed=2024, clean_test_seed=2025, batch_size=bs, alpha=alpha, seed=seed, **out)
        print("cifar", seed, alpha, "P(noisy==clean)=%.4f" % (out["train_noisy"] == clean[:len(out["train_noisy"])]).mean())


def make_mnist():
    for name in ("tensorflow", "tensorflow.python", "tensorflow.python.framework", "tensorflow.python.framework.ops",
                 "tensorflow.contrib", "tensorflow.contrib.slim", "scipy.misc"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            sys.modules[name] = m
    tf = sys.modules["tensorflow"]

    class _Any:
        def __getattr__(self, k):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()
    tf.summary = _Any()
    tf.train = _Any()
    tf.contrib = sys.modules["tensorflow.contrib"]
    tf.contrib.slim = sys.modules["tensorflow.contrib.slim"]
    tf.python = sys.modules["tensorflow.python"]
    tf.python.framework = sys.modules["tensorflow.python.framework"]
    tf.python.framework.ops = sys.modules["tensorflow.python.framework.ops"]
    import scipy
    scipy.misc = sys.modules["scipy.misc"]
    np.float = float
    np.int = int
    sys.path.insert(0, os.path.join(REF, "mnist"))
    import model as ref_model

    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "mnist"))
    ytr = synth_labels(60000, 11).astype(np.uint8)
    yte = synth_labels(10000, 12).astype(np.uint8)
    # images: 1 byte of payload per image is enough to track the shuffle -> but the loader reshapes
    # to (N,28,28,1), so write full-size files whose first pixel carries (index mod 251)
    def write_images(path, n, off):
        a = np.zeros((n, 28, 28), np.uint8)
        a[:, 0, 0] = (np.arange(n) + off) % 251
        with open(path, "wb") as f:
            f.write(bytes(16)); f.write(a.tobytes())
    write_images(os.path.join(tmp, "mnist", "train-images-idx3-ubyte"), 60000, 0)
    write_images(os.path.join(tmp, "mnist", "t10k-images-idx3-ubyte"), 10000, 60000)
    for fn, y in (("train-labels-idx1-ubyte", ytr), ("t10k-labels-idx1-ubyte", yte)):
        with open(os.path.join(tmp, "mnist", fn), "wb") as f:
            f.write(bytes(8)); f.write(y.tobytes())

    for alpha, depend, match in ((0.3, False, False), (0.6, False, False), (0.125, False, False),
                                 (0.3, True, False), (0.3, False, True)):
        self = types.SimpleNamespace(data_dir=tmp, dataset_name="mnist", y_dim=10, alpha=alpha,
                                     config=types.SimpleNamespace(confusion_class_depend=depend, real_match=match))
        # reference opens the files in text mode (model.py:773); np.fromfile works on the fd regardless
        X, y_actual, y_real, y_gen, y_fake, y_w = ref_model.DCGAN.load_mnist(self)
        C = self.confusion_matrix_actual
        Cinv = np.linalg.inv(C)
        wrow = np.argmax(y_real, 1)
        assert np.allclose(y_w, Cinv[wrow])
        fn = "labels_mnist_seed547_alpha%s_dep%d_match%d.npz" % (alpha, int(depend), int(match))
        np.savez_compressed(os.path.join(OUT, fn), alpha=alpha, depend=depend, match=match,
                            train_label_seed=11, test_label_seed=12,
                            y_actual=np.argmax(y_actual, 1).astype(np.int8), y_real=np.argmax(y_real, 1).astype(np.int8),
                            y_gen=np.argmax(y_gen, 1).astype(np.int8), y_fake=np.argmax(y_fake, 1).astype(np.int8),
                            x_first_pixel=np.rint(X[:, 0, 0, 0] * 255).astype(np.uint8), C=C, w_first8=y_w[:8])
        print("mnist", alpha, depend, match, "P(y_real==y)=%.4f" % (np.argmax(y_real, 1) == np.argmax(y_actual, 1)).mean(),
              "P(y_fake==y_gen)=%.4f" % (np.argmax(y_fake, 1) == np.argmax(y_gen, 1)).mean())


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    make_cifar()
    make_mnist()
