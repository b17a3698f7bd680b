#!/usr/bin/env python3
"""tests/golden/ref_grids.npz: the sample-grid LAYOUT of the reference, by importing its own writers --
cifar10/common/misc.py:215-244 ``save_images`` (the array it hands to scipy.misc.imsave, captured by a stub) and
mnist/utils.py:44-67 ``merge`` + :246-250 ``image_manifold_size`` -- on seeded inputs.  Build container only.
(What scipy.misc.imsave then does to the array -- a min/max contrast stretch in the scipy <= 1.1 the reference needs -- is a
third-party behaviour that cannot be imported here and is not pinned.)"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(0, HERE)
from make_golden_reference import install_stubs, REF, OUT  # noqa: E402


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    tf1 = install_stubs()
    tf1.reset(0)
    misc = load("ref_cifar_misc", os.path.join(REF, "cifar10/common/misc.py"))
    saved = sys.modules["scipy.misc"].saved
    rs = np.random.RandomState(5)
    out = {}
    for key, shape in (("cifar100", (100, 6, 6, 3)), ("cifar12", (12, 4, 4, 3)), ("gray16", (16, 5, 5))):
        X = rs.randint(0, 256, size=shape).astype(np.int32)
        misc.save_images(X, "unused.png")
        out[key + "_in"] = X.astype(np.uint8)
        out[key + "_grid"] = np.asarray(saved[-1][1], np.float64).astype(np.uint8)      # (integers 0..255 in a float array)
        assert np.array_equal(out[key + "_grid"], np.asarray(saved[-1][1]))
    sys.modules["six.moves"] = __import__("six").moves
    utils = load("ref_mnist_utils", os.path.join(REF, "mnist/utils.py"))
    for key, n in (("mnist64", 64), ("mnist100", 100)):
        X = rs.randint(0, 256, size=(n, 7, 7, 1)).astype(np.float64)
        out[key + "_in"] = X.astype(np.uint8)
        out[key + "_size"] = np.array(utils.image_manifold_size(n))
        out[key + "_grid"] = utils.merge(X, utils.image_manifold_size(n)).astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "ref_grids.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
