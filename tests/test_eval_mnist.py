"""MNIST generated-label accuracy bookkeeping (rcgan_amd/eval_mnist.py) against the reference's expressions (mnist/utils.py:292-305)."""
import numpy as np
import pytest

import rcgan_amd  # noqa: F401
from rcgan_amd.eval_mnist import generated_label_accuracy, regroup_by_class


def _samples(draws=100):
    s = np.zeros((draws, 100, 2, 2, 1))
    for d in range(draws):
        for j in range(100):
            s[d, j] = d * 1000 + j            # pixel value = (draw, slot); slot // 10 is the class the sampler was asked for
    return s


def test_regrouping_equals_the_reference_expression():
    s = _samples()
    ref = (s.transpose((1, 0, 2, 3, 4)).reshape((10, 10) + s.shape[1:]).reshape((10, -1) + s.shape[2:]))     # utils.py:292-295 verbatim shape algebra
    got = regroup_by_class(s)
    assert np.array_equal(got, ref)
    assert all(((got[c, :, 0, 0, 0] % 1000) // 10 == c).all() for c in range(10))
    assert regroup_by_class(_samples(30)).shape == (10, 300, 2, 2, 1)


def test_accuracy_is_the_mean_of_batch_accuracies():
    s = _samples()
    calls = []

    def predict(batch):                       # right for classes 0..6, always 0 for the others, except: class 9's first batch is right
        assert batch.shape == (100, 2, 2, 1)
        c = int(batch[0, 0, 0, 0] % 1000) // 10
        calls.append(c)
        if c <= 6 or (c == 9 and calls.count(9) == 1):
            return np.full(100, c)
        return np.zeros(100, int)
    acc = generated_label_accuracy('mnist', s, predict)
    assert calls == [c for c in range(10) for _ in range(10)]                 # 1000 per class in batches of 100, class by class
    assert abs(acc - (70 + 1) / 100.0) < 1e-12
    # an incomplete tail is dropped (utils.py:299): 250 per class -> two batches
    calls.clear()
    generated_label_accuracy('mnist', _samples(25), lambda b: (calls.append(0), np.zeros(100, int))[1])
    assert len(calls) == 20


def test_errors():
    with pytest.raises(ValueError, match="only implemented for mnist"):
        generated_label_accuracy('cifar', _samples(), lambda b: np.zeros(100))
    with pytest.raises(RuntimeError, match="no classifier"):
        generated_label_accuracy('mnist', _samples(), None)
    with pytest.raises(ValueError):
        generated_label_accuracy('mnist', np.zeros((3, 50, 2, 2, 1)), lambda b: np.zeros(100))
    with pytest.raises(ValueError, match="nothing to score"):
        generated_label_accuracy('mnist', _samples(5), lambda b: np.zeros(100))
    with pytest.raises(ValueError, match="predictions"):
        generated_label_accuracy('mnist', _samples(), lambda b: np.zeros(7))
