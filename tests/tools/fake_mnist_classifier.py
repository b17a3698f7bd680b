"""A stand-in for the frozen MNIST classifier of mnist/utils.py:276 (tests/test_gpu_cli.py): predicts from mean brightness.
It only has to honour the interface (float [100, 28, 28, 1] -> 100 class indices)."""
import numpy as np


def predict(images):
    images = np.asarray(images)
    assert images.shape == (100, 28, 28, 1), images.shape
    return np.clip((images.mean(axis=(1, 2, 3)) * 10).astype(int), 0, 9)
