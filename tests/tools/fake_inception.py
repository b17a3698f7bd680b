"""A stand-in classifier for the trainer's --inception_logits_fn hook (tests/test_gpu_cli.py): a fixed random projection of the
image batch to 1008 logits.  It only has to honour the interface ([128, 3, H, W] in [-1, 1] -> [128, >= 1000])."""
import numpy as np

_W = None


def logits(images):
    global _W
    images = np.asarray(images, np.float32)
    assert images.shape[0] == 128 and images.shape[1] == 3 and -1.0 <= images.min() and images.max() <= 1.0, images.shape
    flat = images.reshape(128, -1)
    if _W is None:
        _W = np.random.RandomState(5).randn(flat.shape[1], 1008).astype(np.float32) * 0.05
    return flat @ _W
