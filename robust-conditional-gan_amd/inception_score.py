"""Inception score of generated samples: the host arithmetic of the reference's common/inception/inception_score_.py around a
classifier the caller supplies.

The reference pushes 50 000 generator samples through the Inception-v3 graph that TF-GAN downloads when the module is imported
(inception_score_.py:31-48, ``tfgan.eval.run_inception``); neither the graph nor TF-GAN is part of the checkout, and there is no
network here.  What IS in the checkout, and is restated here, is everything on either side of that network:

  * how gan_resnet.py collects the samples (:836-845): 100 random-label samples per Generator call, the [n, 3072] rows viewed as
    (n, 32, 32, 3) and transposed to (n, 3, 32, 32) -- applied to rows that are already channel-major, i.e. the classifier does NOT
    see the picture the generator drew; the trainer reproduces exactly that;
  * get_inception_probs (:50-59): batches of 128, the incomplete last batch dropped, the first 1000 logits, softmax;
  * preds2score (:61-68): exp(mean_x KL(p(y|x) || p(y))) over `splits` contiguous parts, mean and standard deviation over the parts.

`logits_fn(images)` takes float32 [128, 3, H, W] in [-1, 1] and returns [128, >= 1000] logits (the reference resizes to 299 x 299 in
front of the network; that belongs to the supplied callable).  Without one the score cannot be computed and the functions say so.
Pinned by tests/golden/ref_inception_score.npz (scripts/make_golden_inception.py executes the reference's own two functions)."""
import importlib

import numpy as np

BATCH_SIZE = 128                                   # inception_score_.py:27


def load_logits_fn(spec):
    """'package.module:callable' -> the callable (train_cifar.py --inception_logits_fn)."""
    if not spec or ":" not in spec:
        raise ValueError("--inception_logits_fn takes 'package.module:callable', got %r" % (spec,))
    mod, name = spec.split(":", 1)
    fn = getattr(importlib.import_module(mod), name)
    if not callable(fn):
        raise TypeError("%s is not callable" % spec)
    return fn


def get_inception_probs(inps, logits_fn):
    """inception_score_.py:50-59."""
    if logits_fn is None:
        raise RuntimeError("Inception score: no classifier.  The reference downloads Inception-v3 through TF-GAN at import "
                           "(inception_score_.py:31-48); pass logits_fn (train_cifar.py --inception_logits_fn module:callable)")
    preds = []
    for i in range(len(inps) // BATCH_SIZE):
        pred = np.asarray(logits_fn(inps[i * BATCH_SIZE:(i + 1) * BATCH_SIZE]), dtype=np.float64)
        if pred.ndim != 2 or pred.shape[0] != BATCH_SIZE or pred.shape[1] < 1000:
            raise ValueError("logits_fn returned %s for a batch of %d: expected [%d, >= 1000]" % (pred.shape, BATCH_SIZE, BATCH_SIZE))
        preds.append(pred[:, :1000])
    if not preds:
        raise ValueError("Inception score needs at least %d samples, got %d" % (BATCH_SIZE, len(inps)))
    preds = np.concatenate(preds, 0)
    # (the reference exponentiates the raw logits; subtracting the row maximum first is the same quotient without the overflow)
    e = np.exp(preds - preds.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def preds2score(preds, splits):
    """inception_score_.py:61-68."""
    scores = []
    for i in range(splits):
        part = preds[(i * preds.shape[0] // splits):((i + 1) * preds.shape[0] // splits), :]
        kl = part * (np.log(part) - np.log(np.expand_dims(np.mean(part, 0), 0)))
        scores.append(np.exp(np.mean(np.sum(kl, 1))))
    return float(np.mean(scores)), float(np.std(scores))


def get_inception_score(images, logits_fn, splits=10):
    """inception_score_.py:70-84: images float [n, 3, H, W] in [-1, 1] -> (mean, std) over the splits."""
    if not isinstance(images, np.ndarray) or images.ndim != 4 or images.shape[1] != 3:
        raise ValueError("images: numpy [n, 3, H, W], got %s" % (getattr(images, "shape", type(images)),))
    if np.max(images[0]) > 1 or np.min(images[0]) < -1:
        raise ValueError("images must lie in [-1, 1]")
    return preds2score(get_inception_probs(images, logits_fn), splits)


def samples_as_the_reference_feeds_them(rows):
    """gan_resnet.py:843-844: generator rows [n, 3072] (channel-major) -> what reaches the classifier, [n, 3, 32, 32]."""
    return np.asarray(rows).reshape((-1, 32, 32, 3)).transpose(0, 3, 1, 2)
