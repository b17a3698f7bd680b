"""The Inception-score arithmetic (rcgan_amd/inception_score.py) against the reference's own get_inception_probs / preds2score,
executed by scripts/make_golden_inception.py on seeded logits (tests/golden/ref_inception_score.npz)."""
import os

import numpy as np
import pytest

import rcgan_amd  # noqa: F401
from rcgan_amd import inception_score as IS

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_inception_score.npz")


def _case():
    z = np.load(GOLDEN)
    rs = np.random.RandomState(int(z["seed"]))
    n, width = int(z["n"]), int(z["width"])
    logits = (rs.randn(n, width) * 3.0).astype(np.float32)
    images = rs.uniform(-1, 1, size=(n, 3, 4, 4)).astype(np.float32)
    images[:, 0, 0, 0] = np.arange(n)
    images[0, 0, 0, 0] = 0.0
    return z, logits, images


def test_probabilities_and_scores_equal_the_reference():
    z, logits, images = _case()
    ids = np.arange(len(images))
    seen = []

    def logits_fn(batch):
        i0 = len(seen) * IS.BATCH_SIZE
        seen.append(i0)
        return logits[ids[i0:i0 + IS.BATCH_SIZE]]
    probs = IS.get_inception_probs(images, logits_fn)
    assert probs.shape == (int(z["n_probs"]), 1000)                 # the incomplete last batch is dropped, 1000 of 1008 logits kept
    np.testing.assert_allclose(probs[::41, ::97], z["probs_sample"], rtol=2e-6, atol=0)
    for splits in (1, 3, 10):
        got = IS.preds2score(probs, splits)
        # (the reference carries float32 probabilities through the logarithms; float64 here: 1e-6 on the mean, 2e-5 on the spread)
        ref = z["score_splits%d" % splits]
        assert abs(got[0] - ref[0]) <= 1e-6 * ref[0] and abs(got[1] - ref[1]) <= 2e-5 * max(ref[1], 1e-3), (got, ref)


def test_input_checks_and_missing_classifier():
    _, logits, images = _case()
    with pytest.raises(RuntimeError, match="no classifier"):
        IS.get_inception_score(images, None)
    with pytest.raises(ValueError):
        IS.get_inception_score(images * 3.0, lambda b: logits[:128])
    with pytest.raises(ValueError):
        IS.get_inception_score(images[:, :2], lambda b: logits[:128])
    with pytest.raises(ValueError, match="expected"):
        IS.get_inception_probs(images, lambda b: logits[:128, :10])
    with pytest.raises(ValueError, match="at least"):
        IS.get_inception_probs(images[:100], lambda b: logits[:128])
    with pytest.raises(ValueError):
        IS.load_logits_fn("no_colon")
    assert IS.load_logits_fn("numpy:zeros") is np.zeros


def test_sample_view_is_the_reference_reshape():
    rows = np.arange(2 * 3072, dtype=np.float32).reshape(2, 3072)
    v = IS.samples_as_the_reference_feeds_them(rows)
    assert v.shape == (2, 3, 32, 32)
    assert v[1, 2, 5, 7] == rows[1, (5 * 32 + 7) * 3 + 2]           # (n, 32, 32, 3) view of the row, then channels first
