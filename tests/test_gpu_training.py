"""The engine LEARNS: a short end-to-end training run of the production path (captured graphs, device random stream, host
feeds) on the class-pattern synthetic images, scored by the nearest-pattern classifier (scripts/train_synthetic.py; the
long runs behind DESIGN's training section are profiles/r05_train_*.json).  A synthetic stand-in for the reference's
generated-label-accuracy curve (gan_resnet.py:424-455, 995-1005; README.md:75-80), NOT that curve."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


def test_generator_learns_the_conditional_distribution_in_bf16():
    """3000 iterations (1 G + 5 D updates each) of rcgan-u with the reference's run_rcganu.sh flags (learned confusion matrix,
    permutation regulariser, confuse_init) at the CIFAR preset's 40 % label noise, bf16, production path: the generated-label
    accuracy is far above chance (>= 0.6; the committed seeds read 1.00 / 0.78 / 1.00 at 3000 iterations and 1.00 from 4000 on), the learned
    confusion matrix's diagonal has moved from its start (0.2) towards the true 0.6, every loss stays finite.  ~25 s."""
    import train_synthetic as TS
    res = TS.run(algorithm="rcgan-u", dtype="bf16", iters=3000, eval_every=1000, alpha=0.6, batch=64, seed=0, perm_classifier=True, confuse_init=True)
    assert res["losses_finite"]
    accs = [c["gen_label_acc"] for c in res["curve"]]
    assert accs[-1] >= 0.6 and accs[-1] > accs[0], accs
    diag = [c["confusion_diag_mean"] for c in res["curve"]]
    assert 0.25 <= diag[-1] <= 0.7 and diag[-1] > 0.22, diag
    assert all(0.0 <= c["d_cost"] <= 3.0 for c in res["curve"]), res["curve"]


def test_noisy_labels_first_steps_are_finite_and_move_the_losses():
    """300 iterations at the bench's noise level (alpha = 0.6), rcgan and biased: finite losses, critic cost below its
    initial 2.0 (it has started to separate real from fake), generator cost finite -- the cheap guard that runs every round."""
    import train_synthetic as TS
    for alg in ("rcgan", "biased"):
        res = TS.run(algorithm=alg, dtype="bf16", iters=300, eval_every=150, alpha=0.6, batch=64, seed=0, n_train=50000)
        assert res["losses_finite"], alg
        assert res["curve"][-1]["d_cost"] < 1.95, (alg, res["curve"])
        assert np.isfinite(res["curve"][-1]["g_cost"])


def test_mnist_engine_learns_and_rcgan_beats_biased_under_label_noise():
    """One epoch (700 iterations of 1 D + 2 G updates, fp32, batch 100) of the MNIST engine on the class-pattern digits with the
    reference's presets: run_rcgan.sh at 70 % label noise reaches a generated-label accuracy >= 0.9 (the committed seeds: 1.0 from the
    first epoch on), run_biased.sh at 50 % noise stays near its ceiling P(true = y | noisy = y) = 0.5 (the committed seeds: 0.63 after
    one epoch, 0.51 after eight) -- the reference's plot (README.md:62-67) on a synthetic stand-in
    (scripts/train_synthetic_mnist.py, profiles/r05_train_mnist_*.json).  ~25 s."""
    import train_synthetic_mnist as TM
    r = TM.run("rcgan", alpha=0.3, epochs=1, seed=0, draws=10)
    b = TM.run("biased", alpha=0.5, epochs=1, seed=0, draws=10)
    assert r["losses_finite"] and b["losses_finite"]
    assert r["final_gen_label_acc"] >= 0.9, r["curve"]
    assert 0.3 <= b["final_gen_label_acc"] <= 0.8, b["curve"]
    assert r["final_gen_label_acc"] > b["final_gen_label_acc"] + 0.15
