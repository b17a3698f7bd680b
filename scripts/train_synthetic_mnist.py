#!/usr/bin/env python3
"""End-to-end training run of the MNIST engine on the "templates" synthetic digits (no MNIST here, no network): does it learn, and
does rcgan beat the biased baseline under label noise as in the reference's plot (README.md:62-67: generated-label accuracy ~ 0.98 for
rcgan up to 70 % noise, ~ 0.5 for biased at 50 %).

The loop is DCGAN.train's (mnist/model.py:287-372: per iteration one D update and two G updates on the same batch; epochs over the
corrupted data set), flags as the reference's presets: rcgan = run_rcgan.sh (projection D, spectral norm, max norm, hinge, perm
regulariser), biased = run_biased.sh (vanilla D, cross entropy, real_match, no spectral norm).  fp32, batch 100 (the reference's
default).  After every epoch the sampler draws 100 x 100 samples on ten labels per class (model.py:470-489) and
eval_mnist.TemplateClassifier -- the stand-in for the missing mnist_dcnn/graph_optimized.pb -- scores them with the reference's own
bookkeeping (utils.py:273-306).  A synthetic stand-in for the reference's curve, NOT that curve.

  python scripts/train_synthetic_mnist.py --algorithm rcgan --alpha 0.3 --epochs 8 --out gpurun_out/r05_train_mnist_rcgan.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

PRESETS = {
    "rcgan": dict(algorithm="rcgan", disc_type="projection", loss_fn="hinge", estimate_confuse=False, real_match=False, spectral_norm=True),
    "rcgan-u": dict(algorithm="rcgan", disc_type="projection", loss_fn="hinge", estimate_confuse=True, real_match=False, spectral_norm=True),
    "biased": dict(algorithm="biased", disc_type="vanilla", loss_fn="ce", estimate_confuse=False, real_match=True, spectral_norm=False),
}


def run(preset="rcgan", alpha=0.3, epochs=8, batch=100, n_train=70000, seed=0, draws=20, log=None):
    import rcgan_amd  # noqa: F401
    from rcgan_amd import data_mnist as DM
    from rcgan_amd.eval_mnist import TemplateClassifier, generated_label_accuracy
    from rcgan_amd.mnist import Z_DIM, MnistRCGAN
    p = PRESETS[preset]
    X, y = DM.synthetic(n_train, 1234, "templates")
    data = DM.corrupt(X, y, alpha, False, p["real_match"])
    m = MnistRCGAN(algorithm=p["algorithm"], alpha=alpha, batch_size=batch, dtype="f32", seed=seed, disc_type=p["disc_type"], loss_fn=p["loss_fn"],
                   estimate_confuse=p["estimate_confuse"], perm_regularizer=True, perm_multiplier=10.0, spectral_norm=p["spectral_norm"],
                   max_norm=True, confusion_matrix=data["C"])
    clf = TemplateClassifier()
    rs = np.random.RandomState(100 + seed)
    labels100 = np.eye(10, dtype=np.float32)[[c for c in range(10) for _ in range(10)]]
    eval_z = [np.random.RandomState(7 + k).uniform(-1, 1, size=(100, Z_DIM)).astype(np.float32) for k in range(draws)]
    curve = []
    t0 = time.time()
    its = len(data["X"]) // batch
    for epoch in range(epochs):
        for idx in range(its):
            lo, hi = idx * batch, (idx + 1) * batch
            m.set_inputs(images=data["X"][lo:hi], z=rs.uniform(-1, 1, [batch, Z_DIM]).astype(np.float32), y_real=data["y_real"][lo:hi],
                         y_gen=data["y_gen"][lo:hi], y_fake=data["y_fake"][lo:hi], y_real_weights=data["y_real_weights"][lo:hi])
            m.iteration()
        samples = np.array([m.sampler(z, labels100).reshape(100, 28, 28, 1) for z in eval_z])
        acc = generated_label_accuracy("mnist", samples, clf)
        ev = m.evaluate()
        rec = {"epoch": epoch + 1, "iterations": (epoch + 1) * its, "gen_label_acc": round(float(acc), 4),
               "d_loss": round(float(ev["d_loss_real"] + ev["d_loss_fake"]), 4), "g_loss": round(float(ev["g_loss"]), 4),
               "elapsed_s": round(time.time() - t0, 1)}
        curve.append(rec)
        if log:
            log(json.dumps(rec))
    finite = all(np.isfinite(c["d_loss"]) and np.isfinite(c["g_loss"]) for c in curve)
    m.ctx.close()
    return {"what": "synthetic stand-in for the reference's MNIST generated-label-accuracy plot (README.md:62-67); class-pattern digits "
                    "(data_mnist.template_images), classifier = nearest class pattern (eval_mnist.TemplateClassifier); NOT the MNIST curve",
            "preset": preset, "alpha": alpha, "noise_level": round(1 - alpha, 3), "batch": batch, "epochs": epochs, "iterations": epochs * its,
            "dtype": "f32", "seed": seed, "losses_finite": bool(finite), "wall_s": round(time.time() - t0, 1), "curve": curve,
            "final_gen_label_acc": curve[-1]["gen_label_acc"], "max_gen_label_acc": max(c["gen_label_acc"] for c in curve)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="rcgan", choices=sorted(PRESETS))
    ap.add_argument("--alpha", type=float, default=0.3)
    ap.add_argument("--epochs", type=int, default=8)
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    res = run(a.preset, a.alpha, a.epochs, a.batch, seed=a.seed, log=lambda s: print(s, flush=True))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(json.dumps(res) + "\n")
    print(json.dumps({k: v for k, v in res.items() if k != "curve"}))


if __name__ == "__main__":
    main()
