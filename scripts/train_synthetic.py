#!/usr/bin/env python3
"""End-to-end training run of the CIFAR engine on the "templates" synthetic stand-in (no CIFAR-10 here, no network):
does the engine LEARN, does rcgan beat biased under label noise as in the reference's plot (README.md:75-80,
gan_resnet.py:995-1005), and does bf16 track fp32 at matched steps.

The loop is train_cifar.main's (gan_resnet.py:919-947: [G step] + the N_CRITIC critic steps per iteration, host feeds through
CifarRCGAN.feed_host, lr decay by iteration) without its file outputs; every --eval_every iterations the generator draws 1000
samples, 100 per class (gan_resnet.py:847-861), and eval_cifar.TemplateClassifier reads their class (exact on these images).
A synthetic stand-in for the reference's CIFAR-10 curve, NOT that curve.

  python scripts/train_synthetic.py --algorithm rcgan --dtype bf16 --iters 10000 --out gpurun_out/r05_train_rcgan_bf16.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def run(algorithm="rcgan", dtype="bf16", iters=10000, eval_every=500, alpha=0.6, batch=64, seed=0, n_train=50000, log=None,
        perm_classifier=False, confuse_init=False):
    import rcgan_amd  # noqa: F401
    from rcgan_amd import data as D
    from rcgan_amd.cifar import N_CRITIC, Z_DIM, CifarRCGAN
    from rcgan_amd.eval_cifar import TemplateClassifier, generated_label_accuracy
    np.random.seed(1000 + seed)                    # the label-noise stream (the reference leaves it unseeded)
    C = D.C_ALPHA(alpha)
    tx, ty = D.synthetic_cifar(n_train, 1234, "templates")
    train_gen = D.cifar_generator(tx, ty, batch, C)
    gen, gen_G = D.inf_train_gen(train_gen), D.inf_train_gen_G(train_gen, 2)
    m = CifarRCGAN(algorithm=algorithm, alpha=alpha, batch_size=batch, dtype=dtype, seed=seed, perm_classifier=perm_classifier,
                   confuse_init=confuse_init)
    clf = TemplateClassifier()
    ers = np.random.RandomState(77)
    labels100 = [k for k in range(10) for _ in range(10)]
    eval_z = [ers.normal(size=(100, Z_DIM)).astype("float32") for _ in range(10)]      # the same 1000 latents at every evaluation

    def accuracy():
        s = np.concatenate([m.sample(labels100, z) for z in eval_z], axis=0)
        s = ((s + 1.) * (255.99 / 2)).astype("int32").reshape(-1, 32, 32, 3)
        cm = m.confusion_matrix_value() if algorithm == "rcgan-u" else None
        plain = generated_label_accuracy(s, np.concatenate([labels100] * 10), classifier=clf)
        perm = generated_label_accuracy(s, np.concatenate([labels100] * 10), confusion_matrix=cm, classifier=clf) if cm is not None else None
        return plain, perm

    curve, pending = [], []
    d_hist, g_hist = [], []
    t_start = time.time()
    for it in range(iters):
        if it > 0:
            r, b = next(gen_G)
            m.feed_host("g", labels_random_G=r, labels_biased_G=b)
            m.g_step(iteration=it)
        batches = [next(gen) for _ in range(N_CRITIC)]
        m.feed_host("gf", labels_random_all=np.concatenate([b[2] for b in batches]))
        m.prepare_critic_fakes()
        feeds = []
        for images, labels, rnd, bia, inv in batches:
            second = rnd if algorithm in ("biased", "unbiased") else bia
            feeds.append(dict(images=images, labels=labels, labels_random=rnd, labels_biased=bia, inv_weights=inv,
                              labels_all=np.concatenate([labels, second])))
        m.critic_steps(feeds, iteration=it)
        m.iteration = it + 1
        pending.append(m.enqueue_losses())
        if len(pending) >= 256 or (it + 1) % eval_every == 0 or it + 1 == iters:
            for dc, gc in m.fetch_losses(pending):
                d_hist.append(float(dc)), g_hist.append(float(gc))
            pending = []
        if (it + 1) % eval_every == 0 or it + 1 == iters:
            acc, acc_perm = accuracy()
            rec = {"iteration": it + 1, "gen_label_acc": round(acc, 4), "d_cost": round(float(np.mean(d_hist[-eval_every:])), 4),
                   "g_cost": round(float(np.mean(g_hist[-eval_every:])), 4), "elapsed_s": round(time.time() - t_start, 1)}
            if acc_perm is not None:
                rec["gen_label_acc_perm"] = round(acc_perm, 4)
                rec["confusion_diag_mean"] = round(float(np.mean(np.diag(m.confusion_matrix_value()))), 4)
            curve.append(rec)
            if log:
                log(json.dumps(rec))
    finite = bool(np.all(np.isfinite(d_hist)) and np.all(np.isfinite(g_hist)))
    secs = time.time() - t_start
    m.ctx.close()
    return {"what": "synthetic stand-in for README.md:75-80 (generated-label accuracy under label noise); class-pattern images "
                    "(data.template_images), classifier = nearest class pattern (eval_cifar.TemplateClassifier); NOT the CIFAR-10 curve",
            "algorithm": algorithm, "dtype": dtype, "alpha": alpha, "noise_level": round(1 - alpha, 3), "batch": batch, "iterations": iters,
            "seed": seed, "n_critic": N_CRITIC, "losses_finite": finite, "wall_s": round(secs, 1),
            "ms_per_iteration_incl_host_feeds_and_eval": round(secs / iters * 1e3, 3), "curve": curve,
            "final_gen_label_acc": curve[-1]["gen_label_acc"] if curve else None,
            "max_gen_label_acc": max(c["gen_label_acc"] for c in curve) if curve else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algorithm", default="rcgan")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=10000)
    ap.add_argument("--eval_every", type=int, default=500)
    ap.add_argument("--alpha", type=float, default=0.6)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--perm_classifier", action="store_true")
    ap.add_argument("--confuse_init", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    res = run(a.algorithm, a.dtype, a.iters, a.eval_every, a.alpha, a.batch, a.seed, log=lambda s: print(s, flush=True),
              perm_classifier=a.perm_classifier, confuse_init=a.confuse_init)
    line = json.dumps(res)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(line + "\n")
    print(json.dumps({k: v for k, v in res.items() if k != "curve"}))


if __name__ == "__main__":
    main()
