#!/usr/bin/env python3
"""Generate tests/golden/ref_cifar_*.npz by RUNNING the reference's own program: ``cifar10/gan_resnet.py`` is imported from
/root/reference and its ``main()`` executed for two iterations, with ``tensorflow`` replaced in sys.modules by the PyTorch-backed
look-alike scripts/refshim/tf1.py (TensorFlow 1.5 cannot be installed).  Runs only in the build container; the fixtures are data
(seeds, fed batches, random draws, the values the reference's graph produced) and travel with the repo -- the reference source
never does.

What a fixture pins (everything the reference's Python decides):
  * the variable set: names, shapes, creation order, initialiser kind, and sha256 + strided samples of every numpy-initialised
    value under ``np.random.seed(seed)`` (conv2d.py:83-140, linear.py:54-80, embedding.py:27-40, gan_resnet.py:499-520);
  * the graph wiring: two towers of BATCH_SIZE/2 on one device (gan_resnet.py:186-188), tower costs averaged (:697,786), which
    labels feed which projection (:563-586), loss assembly of the four algorithms (:587-685,734-778), disc_params / gen_params
    selection (:788-796), three AdamOptimizers and the lr decay (:700-705,802-817);
  * the training loop: the order and the feed_dict of every session.run (:919-947) including the label-corruption stream of
    common/data/cifar10.py under the same numpy seed, the G-step labels of inf_train_gen_G (:869-882);
  * per recorded run: fetched disc_cost / gen_cost, the gradients the optimiser applied and the variables after the update
    (norms + strided samples).
What it cannot pin: the arithmetic of TensorFlow's kernels -- the look-alike restates their documented semantics (Appendix C).

usage: python scripts/make_golden_reference.py            (writes tests/golden/ref_cifar_<alg>.npz for the four algorithms)
"""
import hashlib
import importlib
import os
import pickle
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.setrecursionlimit(20000)

STRIDE_SAMPLES = 128          # values kept per tensor (evenly strided)


def sample(a):
    a = np.asarray(a).reshape(-1)
    step = max(1, a.size // STRIDE_SAMPLES)
    return a[::step][:STRIDE_SAMPLES].astype(np.float32)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a, np.float32)).tobytes()).hexdigest()


def install_stubs():
    import tf1
    sys.modules["tensorflow"] = tf1
    # `import tensorflow.contrib.slim as slim` (mnist/utils.py:15) needs importable sub-modules
    tf1.__path__ = []
    for sub, obj in (("tensorflow.contrib", tf1.contrib), ("tensorflow.contrib.slim", tf1.contrib.slim)):
        m = types.ModuleType(sub)
        m.__dict__.update(vars(obj))
        sys.modules[sub] = m
    for name in ("imageio",):
        sys.modules[name] = types.ModuleType(name)
    import scipy
    misc = types.ModuleType("scipy.misc")
    misc.saved = []
    misc.imsave = lambda path, img: misc.saved.append((path, np.array(img)))
    misc.imread = lambda *a, **k: None
    sys.modules["scipy.misc"] = misc
    scipy.misc = misc
    # the Inception-score module downloads a frozen graph at import (inception_score_.py:24-48): out of scope
    for name in ("common.inception", "common.inception.inception_score_"):
        m = types.ModuleType(name)
        m.get_inception_score = lambda *a, **k: (0.0, 0.0)
        sys.modules[name] = m
    np.float, np.int = float, int          # aliases numpy >= 1.24 removed; the reference still uses them
    return tf1


def synthetic_cifar(data_dir):
    """50 000 train / 10 000 test rows of seeded uint8 pixels and labels (what cifar10.py:29-32 needs), never np.random's stream."""
    os.makedirs(data_dir)
    rs = np.random.RandomState(777)
    for i in range(5):
        with open(os.path.join(data_dir, "data_batch_%d" % (i + 1)), "wb") as f:
            pickle.dump({b"data": rs.randint(0, 256, size=(10000, 3072)).astype(np.uint8), b"labels": [int(v) for v in rs.randint(10, size=10000)]}, f)
    with open(os.path.join(data_dir, "test_batch"), "wb") as f:
        pickle.dump({b"data": rs.randint(0, 256, size=(10000, 3072)).astype(np.uint8), b"labels": [int(v) for v in rs.randint(10, size=10000)]}, f)


def run_reference(alg, seed, tf_seed, batch_size, niters, extra_flags):
    tf1 = install_stubs()
    tmp = tempfile.mkdtemp(prefix="refrun_")
    try:
        synthetic_cifar(os.path.join(tmp, "data", "cifar10", "cifar-10-batches-py"))
        run_dir = os.path.join(tmp, "cifar10")
        os.makedirs(run_dir)
        # the reference loads ./resnet-110/graph_optimized.pb at the end of main(): the look-alike stops there (StopReference)
        cwd = os.getcwd()
        os.chdir(run_dir)
        for k in [k for k in sys.modules if k == "gan_resnet" or k == "common" or (k.startswith("common.") and "inception" not in k)]:
            del sys.modules[k]
        sys.path.insert(0, os.path.join(REF, "cifar10"))
        tf1.reset(tf_seed)
        tf1.set_dtype(__import__("torch").float64)
        flags = dict(algorithm=alg, alpha=0.6, run="0", log_file=os.path.join(run_dir, "log.txt"), parent_dir=".", ngpus=1,
                     multi_gpu_multi_batch=True, batch_size=batch_size, niters=niters, inception_freq=10 ** 9, sample_freq=10 ** 9,
                     generated_label_accuracy_freq=10 ** 9, sample_save_freq=0, restore=False)
        flags.update(extra_flags)
        tf1._Flags.overrides = flags
        tf1.flags.FLAGS = tf1._Flags()
        for n in ("string", "integer", "float", "boolean", "bool"):
            setattr(tf1.flags, "DEFINE_" + n, tf1.flags.FLAGS._define)
        tf1.AdamOptimizer._count = 0

        # ---- record every session.run: fetched values, gradients applied, variables afterwards
        records = []
        orig_run = tf1.Session.run

        def run(self, fetches, feed_dict=None):
            fl = fetches if isinstance(fetches, (list, tuple)) else [fetches]
            before = {i: o.t for i, o in tf1.S.adam_slots.items()}
            out = orig_run(self, fetches, feed_dict)
            stepped = [i for i, o in tf1.S.adam_slots.items() if o.t != before.get(i, 0)]
            rec = dict(tf1.S.run_log[-1])
            rec["fetched"] = [None if v is None else np.asarray(v) for v in (out if isinstance(out, (list, tuple)) else [out])]
            rec["optimisers"] = stepped
            rec["n_fetches"] = len(fl)
            if stepped:
                rec["vars_after"] = {n: v.value.detach().numpy().copy() for n, v in tf1.S.variables.items()}
                rec["grads"] = {}
                for i in stepped:
                    rec["grads"].update({n: g.copy() for n, g in tf1.S.adam_slots[i].last_grads.items()})
            records.append(rec)
            return out
        tf1.Session.run = run
        # keep the gradients an optimiser applied
        orig_apply = tf1.AdamOptimizer.apply_gradients

        def apply_gradients(self, gv, global_step=None, name=None):
            op = orig_apply(self, gv, global_step, name)
            fn0, names, opt = op.fn, [v.name for g, v in gv if g is not None], self

            def fn(lr, *grads):
                opt.last_grads = {n: g.detach().numpy().copy() for n, g in zip(names, grads)}
                opt.last_lr = float(lr)
                return fn0(lr, *grads)
            op.fn = fn
            return op
        tf1.AdamOptimizer.apply_gradients = apply_gradients

        np.random.seed(seed)
        mod = importlib.import_module("gan_resnet")
        try:
            mod.main(None)
        except tf1.StopReference:
            pass
        finally:
            tf1.Session.run = orig_run
            tf1.AdamOptimizer.apply_gradients = orig_apply
            os.chdir(cwd)
            sys.path.remove(os.path.join(REF, "cifar10"))
        variables = list(tf1.S.variables.values())
        lrs = {i: getattr(o, "last_lr", None) for i, o in tf1.S.adam_slots.items()}
        return variables, records, mod, lrs
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def make(alg, extra_flags, tag, seed=1234, tf_seed=99, batch_size=4, niters=2):
    variables, records, mod, lrs = run_reference(alg, seed, tf_seed, batch_size, niters, extra_flags)
    out = {"algorithm": alg, "seed": seed, "tf_seed": tf_seed, "batch_size": batch_size, "towers": len(mod.DEVICES), "niters": niters,
           "flags": np.array(sorted("%s=%s" % kv for kv in extra_flags.items()))}
    out["var_names"] = np.array([v.name[:-2] for v in variables])
    out["var_shapes"] = np.array([",".join(str(s) for s in v._shape) for v in variables])
    out["var_trainable"] = np.array([v.trainable for v in variables])
    out["var_init_kind"] = np.array([v.init_kind[0] for v in variables])
    out["var_init_sha256"] = np.array([sha(v.initial) for v in variables])
    for v in variables:
        key = "init/" + v.name[:-2]
        # values that came from TensorFlow's own generators (u vectors, a default-initialised confusion matrix) cannot be
        # reproduced from the numpy seed: stored whole (small); numpy-initialised ones: strided samples next to the hash
        out[key] = v.initial.astype(np.float32) if v.init_kind[0] not in ("numpy", "constant") else sample(v.initial)
    # the session.run calls of the training loop (graph construction makes none)
    steps = [r for r in records if r["optimisers"]]
    out["n_runs"] = len(records)
    out["run_kinds"] = np.array(["+".join(str(i) for i in r["optimisers"]) or "eval" for r in records])
    keep_full = {0, len([r for r in steps if r["optimisers"] == [0]][:5]) - 1}      # first and fifth critic run
    gi = [k for k, r in enumerate(steps) if 1 in r["optimisers"]]
    keep_full |= set(gi[:1])                                                          # the first generator run
    for k, r in enumerate(steps):
        p = "run%02d/" % k
        out[p + "optimisers"] = np.array(r["optimisers"])
        out[p + "fetched"] = np.array([np.nan if (v is None or np.size(v) != 1) else float(v) for v in r["fetched"]])
        if k > max(gi[:1] or [0]):
            continue            # later runs: the fetched costs only (the replay tests stop after the first generator run)
        # feed_dict entries by POSITION (most placeholders are unnamed): gan_resnet.py:936-947 (critic run), :928-934 (generator run)
        fnames = (["images", "labels", "labels_random", "labels_biased", "inv_weights", "labels_random_G", "labels_biased_G", "iteration"]
                  if 0 in r["optimisers"] else ["iteration", "labels_random_G", "labels_biased_G"])
        assert len(fnames) == len(r["feeds"]), (fnames, list(r["feeds"]))
        for name, a in zip(fnames, r["feeds"].values()):
            out[p + "feed/" + name] = np.asarray(a)
        for j, (kind, a) in enumerate(r["draws"]):
            out[p + "draw%02d/%s" % (j, kind)] = a
        if k in keep_full:
            for n, g in r["grads"].items():
                out[p + "grad_norm/" + n[:-2]] = np.float64(np.linalg.norm(g))
                out[p + "grad/" + n[:-2]] = sample(g)
            for n, v in r["vars_after"].items():
                out[p + "after/" + n[:-2]] = sample(v)
    out["adam_lrs"] = np.array([-1.0 if lrs.get(i) is None else lrs[i] for i in sorted(lrs)])
    path = os.path.join(OUT, "ref_cifar_%s.npz" % tag)
    np.savez_compressed(path, **out)
    print(tag, "variables", len(variables), "runs", len(records), "optimiser runs", len(steps), "->", path, "%.1f kB" % (os.path.getsize(path) / 1e3))
    for k, r in enumerate(steps[:7]):
        print("   run", k, "optimisers", r["optimisers"], "fetched", [None if v is None else (float(v) if np.size(v) == 1 else v.shape) for v in r["fetched"]])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["rcgan", "rcgan-u", "biased", "unbiased"]
    for alg in which:
        if alg == "rcgan-u":
            make(alg, dict(perm_classifier=True, confuse_init=True), "rcganu")
        else:
            make(alg, {}, alg)
