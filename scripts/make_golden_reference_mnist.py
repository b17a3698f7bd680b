#!/usr/bin/env python3
"""tests/golden/ref_mnist_<case>.npz: the reference's own MNIST program (mnist/main.py -> model.DCGAN: build_model + train) run for
two iterations at batch 8 on the PyTorch-backed TensorFlow-1.x look-alike (scripts/refshim/tf1.py), with the run_*.sh flag sets.
Build container only; see scripts/make_golden_reference.py for what such a run pins and what it cannot.

MNIST specifics:
  * every trainable variable is initialised by TensorFlow's own generators (truncated_normal / random_normal, ops.py:57,74,108), so
    the fixture pins names, creation order, shapes, initialiser kind + stddev, the `constraint` (max-norm clip) of each variable and
    which variables each optimiser owns -- values are regenerated in the tests from the look-alike's seed in creation order and
    checked against the sha256 stored here;
  * the data: synthetic idx files (seeded), DCGAN.load_mnist's own seed-547 shuffle and label corruption, batch_z from numpy's stream
    as the reference draws it (model.py:342);
  * per session.run of the first iteration (1 D run, 2 G runs, model.py:347-372): feeds, fetched losses where the reference fetches
    them, applied gradients and variables afterwards (strided samples + norms), batch-norm moving statistics of the generator.
"""
import hashlib
import importlib
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(0, HERE)
from make_golden_reference import install_stubs, REF, OUT, sample, sha  # noqa: E402

sys.setrecursionlimit(20000)


def synthetic_mnist(data_dir):
    os.makedirs(data_dir)
    rs = np.random.RandomState(4321)
    for name, n in (("train", 60000), ("t10k", 10000)):
        with open(os.path.join(data_dir, "%s-images-idx3-ubyte" % name), "wb") as f:
            f.write(bytes(16))
            f.write(rs.randint(0, 256, size=(n, 28, 28), dtype=np.uint8).tobytes())
        with open(os.path.join(data_dir, "%s-labels-idx1-ubyte" % name), "wb") as f:
            f.write(bytes(8))
            f.write(rs.randint(0, 10, size=n).astype(np.uint8).tobytes())


def run(case, flags, tf_seed=7, batch=8):
    import types
    tf1 = install_stubs()
    for sub in ("tensorflow.python", "tensorflow.python.framework"):
        m = types.ModuleType(sub)
        sys.modules[sub] = m
    sys.modules["tensorflow.python.framework"].ops = types.SimpleNamespace()
    tmp = tempfile.mkdtemp(prefix="refmnist_")
    cwd = os.getcwd()
    try:
        synthetic_mnist(os.path.join(tmp, "data", "mnist"))
        run_dir = os.path.join(tmp, "mnist_run")
        os.makedirs(run_dir)
        os.chdir(run_dir)
        for k in [k for k in sys.modules if k in ("main", "model", "ops", "utils", "sn")]:
            del sys.modules[k]
        sys.path.insert(0, os.path.join(REF, "mnist"))
        tf1.reset(tf_seed)
        tf1.set_dtype(__import__("torch").float64)
        tf1.NONE_DIM = batch
        tf1.AdamOptimizer._count = 0
        fl = dict(train=True, epoch=1, batch_size=batch, train_size=2 * batch, data_dir=os.path.join(tmp, "data"), checkpoint_dir="ckpt",
                  logs_dir=os.path.join(run_dir, "logs"))
        fl.update(flags)
        tf1._Flags.overrides = fl
        tf1.flags.FLAGS = tf1._Flags()
        for n in ("string", "integer", "float", "boolean", "bool", "list"):
            setattr(tf1.flags, "DEFINE_" + n, tf1.flags.FLAGS._define)

        records = []
        orig_run = tf1.Session.run

        def srun(self, fetches, feed_dict=None):
            before = {i: o.t for i, o in tf1.S.adam_slots.items()}
            out = orig_run(self, fetches, feed_dict)
            stepped = [i for i, o in tf1.S.adam_slots.items() if o.t != before.get(i, 0)]
            rec = {"feeds": {k.name: np.asarray(v) for k, v in (feed_dict or {}).items()}, "optimisers": stepped,
                   "fetched": out if isinstance(out, (list, tuple)) else [out]}
            if stepped:
                rec["vars_after"] = {n: v.value.detach().numpy().copy() for n, v in tf1.S.variables.items()}
                rec["grads"] = {}
                for i in stepped:
                    rec["grads"].update(tf1.S.adam_slots[i].last_grads)
            records.append(rec)
            return out
        tf1.Session.run = srun
        orig_apply = tf1.AdamOptimizer.apply_gradients

        def apply_gradients(self, gv, global_step=None, name=None):
            op = orig_apply(self, gv, global_step, name)
            fn0, names, opt = op.fn, [v.name for g, v in gv if g is not None], self
            opt.var_names = names

            def fn(lr, *grads):
                opt.last_grads = {n: g.detach().numpy().copy() for n, g in zip(names, grads)}
                opt.last_lr = float(lr)
                return fn0(lr, *grads)
            op.fn = fn
            return op
        tf1.AdamOptimizer.apply_gradients = apply_gradients
        try:
            mod = importlib.import_module("main")
            import model as ref_model
            import utils as ref_utils
            ref_utils.dump_script = lambda *a, **k: None                     # copies *.py of the working directory

            def stop(self, config):
                raise tf1.StopReference("recover_labels")
            ref_model.DCGAN.recover_labels = stop
            try:
                mod.main(None)
            except tf1.StopReference:
                pass
        finally:
            tf1.Session.run = orig_run
            tf1.AdamOptimizer.apply_gradients = orig_apply
            os.chdir(cwd)
            sys.path.remove(os.path.join(REF, "mnist"))
        variables = list(tf1.S.variables.values())
        out = {"case": case, "tf_seed": tf_seed, "batch_size": batch, "flags": np.array(sorted("%s=%s" % kv for kv in flags.items()))}
        out["var_names"] = np.array([v.name[:-2] for v in variables])
        out["var_shapes"] = np.array([",".join(str(s) for s in v._shape) for v in variables])
        out["var_trainable"] = np.array([v.trainable for v in variables])
        out["var_init_kind"] = np.array([v.init_kind[0] for v in variables])
        out["var_init_std"] = np.array([float(v.init_kind[2]) if len(v.init_kind) > 2 else np.nan for v in variables])
        out["var_constrained"] = np.array([v.constraint is not None for v in variables])
        out["var_init_sha256"] = np.array([sha(v.initial) for v in variables])
        for i, o in sorted(tf1.S.adam_slots.items()):
            out["optimiser%d_vars" % i] = np.array([n[:-2] for n in o.var_names])
            out["optimiser%d_hyper" % i] = np.array([getattr(o, "last_lr", np.nan), o.b1, o.b2, o.eps])
        steps = [r for r in records if r["optimisers"]]
        out["run_kinds"] = np.array(["+".join(str(i) for i in r["optimisers"]) or "eval" for r in records])
        for k, r in enumerate(steps[:3]):                                    # the first iteration: D run, G run, G run
            p = "run%02d/" % k
            out[p + "optimisers"] = np.array(r["optimisers"])
            for name, a in r["feeds"].items():
                out[p + "feed/" + name] = np.asarray(a, np.float32)
            for n, g in r["grads"].items():
                out[p + "grad_norm/" + n[:-2]] = np.float64(np.linalg.norm(g))
                out[p + "grad/" + n[:-2]] = sample(g)
            for n, v in r["vars_after"].items():
                out[p + "after/" + n[:-2]] = sample(v)
        # the logging evals that follow the three optimiser runs of iteration 0 (model.py:374-398): d_loss_fake, d_loss_real, g_loss
        first = [i for i, r in enumerate(records) if r["optimisers"]][2]
        evals = [r for r in records[first + 1:first + 4]]
        out["log_losses_after_it0"] = np.array([float(np.asarray(r["fetched"][0])) for r in evals])
        path = os.path.join(OUT, "ref_mnist_%s.npz" % case)
        np.savez_compressed(path, **out)
        print(case, "variables", len(variables), "runs", len(records), [str(k) for k in out["run_kinds"]][:14], "->", "%.1f kB" % (os.path.getsize(path) / 1e3))
        print("   losses logged after iteration 0 (d_fake, d_real, g):", out["log_losses_after_it0"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    cases = {
        "rcgan": dict(algorithm="rcgan", alpha=0.5, disc_type="projection", estimate_confuse=False, add_noise=False, concat_y=False,
                      spectral_norm=True, max_norm=True),                                                     # run_rcgan.sh
        "rcganu": dict(algorithm="rcgan", alpha=0.5, disc_type="projection", estimate_confuse=True, add_noise=False, concat_y=False,
                       spectral_norm=True, max_norm=True),                                                    # run_rcganu.sh
        "biased": dict(algorithm="biased", alpha=0.5, disc_type="vanilla", loss_fn="ce", real_match=True, estimate_confuse=False,
                       add_noise=False, concat_y=False, spectral_norm=False, max_norm=False),                 # run_biased.sh
    }
    for c in (sys.argv[1:] or list(cases)):
        run(c, cases[c])
