"""The oracle and the product's host logic against vectors produced by RUNNING THE REFERENCE'S OWN CODE
(tests/golden/ref_cifar_<alg>.npz, written by scripts/make_golden_reference.py: cifar10/gan_resnet.py imported from /root/reference
and its main() executed for two iterations at BATCH_SIZE 4 on a PyTorch-backed stand-in for the TensorFlow-1.x API).

Pinned to the reference by these tests (CPU, -m "not gpu"):
  * variable set: names, creation order, shapes, trainable flags; every numpy-initialised value bit for bit (sha256) from the numpy
    seed -- for the product's ``create_variables`` AND the oracle's ``init_params`` (conv2d.py:83-140, linear.py:54-80,
    embedding.py:27-40, gan_resnet.py:499-520; the draws of reuse=True calls included);
  * the step: tower split (two towers of B/2, gan_resnet.py:186-188,529-546), loss assembly of the four algorithms, which
    variables each optimiser owns, the lr schedule, TF-form Adam, the spectral-norm u update, the order of the training loop
    (five critic runs, one generator run) -- the numpy oracle (first run, all four algorithms) and the PyTorch-CPU restatement (the
    whole first iteration + the generator run) replay the recorded feeds / random draws in float64 and must land on the recorded
    costs (1e-9), gradients (1e-7) and updated variables (1e-9) of the reference's graph.
What stays unpinned: the arithmetic of TensorFlow's kernels (both sides restate it).
The GPU counterpart (product kernels on the same recorded runs) is tests/test_gpu_reference_golden.py.
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import cifar as oc
from oracle import nn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = [("rcgan", "rcgan", {}), ("rcganu", "rcgan-u", dict(perm_classifier=True, confuse_init=True)), ("biased", "biased", {}),
         ("unbiased", "unbiased", {})]
NSAMP = 128


def _draws(z, p):
    """(dequantisation noise, [z draws in evaluation order]) of a recorded run: the order in which the reference's graph reaches its
    random ops differs between the algorithms (rcgan-u evaluates the fake term of the critic cost first), the kinds do not."""
    keys = sorted(k for k in z.files if k.startswith(p + "draw"))
    noise = [z[k] for k in keys if k.endswith("random_uniform")]
    return (noise[0] if noise else None), [z[k] for k in keys if k.endswith("random_normal")]


def sample(a):
    a = np.asarray(a).reshape(-1)
    step = max(1, a.size // NSAMP)
    return a[::step][:NSAMP]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a, np.float32)).tobytes()).hexdigest()


def load(tag):
    return np.load(os.path.join(GOLDEN, "ref_cifar_%s.npz" % tag), allow_pickle=False)


@pytest.mark.parametrize("tag,alg,flags", CASES)
def test_variable_set_and_numpy_initial_values_equal_the_reference(tag, alg, flags):
    import rcgan_amd  # noqa: F401
    from rcgan_amd.cifar import create_variables
    z = load(tag)
    seed = int(z["seed"])
    names = [str(n) for n in z["var_names"]]
    shapes = {n: tuple(int(x) for x in str(s).split(",")) for n, s in zip(names, z["var_shapes"])}
    kinds = dict(zip(names, [str(k) for k in z["var_init_kind"]]))
    shas = dict(zip(names, [str(h) for h in z["var_init_sha256"]]))
    trainable = [n for n, t in zip(names, z["var_trainable"]) if t]
    # ---- product
    gs, ds, cs, U = create_variables(seed, alg, flags.get("perm_classifier", False), "linear", flags.get("confuse_init", False), 0.2)
    # creation order in the reference's graph: confusion_logits (:499-520), the generator tower (:541-546), the critic (:557-695)
    assert [n for n, _, _ in cs + gs + ds] == trainable
    for n, shp, v in cs + gs + ds:
        assert tuple(shp) == shapes[n] == tuple(v.shape), n
        if kinds[n] in ("numpy", "constant"):
            assert sha(v) == shas[n], "initial value of %s differs from the reference's" % n
    assert set(U) == {n for n in names if n not in trainable}
    for n in U:
        assert kinds[n] == "truncated_normal" and U[n].shape == shapes[n] and np.abs(U[n]).max() <= 2.0
    # ---- oracle
    P, Uo = oc.init_params(seed, alg, flags.get("perm_classifier", False), "linear", flags.get("confuse_init", False), 0.2)
    assert set(P) == set(trainable) and set(Uo) == set(U)
    for n in P:
        if kinds[n] in ("numpy", "constant"):
            assert sha(P[n]) == shas[n], n
    # the strided samples stored next to the hashes are consistent with them (guards the fixture itself)
    for n, _, v in gs + ds:
        if kinds[n] in ("numpy", "constant"):
            assert np.array_equal(sample(v).astype(np.float32), z["init/" + n])


def _initial_state(z, alg, flags):
    seed = int(z["seed"])
    P, U = oc.init_params(seed, alg, flags.get("perm_classifier", False), "linear", flags.get("confuse_init", False), 0.2)
    # what TensorFlow's own generators initialised (u vectors; stored whole in the fixture) replaces the second stream's values
    for n in U:
        U[n] = z["init/" + n].astype(np.float64).reshape(U[n].shape)
    if "confusion_logits" in P and not flags.get("confuse_init", False):
        P["confusion_logits"] = z["init/confusion_logits"]
    return {k: v.astype(np.float64) for k, v in P.items()}, U


def _d_batch(z, p, B):
    imgs = z[p + "feed/images"]
    noise, normals = _draws(z, p)
    assert noise.shape == (B, 3072)
    towers = normals[:2]
    return dict(real=oc.preprocess_real(imgs, noise), labels=z[p + "feed/labels"], labels_random=z[p + "feed/labels_random"],
                labels_biased=z[p + "feed/labels_biased"], inv_weights=z[p + "feed/inv_weights"], z=np.concatenate(towers))


def _check_tensors(z, prefix, got, tol, skip_below=0.0, floor=1e-30):
    keys = [k for k in z.files if k.startswith(prefix)]
    assert keys
    for k in keys:
        name = k[len(prefix):]
        ref = z[k].astype(np.float64)
        a = sample(got[name]).astype(np.float64)
        scale = max(float(np.abs(ref).max()), floor)
        if scale <= skip_below:
            continue            # a per-channel constant in front of a batch norm: the gradient is rounding noise (1e-16)
        assert float(np.abs(a - ref).max()) <= (tol + 3e-7) * scale, (k, float(np.abs(a - ref).max()), scale)   # (samples are stored as fp32)


@pytest.mark.parametrize("tag,alg,flags", CASES)
def test_numpy_oracle_reproduces_the_first_critic_run(tag, alg, flags):
    """The numpy oracle (float64) on the recorded feeds / random draws of the reference's first session.run: the logged gen_cost,
    disc_cost (mean of the two tower costs), the gradients the reference's optimiser applied, the variables after its update."""
    z = load(tag)
    B, ntow = int(z["batch_size"]), int(z["towers"])
    assert ntow == 2
    P, U = _initial_state(z, alg, flags)
    cfg = dict(algorithm=alg, C=oc.c_alpha(0.6), perm_classifier=flags.get("perm_classifier", False), perm_multiplier=1.0)
    kinds = [str(k) for k in z["run_kinds"] if str(k) != "eval"]
    assert kinds[:6] == ["0"] * 5 + (["1+2"] if alg == "rcgan-u" else ["1"]), kinds      # five critic runs, then the generator run (:919-947)
    p = "run00/"
    # the run also fetches gen_cost for the log (gan_resnet.py:936-947): evaluated on the weights BEFORE the update
    gb = dict(labels_random_G=z[p + "feed/labels_random_G"], labels_biased_G=z[p + "feed/labels_biased_G"],
              z=np.concatenate(_draws(z, p)[1][2:4]))
    gcost, _ = oc.g_grads({k: v.copy() for k, v in P.items()}, {k: v.copy() for k, v in U.items()}, cfg, gb, ntow, dtype=np.float64)
    assert abs(gcost - z[p + "fetched"][2]) <= 1e-9 * max(1.0, abs(gcost)), (gcost, z[p + "fetched"][2])
    cost, grads = oc.d_grads(P, U, cfg, _d_batch(z, p, B), ntow, dtype=np.float64)       # (mutates U: the spectral-norm u update)
    oc.apply_adam(P, grads, oc.AdamState(), 2e-4 * oc.lr_decay(int(z[p + "feed/iteration"])))
    assert abs(cost - z[p + "fetched"][0]) <= 1e-9 * max(1.0, abs(cost)), (cost, z[p + "fetched"][0])
    assert abs(z[p + "fetched"][0] - z[p + "fetched"][1]) == 0.0            # disc_cost is disc_wgan (:697-699)
    gmax = max(float(z[k]) for k in z.files if k.startswith(p + "grad_norm/"))
    for k in grads:
        assert abs(np.linalg.norm(grads[k]) - float(z[p + "grad_norm/" + k])) <= 1e-7 * max(float(z[p + "grad_norm/" + k]), 1e-9 * gmax), k
    _check_tensors(z, p + "grad/", grads, 1e-7, skip_below=1e-9 * gmax)
    _check_tensors(z, p + "after/", {**P, **U}, 1e-9, floor=1e-4)      # (a variable that is nothing but rounding-noise updates: |w| ~ 1e-13)


@pytest.mark.parametrize("tag,alg,flags", CASES[:2])
def test_torch_port_replays_five_critic_runs_and_the_generator_run(tag, alg, flags):
    """The whole recorded iteration (five critic runs, then the generator run -- with the confusion-matrix optimiser for rcgan-u)
    replayed by the PyTorch-CPU restatement in float64 from the seeded initial state: every run's cost, and after the generator
    run its gradients and the updated variables, as the reference's graph produced them."""
    import torch
    from oracle.torch_port import CifarTorch, adam_tf_torch
    z = load(tag)
    B, ntow = int(z["batch_size"]), int(z["towers"])
    P, U = _initial_state(z, alg, flags)
    cfg = dict(algorithm=alg, C=oc.c_alpha(0.6), perm_classifier=flags.get("perm_classifier", False), perm_multiplier=1.0)
    net = CifarTorch(P, U, torch.float64)
    slots = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in net.P.items()}
    steps = {"D": 0, "G": 0, "C": 0}

    def update(names, grads, key, lr):
        steps[key] += 1
        with torch.no_grad():
            for k, g in zip(names, grads):
                w, m, v = adam_tf_torch(net.P[k], g, slots[k][0], slots[k][1], steps[key], lr, 0.0, 0.9)
                net.P[k].copy_(w)
                slots[k] = (m, v)
            net.U.update(net.U_new)
            net.U_new = {}

    for run in range(5):
        p = "run%02d/" % run
        b = _d_batch(z, p, B)
        Bt = B // ntow
        costs = [net.disc_cost(cfg, {k: v[t * Bt:(t + 1) * Bt] for k, v in b.items()}) for t in range(ntow)]
        cost = sum(costs) / ntow
        names = [k for k in net.P if k.startswith("Discriminator/")]
        grads = torch.autograd.grad(cost, [net.P[k] for k in names])
        assert abs(float(cost.detach()) - z[p + "fetched"][0]) <= 1e-9 * max(1.0, abs(float(cost.detach()))), (run, float(cost.detach()), z[p + "fetched"][0])
        update(names, grads, "D", 2e-4 * oc.lr_decay(int(z[p + "feed/iteration"])))
    p = "run05/"
    zs = _draws(z, p)[1]
    costs = []
    for t in range(ntow):
        sl = slice(t * B, (t + 1) * B)          # GEN_BS_MULTIPLE * B / towers = B fakes per tower (:715-719)
        costs.append(net.gen_cost(cfg, dict(labels_random_G=z[p + "feed/labels_random_G"][sl], labels_biased_G=z[p + "feed/labels_biased_G"][sl], z=zs[t])))
    cost = sum(costs) / ntow
    names = [k for k in net.P if k.startswith("Generator/")]
    cn = [k for k in net.P if k == "confusion_logits"]
    grads = torch.autograd.grad(cost, [net.P[k] for k in names + cn])
    got = {k: g.numpy() for k, g in zip(names + cn, grads)}
    gmax = max(float(z[k]) for k in z.files if k.startswith(p + "grad_norm/"))
    for k in got:
        assert abs(np.linalg.norm(got[k]) - float(z[p + "grad_norm/" + k])) <= 1e-6 * max(float(z[p + "grad_norm/" + k]), 1e-9 * gmax), k
    _check_tensors(z, p + "grad/", got, 1e-6, skip_below=1e-9 * gmax)
    update(names, grads[:len(names)], "G", 2e-4 * oc.lr_decay(int(z[p + "feed/iteration"])))
    if cn:
        update(cn, grads[len(names):], "C", 2e-4)           # lr * confuse_multiplier (1.0), undecayed (gan_resnet.py:810-817)
    _check_tensors(z, p + "after/", {k: v.detach().numpy() for k, v in {**net.P, **net.U}.items()}, 1e-8, floor=1e-4)
    assert float(z["adam_lrs"][0]) == pytest.approx(2e-4 * oc.lr_decay(1), rel=1e-12)


def test_sample_grid_layout_equals_the_reference(tmp_path):
    """common/misc.py:215-244 save_images and mnist/utils.py:44-67 merge + :246-250 image_manifold_size, imported from the
    reference when the fixture was made (scripts/make_golden_grids.py): the array they hand to the image writer for seeded
    inputs == the pixels of the PNG host.save_images writes (both CLIs use it: train_cifar.py, train_mnist.py)."""
    import rcgan_amd  # noqa: F401
    from PIL import Image
    from rcgan_amd import host
    z = np.load(os.path.join(GOLDEN, "ref_grids.npz"))
    for key in ("cifar100", "cifar12", "gray16"):
        path = str(tmp_path / (key + ".png"))
        host.save_images(z[key + "_in"], path)
        got = np.asarray(Image.open(path)).astype(np.float64)
        assert got.shape == z[key + "_grid"].shape and np.array_equal(got, z[key + "_grid"]), key
    for key in ("mnist64", "mnist100"):
        path = str(tmp_path / (key + ".png"))
        host.save_images(z[key + "_in"][..., 0].astype(np.uint8), path)           # what train_mnist.py passes: [n, 28, 28] uint8
        got = np.asarray(Image.open(path)).astype(np.float64)
        assert got.shape == z[key + "_grid"].shape and np.array_equal(got, z[key + "_grid"]), key


# ----------------------------------------------------------------------------------------------------------------------
# MNIST: mnist/main.py -> model.DCGAN (build_model + train) run by scripts/make_golden_reference_mnist.py
# ----------------------------------------------------------------------------------------------------------------------
MNIST_CASES = [("rcgan", dict(algorithm="rcgan", disc_type="projection", estimate_confuse=False, loss_fn="hinge", spectral_norm=True, max_norm=True)),
               ("rcganu", dict(algorithm="rcgan", disc_type="projection", estimate_confuse=True, loss_fn="hinge", spectral_norm=True, max_norm=True)),
               ("biased", dict(algorithm="biased", disc_type="vanilla", estimate_confuse=False, loss_fn="ce", spectral_norm=False, max_norm=False))]


def _mnist_initial_values(z):
    """Every variable of the recorded run, regenerated from the look-alike's seed in creation order (all MNIST initialisers are
    TensorFlow-side generators: ops.py:57,74,108, sn.py:36) and checked against the stored sha256."""
    rs = np.random.RandomState(int(z["tf_seed"]))
    vals = {}
    for name, shp, kind, std, h in zip(z["var_names"], z["var_shapes"], z["var_init_kind"], z["var_init_std"], z["var_init_sha256"]):
        name, kind = str(name), str(kind)
        shape = tuple(int(x) for x in str(shp).split(","))
        if kind == "truncated_normal":
            x = rs.normal(0.0, 1.0, size=shape)
            while True:
                bad = np.abs(x) > 2.0
                if not bad.any():
                    break
                x[bad] = rs.normal(0.0, 1.0, size=int(bad.sum()))
            v = (float(std) * x).astype(np.float32)
        elif kind == "random_normal":
            v = (float(std) * rs.normal(size=shape)).astype(np.float32)
        elif kind == "glorot_uniform":
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            v = rs.uniform(-lim, lim, size=shape).astype(np.float32)
        else:
            v = (np.ones(shape) if name.endswith(("gamma", "moving_variance")) else np.zeros(shape)).astype(np.float32)
        assert sha(v) == str(h), name
        vals[name] = v
    return vals


@pytest.mark.parametrize("case,cfg", MNIST_CASES)
def test_mnist_variable_set_equals_the_reference(case, cfg):
    import rcgan_amd  # noqa: F401
    from oracle import mnist as om
    from rcgan_amd.mnist import create_variables
    z = np.load(os.path.join(GOLDEN, "ref_mnist_%s.npz" % case))
    names = [str(n) for n in z["var_names"]]
    shapes = {n: tuple(int(x) for x in str(s).split(",")) for n, s in zip(names, z["var_shapes"])}
    kinds = dict(zip(names, [str(k) for k in z["var_init_kind"]]))
    stds = dict(zip(names, [float(s) for s in z["var_init_std"]]))
    trainable = [n for n, t in zip(names, z["var_trainable"]) if t]
    proj = cfg["disc_type"] == "projection"
    gs, ds, cs, S, U = create_variables(0, cfg["disc_type"], cfg["estimate_confuse"], True, cfg["spectral_norm"], ())
    P, So, Uo = om.init_params(0, cfg["disc_type"], cfg["estimate_confuse"], True, cfg["spectral_norm"], ())
    # creation order of the trainable variables (model.py:96-262: confusion_logits, generator, discriminator, classifier)
    assert list(P) == trainable
    assert sorted(n for n, _, _ in gs + ds + cs) == sorted(trainable)
    assert set(S) | set(U) == set(names) - set(trainable) == set(So) | set(Uo)
    for n, shp, v in gs + ds + cs:
        assert tuple(shp) == shapes[n] == tuple(v.shape) == tuple(P[n].shape), n
        if kinds[n] == "constant":
            assert np.array_equal(v, P[n]) and len(np.unique(v)) == 1, n
        elif kinds[n] == "truncated_normal":         # conv filters (ops.py:57): N(0, 0.02) re-drawn beyond two sigma
            assert np.abs(v).max() <= 2 * stds[n] * (1 + 1e-6) and abs(v.std() / (0.8796 * stds[n]) - 1) < (0.05 if v.size >= 2000 else 0.5), (n, v.std())
        elif kinds[n] == "random_normal":            # deconv filters / dense matrices (ops.py:74,108): N(0, 0.02), not truncated
            assert abs(v.std() / stds[n] - 1) < (0.05 if v.size >= 2000 else 0.5) and (v.size < 5000 or np.abs(v).max() > 2 * stds[n]), (n, v.std())
    for n in U:
        assert kinds[n] == "truncated_normal" and stds[n] == 1.0 and U[n].shape == shapes[n]
    # which optimiser owns what (model.py:243-262): d_vars = names containing 'd_' (the classifier lands there), g_vars, the confusion matrix
    assert sorted(str(n) for n in z["optimiser0_vars"]) == sorted(n for n, _, _ in ds)
    assert sorted(str(n) for n in z["optimiser1_vars"]) == sorted(n for n, _, _ in gs)
    if cfg["estimate_confuse"]:
        assert [str(n) for n in z["optimiser2_vars"]] == ["confusion_logits"] and float(z["optimiser2_hyper"][0]) == pytest.approx(2e-4 * 10.0)
    for i in (0, 1):
        assert tuple(z["optimiser%d_hyper" % i][1:]) == (0.5, 0.999, 1e-8) and float(z["optimiser%d_hyper" % i][0]) == pytest.approx(2e-4)
    # the max-norm constraint (ops.py:102-111) sits on exactly the projection head's dense layers
    constrained = sorted(n for n, c in zip(names, z["var_constrained"]) if c)
    want = sorted(n for n in trainable if n.startswith(("discriminator/d_h4_lin", "discriminator/d_h5_y_lin"))) if (proj and cfg["max_norm"]) else []
    assert constrained == want


@pytest.mark.parametrize("case,cfg", MNIST_CASES)
def test_mnist_oracle_replays_the_first_iteration_of_the_reference(case, cfg):
    """One D run and two G runs (model.py:347-372) by the numpy oracle in float64 on the recorded feeds, from the regenerated initial
    values: the gradients the reference's optimisers applied, the variables after each update (max-norm clip and the generator's
    batch-norm moving averages included), and the losses the reference logs after the iteration (model.py:374-390)."""
    from oracle import labels as LB
    from oracle import mnist as om
    z = np.load(os.path.join(GOLDEN, "ref_mnist_%s.npz" % case))
    vals = _mnist_initial_values(z)
    trainable = [str(n) for n, t in zip(z["var_names"], z["var_trainable"]) if t]
    P = {n: vals[n].astype(np.float64) for n in trainable}
    S = {n: v.astype(np.float64) for n, v in vals.items() if n.endswith(("moving_mean", "moving_variance"))}
    U = {n: v.astype(np.float64) for n, v in vals.items() if n.endswith("/u")}
    ocfg = dict(cfg, perm_regularizer=True, perm_multiplier=10.0, C=LB.one_coin(0.5), concat_y=False, concat_y_layers=(), confuse_multiplier=10.0)
    tr = om.Trainer(P, S, U, ocfg)
    kinds = [str(k) for k in z["run_kinds"]]
    assert kinds[:3] == ["0", "1+2", "1+2"] if cfg["estimate_confuse"] else kinds[:3] == ["0", "1", "1"]
    b = dict(images=z["run00/feed/real_images"], z=z["run00/feed/z"], y_real=z["run00/feed/y_real"], y_gen=z["run00/feed/y_gen"],
             y_fake=z["run00/feed/y_fake"], y_real_weights=z["run00/feed/y_real_weights"])

    def cmp(p, grads):
        gmax = max(float(z[k]) for k in z.files if k.startswith(p + "grad_norm/"))
        for k, g in grads.items():
            ref = float(z[p + "grad_norm/" + k])
            assert abs(np.linalg.norm(g) - ref) <= 1e-6 * max(ref, 1e-9 * gmax), (p, k, np.linalg.norm(g), ref)
        _check_tensors(z, p + "grad/", grads, 1e-6, skip_below=1e-9 * gmax)
        for k in [k for k in z.files if k.startswith(p + "after/")]:
            name = k[len(p + "after/"):]
            if name.startswith("discriminator/") and name.endswith(("moving_mean", "moving_variance")):
                continue        # the critic's moving averages are never read; their update order inside one run is TensorFlow's business
            got = {**P, **S, **U}[name]
            ref = z[k].astype(np.float64)
            scale = max(float(np.abs(ref).max()), 1e-4)
            # (Adam divides by |g| + eps: where |g| ~ 1e-5 a 1e-9 difference of the gradient shows as 1e-6 of the step)
            assert float(np.abs(sample(got) - ref).max()) <= 1e-5 * scale, (k, float(np.abs(sample(got) - ref).max()), scale)

    for dt in (P, S, U):
        for k in dt:
            dt[k] = dt[k].astype(np.float64)
    _, g = om.d_grads(P, S, U, ocfg, b, dtype=np.float64)
    om.apply_adam(P, g, tr.ad, tr.lr, tr.beta1, tr.clip)
    cmp("run00/", g)
    for run in (1, 2):
        _, g = om.g_grads(P, S, U, ocfg, b, dtype=np.float64)
        gc = {k: g.pop(k) for k in list(g) if k == "confusion_logits"}
        om.apply_adam(P, g, tr.ag, tr.lr, tr.beta1)
        if gc:
            om.apply_adam(P, gc, tr.ac, tr.lr * 10.0, tr.beta1)
        cmp("run%02d/" % run, {**g, **gc})
    # the three logging evals are three session runs, and every run of the critic advances the spectral-norm power iteration
    # (sn.py:53-58, update_collection=None): errD_fake, errD_real and errG each see a different u
    got = []
    for key in ("d_loss_fake", "d_loss_real", "g_loss"):
        L = om.losses(om.Net(P, S, U, "d", ocfg, np.float64), b)
        got.append(float(L[key].v))
    assert np.allclose(got, z["log_losses_after_it0"], rtol=1e-7, atol=1e-9), (got, z["log_losses_after_it0"])
