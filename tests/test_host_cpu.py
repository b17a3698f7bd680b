"""CPU-only checks (no GPU needed): the C-ABI library loads and exports every symbol the header declares,
the product's host-side logic (variable creation, label streams, schedules) matches the oracle / golden
vectors, and the data-parallel averaging is right under a 2-process gloo group."""
import glob
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    import rcgan_amd  # noqa: F401
    from rcgan_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "rcgan_hip.h")).read()
    declared = set(re.findall(r"\b(rcgan_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"rcgan_ctx"}
    assert len(declared) > 50
    lib = _lib.load()                       # raises AttributeError on a missing export
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.rcgan_version().startswith(b"rcgan_hip")
    # size helpers are pure host functions: callable without a GPU
    d = _lib.ConvDesc(128, 32, 32, 256, 256, 3, 3, 1, _lib.BF16, 0)
    assert lib.rcgan_conv_prepared_bytes(d) >= 2 * 2 * 9 * 256 * 256
    assert lib.rcgan_sn_save_floats(1152, 128) == 3 * 1152 + 4 * 128 + 4 + 36 * (128 + 2)
    assert lib.rcgan_half_dtype() == _lib.BF16
    # the fp16 build of the same sources exports the same ABI and reports its 16-bit dtype
    lib16 = _lib.load("f16")
    for name in declared:
        assert hasattr(lib16, name), name
    assert lib16.rcgan_half_dtype() == _lib.F16 and b"fp16" in lib16.rcgan_version()
    d16 = _lib.ConvDesc(128, 32, 32, 256, 256, 3, 3, 1, _lib.F16, 0)
    assert lib16.rcgan_conv_prepared_bytes(d16) == lib.rcgan_conv_prepared_bytes(d)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import rcgan_amd  # noqa: F401
    from rcgan_amd.runtime import Context
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Context(0, "bf16")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "robust-conditional-gan_amd")
    for f in glob.glob(os.path.join(pkg, "*.py")):
        src = open(f).read()
        assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_variable_creation_matches_oracle():
    import rcgan_amd  # noqa: F401
    from rcgan_amd import cifar as pc
    from oracle import cifar as oc
    for alg, perm, ptype in (("rcgan", False, "linear"), ("rcgan-u", True, "linear"), ("rcgan-u", True, "2layer")):
        gs, ds, cs, U = pc.create_variables(5, alg, perm, ptype, True, 0.2)
        P, U2 = oc.init_params(5, alg, perm, ptype, True, 0.2)
        names = [n for n, _, _ in cs + gs + ds]
        assert names == list(P)            # reference creation order
        for n, shp, v in cs + gs + ds:
            assert tuple(shp) == P[n].shape and np.array_equal(v, P[n]), n
        assert set(U) == set(U2) and all(np.array_equal(U[k], U2[k]) for k in U)
    assert np.array_equal(pc.C_ALPHA(0.6), oc.c_alpha(0.6))
    for it in (0, 1, 49999, 50000, 99999):
        assert pc.lr_decay(it) == oc.lr_decay(it)


@pytest.mark.parametrize("fn", sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "labels_cifar_*.npz"))))
def test_product_label_streams_bit_exact(fn):
    """The product's own loader (data.py) against the vectors captured from the reference's numpy code."""
    import rcgan_amd  # noqa: F401
    from rcgan_amd import data as D
    g = np.load(fn)
    C = D.C_ALPHA(float(g["alpha"]))
    rng = np.random.RandomState(int(g["seed"]))
    bs = int(g["batch_size"])
    clean = np.random.RandomState(int(g["clean_train_seed"])).randint(10, size=50000)
    images = np.zeros((50000, 1), np.uint8)
    gen = D.cifar_generator(images, clean, bs, C, rng)
    labs, rnd, bia, inv = [], [], [], []
    for _, l, r, b, w in gen():
        labs.append(l); rnd.append(r); bia.append(b); inv.append(w)
    labs, rnd, bia, inv = (np.concatenate(a) for a in (labs, rnd, bia, inv))
    assert np.array_equal(labs, g["train_noisy"]) and np.array_equal(rnd, g["train_random"])
    assert np.array_equal(bia.astype(np.int64), g["train_biased"])
    assert np.array_equal(np.argmax(inv, 1), g["train_invrow"]) and np.array_equal(inv[:8], g["train_inv_first8"])
    # generator-label stream: two consecutive batches per G step, restarting at the end of the pass
    gg = D.inf_train_gen_G(gen, 2)
    r0, b0 = next(gg)
    assert np.array_equal(r0, g["train_random"][:2 * bs]) and np.array_equal(b0.astype(np.int64), g["train_biased"][:2 * bs])


_DP_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import rcgan_amd
from rcgan_amd.dp import shard_rows, allreduce_sum_, world_info
from oracle import cifar as oc
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
rank, world = world_info()
rs = np.random.RandomState(3)
B = 4
P, U = oc.init_params(0, "rcgan")
C = oc.c_alpha(0.6)
lab = rs.randint(10, size=B)
full = dict(real=oc.preprocess_real(rs.randint(0, 256, size=(B, 3072)), rs.uniform(0, 1 / 128., size=(B, 3072))),
            labels=lab, labels_random=rs.randint(10, size=B), labels_biased=rs.randint(10, size=B),
            inv_weights=np.linalg.inv(C)[lab], z=rs.randn(B, 128))
mine = {k: shard_rows(v, rank, world) for k, v in full.items()}
cfg = dict(algorithm="rcgan", C=C)
cost, grads = oc.d_grads(P, dict(U), cfg, mine, dtype=np.float64)
names = sorted(grads)
flat = torch.from_numpy(np.concatenate([grads[k].ravel() for k in names]))
allreduce_sum_(flat)
flat /= world                                   # grad_scale = 1/world in the Adam kernel
if rank == 0:
    cost2, g2 = oc.d_grads(P, dict(U), cfg, full, ntowers=2, dtype=np.float64)   # the reference's tower graph
    ref = np.concatenate([g2[k].ravel() for k in names])
    err = float(np.abs(flat.numpy() - ref).max())
    print("DP_MAXERR %%.3e" %% err)
    assert err < 1e-12, err
dist.barrier()
dist.destroy_process_group()
'''


def test_dp_two_ranks_gloo_equals_tower_mean(tmp_path):
    """2 gloo ranks, each differentiating its contiguous shard, all-reduce(sum)/2 == gradient of the
    reference's 2-tower cost (mean of tower costs, per-shard batch statistics)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER % dict(root=ROOT, port=port))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "DP_MAXERR" in outs[0]


@pytest.mark.parametrize("fn", sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "labels_mnist_*.npz"))))
def test_product_mnist_label_streams_bit_exact(fn):
    """The product's load_mnist corruption (rcgan_amd.data_mnist) against the vectors the reference's own
    DCGAN.load_mnist produced (scripts/make_golden_labels.py)."""
    import rcgan_amd  # noqa: F401
    from rcgan_amd import data_mnist as DM
    g = np.load(fn)
    synth = lambda n, seed: np.random.RandomState(seed).randint(0, 10, size=n).astype(np.int64)
    y = np.concatenate([synth(60000, int(g["train_label_seed"])), synth(10000, int(g["test_label_seed"]))])
    X = (np.arange(70000) % 251).astype(np.float64)
    d = DM.corrupt(X, y, float(g["alpha"]), confusion_class_depend=bool(g["depend"]), real_match=bool(g["match"]))
    assert np.array_equal(np.argmax(d["y_actual"], 1), g["y_actual"])
    assert np.array_equal(np.argmax(d["y_real"], 1), g["y_real"])
    assert np.array_equal(np.argmax(d["y_gen"], 1), g["y_gen"])
    assert np.array_equal(np.argmax(d["y_fake"], 1), g["y_fake"])
    assert np.array_equal(np.round(d["X"] * 255.).astype(np.uint8), g["x_first_pixel"])
    assert np.array_equal(d["C"], g["C"])
    assert np.array_equal(d["y_real_weights"][:8], g["w_first8"])


def test_product_mnist_noise_schedule_matches_oracle():
    """--add_noise annealing (mnist/model.py:293-333): product host code vs the oracle restatement, schedule values and
    the re-corruption stream."""
    import rcgan_amd  # noqa: F401
    from oracle import labels as OL
    from rcgan_amd import data_mnist as DM
    for alpha, na, s, e in ((0.125, 0.3, 30, 80), (0.6, 0.3, 2, 9), (0.3, 0.3, 5, 6)):
        for ep in (0, 1, s - 1, s, s + 1, (s + e) // 2, e, e + 3):
            assert DM.noise_schedule(ep, alpha, na, s, e) == OL.mnist_noise_schedule(ep, alpha, na, s, e)
    rs = np.random.RandomState(3)
    yr = np.eye(10)[rs.randint(10, size=500)]
    yf = np.eye(10)[rs.randint(10, size=500)]
    a = DM.add_noise(yr, yf, 0.7, np.random.RandomState(11))
    b = OL.mnist_add_noise(yr, yf, 0.7, np.random.RandomState(11))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    with pytest.raises(ValueError):
        DM.noise_schedule(0, 0.5, 0.95, 1, 2)


def test_bench_gpus_n_starts_n_ranks(tmp_path):
    """`python bench.py --gpus N` outside a launcher must start N ranks itself (round-3 verdict: it printed n_gpus: 1): --dry-run prints the
    torch.distributed.run command; inside a launcher (WORLD_SIZE set) or at N = 1 there is none; a rank whose WORLD_SIZE disagrees with
    --gpus refuses to run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    run = lambda args, e: subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=300)
    r = run(["--gpus", "2", "--steps", "3", "--master-port", "29777", "--dry-run"], env)
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launcher"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29777"
    tail = cmd[cmd.index(os.path.join(root, "bench.py")) + 1:]
    assert tail == ["--gpus", "2", "--steps", "3"]                   # the ranks get the same flags, minus the launcher's own
    assert json.loads(run(["--dry-run"], env).stdout.strip().splitlines()[-1])["launcher"] is None
    assert json.loads(run(["--gpus", "2", "--dry-run"], dict(env, WORLD_SIZE="2")).stdout.strip().splitlines()[-1])["launcher"] is None
    bad = run(["--gpus", "4", "--steps", "1"], dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "WORLD_SIZE=2" in (bad.stderr + bad.stdout)


def test_rccl_that_cannot_be_loaded_is_an_error_code_not_a_crash():
    """comm.hip binds librccl at run time; when it cannot be loaded rcgan_comm_unique_id returns RCGAN_ERCCL and rcgan_comm_load_error says why
    (round-3 advisor: the message was built from a second dlerror() call, i.e. from NULL).  RCGAN_RCCL_LIB points the loader at a file that does
    not exist; a fresh process, because the binding is cached per process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); import rcgan_amd; from rcgan_amd import _lib as L; lib = L.load(); "
            "buf = C.create_string_buffer(128); rc = lib.rcgan_comm_unique_id(buf); why = lib.rcgan_comm_load_error().decode(); "
            "print(rc, '|', why)") % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RCGAN_RCCL_LIB="/nonexistent/librccl_missing.so"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rc, why = r.stdout.strip().splitlines()[-1].split(" | ")
    assert int(rc) == -5 and "librccl_missing.so" in why             # RCGAN_ERCCL


def test_oracle_kink_policy_lists_and_flips_undecidable_rectifier_inputs():
    """oracle.tape.Kinks (test infrastructure of the MNIST step-parity tests): inputs within eps of zero are listed in call order,
    a flipped one takes the other branch's derivative, values are untouched, and the policy is off by default."""
    from oracle.tape import Kinks, Tape, Var
    x = np.array([[1.0, -2.0, 3e-9, -1e-9]])
    def grad(leaky):
        t = Tape()
        v = Var(x.copy(), req=True)
        y = t.lrelu(v) if leaky else t.relu(v)
        t.backward(t.sum_axis(y, 1), np.ones(1))
        return y.v, v.g
    try:
        Kinks.reset()
        y0, g0 = grad(False)
        assert Kinks.found == [] and np.array_equal(g0, [[1, 0, 1, 0]])
        Kinks.reset(eps=1e-6)
        y1, g1 = grad(False)
        assert [f[1] for f in Kinks.found] == [2, 3] and np.array_equal(g1, g0) and np.array_equal(y1, y0)
        Kinks.reset(eps=1e-6, flip=[1])
        y2, g2 = grad(False)
        assert np.array_equal(g2, [[1, 0, 1, 1]]) and np.array_equal(y2, y0)
        Kinks.reset(eps=1e-6, flip=[0])
        _, g3 = grad(True)
        assert np.allclose(g3, [[1, 0.2, 0.2, 0.2]])
    finally:
        Kinks.reset()


def test_template_images_carry_their_label_and_the_stand_in_classifier_reads_it():
    """data.synthetic_cifar(kind="templates") + eval_cifar.TemplateClassifier: the synthetic stand-in of the end-to-end training runs
    (scripts/train_synthetic.py, profiles/r05_train_*.json).  The classifier is >= 99.5 % correct on the real synthetic images
    and at chance on label-free uniform images; bench.py's images are these."""
    import rcgan_amd  # noqa: F401
    from rcgan_amd import data as D
    from rcgan_amd.eval_cifar import TemplateClassifier, generated_label_accuracy
    x, y = D.synthetic_cifar(2000, 7, "templates")
    assert x.dtype == np.uint8 and x.shape == (2000, 3072)
    imgs = x.reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1)
    clf = TemplateClassifier()
    assert generated_label_accuracy(imgs, y, classifier=clf) >= 0.995
    xu, yu = D.synthetic_cifar(2000, 7)
    acc = generated_label_accuracy(xu.reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1), yu, classifier=clf)
    assert 0.05 <= acc <= 0.15
    import bench
    rs = np.random.RandomState(3)
    lab = rs.randint(10, size=16)
    a = bench.synthetic_images(rs, lab)
    assert np.array_equal(a, D.template_images(_rs_after_labels(3, 16), lab))


def _rs_after_labels(seed, n):
    rs = np.random.RandomState(seed)
    rs.randint(10, size=n)
    return rs


def test_mnist_class_pattern_digits_and_their_stand_in_classifier():
    """data_mnist.synthetic(kind="templates") + eval_mnist.TemplateClassifier: the stand-in for MNIST and for the missing frozen
    classifier (mnist/utils.py:276) in the MNIST training runs; through the reference's own bookkeeping (utils.py:292-305)."""
    import rcgan_amd  # noqa: F401
    from rcgan_amd import data_mnist as DM
    from rcgan_amd.eval_mnist import TemplateClassifier, generated_label_accuracy, template_predict
    X, y = DM.synthetic(3000, 5, "templates")
    assert X.shape == (3000, 28, 28, 1) and 0 <= X.min() and X.max() <= 255
    clf = TemplateClassifier()
    assert (clf(X / 255.) == y).mean() >= 0.995
    Xu, yu = DM.synthetic(3000, 5)
    assert 0.05 <= (clf(Xu / 255.) == yu).mean() <= 0.15
    # [draws, 100, 28, 28, 1] on ten labels per class in class order: accuracy 1 on real images of those classes
    byc = [X[y == c][:20] / 255. for c in range(10)]
    samples = np.stack([np.concatenate([byc[c][10 * d:10 * d + 10] for c in range(10)]) for d in range(2)])
    samples = np.concatenate([samples] * 5)                      # 10 draws -> one batch of 100 per class
    assert generated_label_accuracy("mnist", samples, template_predict) >= 0.99


def test_precision_study_bookkeeping():
    """scripts/precision_study.py: arm specifications, the tail-mean score and the paired differences (no GPU: the runs are child processes)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("precision_study", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                  "scripts", "precision_study.py"))
    ps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ps)
    assert ps.parse_arm("bf16") == ("bf16", "bf16", {})
    assert ps.parse_arm("plain=bf16:RCGAN_UP_PHASE=0:X=1") == ("plain", "bf16", {"RCGAN_UP_PHASE": "0", "X": "1"})
    curve = [{"iteration": 250 * i, "gen_label_acc": i / 20.0} for i in range(1, 21)]
    sc, n = ps.score(curve, 0.4)
    assert n == 8 and abs(sc - np.mean([i / 20.0 for i in range(13, 21)])) < 1e-12
    res = [{"arm": a, "seed": s, "losses_finite": True,
            "curve": [{"iteration": 250 * i, "gen_label_acc": i / 20.0 + 0.01 * s + (0.1 + 0.02 * s if a == "f32" else 0.0)} for i in range(1, 21)]}
           for a in ("bf16", "f32") for s in range(4)]
    per_arm, paired = ps.summarise(res, [("bf16", "bf16", {}), ("f32", "f32", {})], "f32", 0.4)
    assert per_arm["bf16"]["n_seeds"] == 4 and per_arm["f32"]["n_seeds"] == 4
    d = paired["f32_minus_bf16"]
    assert d["n_pairs"] == 4 and abs(d["mean"] - 0.13) < 1e-9 and d["se"] > 0 and abs(d["t"] - d["mean"] / d["se"]) < 1e-9
