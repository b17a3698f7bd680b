"""The product's kernels against vectors produced by running the reference's own program (tests/golden/ref_cifar_<alg>.npz, see
tests/test_reference_golden.py and scripts/make_golden_reference.py): the first critic run and the generator-cost log fetch of
the recorded session.run -- same seeded initial weights (bit-exact for every numpy-initialised tensor), same fed batch, same
random draws.  The reference splits its batch of 4 into two towers of 2 with their own batch statistics and averages the tower
costs (gan_resnet.py:186-188,529-546,697): the product evaluates each tower as one rank's step (fp32 activations, through the
C ABI) and the test averages what the all-reduce would.

Tolerances: cost 2e-5; gradients 1e-2 norm-relative on the recorded strided samples (batch norm over two samples is badly
conditioned in fp32: the fp32 oracle sits 2e-4..4e-3 from float64 on such gradients, tests/test_gpu_cifar_step.py)."""
import os

import numpy as np
import pytest

from oracle import cifar as oc

pytestmark = pytest.mark.gpu
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


@pytest.mark.parametrize("tag,alg,flags", CASES)
def test_first_critic_run_of_the_reference(tag, alg, flags):
    import rcgan_amd  # noqa: F401
    from rcgan_amd.cifar import CifarRCGAN, create_variables
    z = np.load(os.path.join(GOLDEN, "ref_cifar_%s.npz" % tag))
    B, ntow, seed = int(z["batch_size"]), int(z["towers"]), int(z["seed"])
    Bt = B // ntow
    perm, cinit = flags.get("perm_classifier", False), flags.get("confuse_init", False)
    p = "run00/"
    noise, zs = _draws(z, p)
    costs, gsum = [], None
    for t in range(ntow):
        gs, ds, cs, U = create_variables(seed, alg, perm, "linear", cinit, 0.2)
        U = {n: z["init/" + n].reshape(U[n].shape).astype(np.float32) for n in U}      # what TensorFlow's generator drew in the recorded run
        m = CifarRCGAN(algorithm=alg, alpha=0.6, batch_size=Bt, dtype="f32", perm_classifier=perm, perm_multiplier=1.0, confuse_init=cinit,
                       use_graphs=False, device_rng=False, variables=(gs, ds, cs, U), arena_bytes=1 << 30)
        try:
            sl = slice(t * Bt, (t + 1) * Bt)
            lab, lr_, lb = z[p + "feed/labels"][sl], z[p + "feed/labels_random"][sl], z[p + "feed/labels_biased"][sl]
            second = lr_ if alg in ("biased", "unbiased") else lb
            m.set_inputs(images=z[p + "feed/images"][sl], noise=noise[sl], labels=lab, labels_random=lr_, labels_biased=lb,
                         inv_weights=z[p + "feed/inv_weights"][sl].astype(np.float32), z=zs[t].astype(np.float32),
                         labels_all=np.concatenate([lab, second]))
            m.d_step(iteration=0)
            costs.append(m.losses()[0])
            g = m.get_grads(m.PD)
            gsum = g if gsum is None else {k: gsum[k] + g[k] for k in g}
        finally:
            m.ctx.close()
    cost = float(np.mean(costs))
    assert abs(cost - float(z[p + "fetched"][0])) <= 2e-5 * max(1.0, abs(cost)), (cost, float(z[p + "fetched"][0]), costs)
    gmax = max(float(z[k]) for k in z.files if k.startswith(p + "grad_norm/"))
    checked = 0
    for k in [k for k in z.files if k.startswith(p + "grad/")]:
        name = k[len(p + "grad/"):]
        ref = z[k].astype(np.float64)
        got = sample(gsum[name] / ntow).astype(np.float64)
        nref = float(np.linalg.norm(ref))
        if float(z[p + "grad_norm/" + name]) <= 1e-3 * gmax or nref == 0.0:
            continue          # ~0 true gradients (conv biases in front of nothing that sees them)
        err = float(np.linalg.norm(got - ref)) / nref
        assert err <= 1e-2, (name, err)
        checked += 1
    assert checked >= 30, checked
