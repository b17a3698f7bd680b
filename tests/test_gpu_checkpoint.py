"""Device-side checkpoint round trip (tf.train.Saver at cifar10/gan_resnet.py:906-914,1007-1013): train two
iterations, write a TensorFlow-V2 bundle with host.Saver, restore it into a FRESH model built from a different seed, and
run a third iteration on both: variables, Adam slots, spectral-norm ``u`` vectors and losses must be bit-identical.
(The bundle FORMAT is checked against the published layout by tests/test_tf_bundle_cpu.py; nothing TensorFlow wrote exists
here to read, so interop with TensorFlow itself stays unpinned -- README "blocked on artefacts".)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _iteration(m, rs, B, it):
    """gan_resnet.py:928-947 on explicit host draws (device_rng off: the restored model must see the same z / noise)."""
    from rcgan_amd.cifar import N_CRITIC
    if it > 0:
        m.set_inputs(labels_random_G=rs.randint(10, size=2 * B), labels_biased_G=rs.randint(10, size=2 * B),
                     z_G=rs.randn(2 * B, 128).astype(np.float32))
        m.g_step(iteration=it)
    Cinv = np.linalg.inv(__import__("oracle.cifar", fromlist=["c_alpha"]).c_alpha(0.6))
    for _ in range(N_CRITIC):
        lab = rs.randint(10, size=B)
        bia = rs.randint(10, size=B)
        m.set_inputs(images=rs.randint(0, 256, size=(B, 3072)), noise=rs.uniform(0, 1 / 128., size=(B, 3072)).astype(np.float32),
                     labels=lab, labels_random=rs.randint(10, size=B), labels_biased=bia, inv_weights=Cinv[lab].astype(np.float32),
                     z=rs.randn(B, 128).astype(np.float32), labels_all=np.concatenate([lab, bia]))
        m.d_step(iteration=it)


@pytest.mark.parametrize("alg,dtype,graphs", [("rcgan", "bf16", True), ("rcgan-u", "bf16", False), ("rcgan", "f32", False)])
def test_checkpoint_round_trip_resumes_bit_identically(tmp_path, alg, dtype, graphs):
    import rcgan_amd  # noqa: F401
    from rcgan_amd.cifar import CifarRCGAN
    from rcgan_amd.host import Saver, latest_checkpoint, load_checkpoint
    B = 8
    kw = dict(algorithm=alg, alpha=0.6, batch_size=B, dtype=dtype, perm_classifier=(alg == "rcgan-u"), confuse_init=(alg == "rcgan-u"),
              use_graphs=graphs, device_rng=False, arena_bytes=2 << 30)
    a = CifarRCGAN(seed=0, **kw)
    b = None
    try:
        rs = np.random.RandomState(5)
        _iteration(a, rs, B, 0)
        _iteration(a, rs, B, 1)
        a.iteration = 2
        saver = Saver(max_to_keep=5)
        d = str(tmp_path / "checkpoint")
        path = saver.save(a.state_dict(), d, "model.ckpt", 2)
        assert latest_checkpoint(d) == path
        sd = load_checkpoint(path)
        # what the bundle holds: every variable with both Adam slots, the u vectors, the optimisers' beta powers
        for grp in a.groups:
            for n in grp.names:
                assert n in sd and n + "/Adam" in sd and n + "/Adam_1" in sd, n
        assert all(k in sd for k in a.state) and "beta2_power" in sd and "beta2_power_1" in sd
        b = CifarRCGAN(seed=123, **kw)                       # different initial values everywhere
        n0 = a.groups[0].names[0]
        assert not np.array_equal(b.groups[0].get(n0), a.groups[0].get(n0))
        b.load_state_dict(sd)
        assert b.iteration == 2
        state = rs.get_state()
        _iteration(a, rs, B, 2)
        rs.set_state(state)
        _iteration(b, rs, B, 2)
        for ga, gb in zip(a.groups, b.groups):
            assert ga.steps_applied() == gb.steps_applied()
            for n in ga.names:
                for slot in (None, "m", "v"):
                    x, y = (ga.get(n), gb.get(n)) if slot is None else (ga.get(n, slot), gb.get(n, slot))
                    assert np.array_equal(x, y), (n, slot, float(np.abs(x - y).max()))
        ua, ub = a.get_state(), b.get_state()
        for k in ua:
            assert np.array_equal(ua[k], ub[k]), k
        assert a.losses() == b.losses()
    finally:
        a.ctx.close()
        if b is not None:
            b.ctx.close()
