#!/usr/bin/env python3
"""Repeat the capture-abort scenario of tests/test_gpu_dp.py (world 2 against the test double whose all-reduce cannot be captured) in ONE
process and report every run whose losses differ from the single-rank run's, with what else differs."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_dp as T  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(14)
B = 8
its = T._feeds(rs, B, 2, "rcgan")
m = T._model("rcgan", "bf16", B, world_size=1, comm=None)
ref = T._run_iterations(m, its)
m.ctx.close()
if not (os.environ.get("DBG_EAGER1") or os.environ.get("DBG_EAGER2") or os.environ.get("DBG_GRAPH2")):      # DBG_GRAPH2: the captured data-parallel steps, nothing injected
    os.environ["RCGAN_COMM_STUB_FAIL_IN_CAPTURE"] = "1"
bad = 0
for r in range(runs):
    if os.environ.get("DBG_EAGER1"):       # single rank, no graphs at all: is the eager path itself at fault?
        import rcgan_amd  # noqa: F401
        from rcgan_amd.cifar import CifarRCGAN
        m = CifarRCGAN(algorithm="rcgan", alpha=0.6, batch_size=B, dtype="bf16", seed=3, perm_classifier=False, confuse_init=False,
                       use_graphs=False, device_rng=True, arena_bytes=2 << 30)
    elif os.environ.get("DBG_EAGER2"):     # the test double, no graphs at all (every step eager, collectives included)
        import rcgan_amd  # noqa: F401
        from rcgan_amd.cifar import CifarRCGAN
        m = CifarRCGAN(algorithm="rcgan", alpha=0.6, batch_size=B, dtype="bf16", seed=3, perm_classifier=False, confuse_init=False,
                       use_graphs=False, device_rng=True, arena_bytes=2 << 30, world_size=2, comm="stub")
    else:
        m = T._model("rcgan", "bf16", B, world_size=2, comm="stub")
    l1 = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if os.environ.get("DBG_STEPWISE"):
            # the production loop with a finiteness check of both parameter groups after every step (synchronises: changes the timing)
            first_bad = None
            for it, (lra, ds, g) in enumerate(its):
                m.set_feed("gf", m.pack_feed("gf", labels_random_all=lra))
                m.prepare_critic_fakes()
                for k, d in enumerate(ds):
                    m.set_feed("d", m.pack_feed("d", **d))
                    m.d_step(iteration=it)
                    if first_bad is None and not all(np.isfinite(v).all() for v in m.get_params().values()):
                        first_bad = ("d_step", it, k, m.losses())
                m.set_feed("g", m.pack_feed("g", **g))
                m.g_step(iteration=it + 1)
                if first_bad is None and not all(np.isfinite(v).all() for v in m.get_params().values()):
                    first_bad = ("g_step", it, m.losses())
            m.ctx.sync()
            out = (m.get_params(), m.losses(), m.get_state())
            if first_bad:
                print("run %d: first non-finite parameters after %s" % (r, first_bad))
        elif os.environ.get("DBG_KEEP_FEEDS"):
            keep = []          # every packed feed stays referenced until the run has synchronised
            for it, (lra, ds, g) in enumerate(its):
                keep.append(m.pack_feed("gf", labels_random_all=lra)); m.set_feed("gf", keep[-1])
                m.prepare_critic_fakes()
                for k, d in enumerate(ds):
                    keep.append(m.pack_feed("d", **d)); m.set_feed("d", keep[-1])
                    m.d_step(iteration=it)
                keep.append(m.pack_feed("g", **g)); m.set_feed("g", keep[-1])
                m.g_step(iteration=it + 1)
            m.ctx.sync()
            out = (m.get_params(), m.losses(), m.get_state())
        else:
            out = T._run_iterations(m, its, losses_after_first=l1)
    kinds = {k: (g is None) for k, g in m._graphs.items()}
    m.ctx.close()
    if out[1] != ref[1]:
        bad += 1
        diffp = [k for k in ref[0] if not np.array_equal(ref[0][k], out[0][k])]
        nonfin = [k for k in out[0] if not np.isfinite(out[0][k]).all()]
        print("run %d: losses %s (ref %s), after first iteration %s; %d of %d parameters differ, %d non-finite; graphs %s" %
              (r, out[1], ref[1], l1, len(diffp), len(ref[0]), len(nonfin), kinds))
        for k in diffp[:6]:
            print("    %s max|d| %.3e  |ref|max %.3e" % (k, float(np.abs(ref[0][k].astype(np.float64) - out[0][k]).max()), float(np.abs(ref[0][k]).max())))
print("%d of %d runs differ" % (bad, runs))
