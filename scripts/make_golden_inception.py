#!/usr/bin/env python3
"""tests/golden/ref_inception_score.npz: the reference's own get_inception_probs / preds2score
(cifar10/common/inception/inception_score_.py:50-68) executed on seeded logits.  The module cannot be imported (it builds the
TF-GAN Inception graph at import), so the two function definitions are taken out of its syntax tree and executed with numpy and a
stand-in for the module-level `logits` tensor whose .eval returns the next seeded batch.  Build container only."""
import ast
import os

import numpy as np

REF = "/root/reference/cifar10/common/inception/inception_score_.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_inception_score.npz")


def main():
    tree = ast.parse(open(REF).read())
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("get_inception_probs", "preds2score")]
    assert len(wanted) == 2
    rs = np.random.RandomState(99)
    n, width = 128 * 5 + 37, 1008                      # an incomplete last batch (dropped by the reference), logits wider than 1000
    all_logits = (rs.randn(n, width) * 3.0).astype(np.float32)
    images = rs.uniform(-1, 1, size=(n, 3, 4, 4)).astype(np.float32)
    images[:, 0, 0, 0] = np.arange(n)                  # row id, so that the stand-in can tell which batch it was handed

    class Logits:
        def eval(self, feed):
            (inp,) = feed.values()
            ids = inp[:, 0, 0, 0].astype(int)
            return all_logits[ids]
    env = {"np": np, "BATCH_SIZE": 128, "logits": Logits(), "inception_images": "inception_images"}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), REF, "exec"), env)
    probs = env["get_inception_probs"](images)
    # (the logits are not stored: the test redraws them from the same seeded numpy stream)
    out = {"seed": 99, "n": n, "width": width, "probs_sample": probs[::41, ::97].astype(np.float64), "n_probs": probs.shape[0]}
    for splits in (1, 3, 10):
        out["score_splits%d" % splits] = np.array(env["preds2score"](probs, splits), dtype=np.float64)
    np.savez_compressed(OUT, **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()}, "%.1f kB" % (os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
