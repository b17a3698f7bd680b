#!/usr/bin/env python3
"""Wall time of one MNIST iteration (1 D update + 2 G updates, hipGraph replay) -- BASELINE cfg2: B=256, fp32, SN projection D."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import rcgan_amd  # noqa: E402,F401
from rcgan_amd.mnist import MnistRCGAN  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
    m = MnistRCGAN(algorithm="rcgan", alpha=0.3, batch_size=B, dtype=dtype, disc_type="projection", estimate_confuse=False)
    rs = np.random.RandomState(0)
    eye = np.eye(10, dtype=np.float32)
    m.set_inputs(images=rs.rand(B, 28, 28, 1).astype(np.float32), z=rs.uniform(-1, 1, size=(B, 100)).astype(np.float32),
                 y_real=eye[rs.randint(10, size=B)], y_gen=eye[rs.randint(10, size=B)], y_fake=eye[rs.randint(10, size=B)],
                 y_real_weights=rs.randn(B, 10).astype(np.float32))
    for _ in range(5):
        m.iteration()
    torch.cuda.synchronize()
    t0 = time.time()
    reps = 50
    for _ in range(reps):
        m.iteration()
    torch.cuda.synchronize()
    ms = (time.time() - t0) * 1e3 / reps
    print("MNIST RCGAN B=%d %s: %.3f ms / iteration (1 D + 2 G updates) = %.0f images/s" % (B, dtype, ms, B / ms * 1e3))


if __name__ == "__main__":
    main()
