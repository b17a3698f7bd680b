"""CPU oracle for the RCGAN G/D training-step hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-numpy restatement of the
arithmetic the reference (tkkiran/Robust-Conditional-GAN, TensorFlow 1.5 graph
code) executes for one optimiser step.  It exists so that the HIP product path
can be checked against something independent.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product package never does.

Pinning status
--------------
* label-corruption indexing (``oracle.labels``): PINNED, bit-exact against
  vectors captured by importing the reference's own numpy code
  (``cifar10/common/data/cifar10.py:19-45`` and ``mnist/model.py:770-834``);
  fixtures under ``tests/golden/`` with ``scripts/make_golden_labels.py``.
* every floating-point function (conv, BN, SN, losses, Adam): PARITY UNPINNED.
  The arithmetic lives in TensorFlow 1.5 (un-vendored, prose pin only,
  reference ``README.md:88``), which is not installable in this image, and the
  reference holds no tests or golden vectors.  The restatement follows the
  TF-1.5 semantics listed in SURVEY.md Appendix C and is cross-checked against
  an independent second implementation (PyTorch-CPU autograd) in
  ``tests/test_oracle_vs_torch.py``.
"""
