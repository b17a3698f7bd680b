"""robust-conditional-gan_amd: MI355X-native engine for the RCGAN generator/discriminator training step.

Import as ``rcgan_amd`` (the directory name carries a hyphen; ``rcgan_amd.py`` at the repo root loads it).
Layout: ``csrc/`` hand-written gfx950 HIP kernels + the C ABI (include/rcgan_hip.h); ``_lib`` ctypes
binding; ``runtime`` stream / arena / parameter slabs; ``ops`` differentiable device ops; ``ops_cifar`` /
``ops_mnist`` the reference's L1 op API; ``cifar`` / ``mnist`` models, losses and step functions.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
