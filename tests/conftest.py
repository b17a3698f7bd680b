import os
import sys

import pytest

# the two-level ("tree") batch-norm reduction only takes tensors of >= 48M elements in production; the parity tests run it
# from 4M elements on (read once by the library, so it has to be set before the first call)
os.environ.setdefault("RCGAN_BN_TREE_MIN", str(4 << 20))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
