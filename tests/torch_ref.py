"""The PyTorch-CPU second implementation lives in oracle/torch_port.py (it is also bench.py's threaded CPU baseline)."""
from oracle.torch_port import *  # noqa: F401,F403
