#!/usr/bin/env python3
"""Drop-in for the reference's mnist/main.py: same flags, same output tree, MI355X engine underneath."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rcgan_amd  # noqa: E402,F401
from rcgan_amd.train_mnist import main  # noqa: E402

if __name__ == "__main__":
    main()
