import os
import sys
import pytest

# torch before the HIP library of this package: both bring a HIP runtime along (torch its bundled one),
# and a process can only use the one that was loaded first -- with torch first they share it
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "spawns: starts child processes of its own (left out when a test re-runs a whole file in a child)")
