import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Load PyTorch (the oracle for the demixer uses it on the CPU) BEFORE libcaviar_hip.so: torch brings its own copy of the HIP
# runtime, and a process must end up with ONE runtime -- whichever library is loaded first decides.
import torch  # noqa: E402,F401


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
