import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


# PyTorch (used by a few GPU tests and by bench.py for device buffers) must be imported BEFORE
# libfqgpu.so initialises the HIP runtime: registering the wheel's thousands of code objects with an
# already running runtime takes minutes.  Collection happens before any test touches the GPU.
try:  # pragma: no cover - plumbing
    import torch  # noqa: F401
except Exception:  # torch is optional for the CPU suite
    torch = None
