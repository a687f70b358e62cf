import os
import sys

import pytest

# torch bundles its own HIP runtime: it must be loaded before libgdca.so (which links the system one), otherwise torch.cuda
# finds no device later in the same process (INTEGRATION.md "load order")
import torch  # noqa: E402,F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFDATA = os.path.join(GOLDEN, "reference")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "slow: a heavy case (gigabytes of temporary files or minutes of work); part of the default runs, "
                                       "deselect with -m 'gpu and not slow' / -m 'not gpu and not slow' for a quick pass")


@pytest.fixture(scope="session")
def refdata():
    return REFDATA


@pytest.fixture(scope="session", autouse=True)
def _oracle_c_kernels():
    """Build the oracle's C accelerators once if they are missing (gcc only, no GPU)."""
    so = os.path.join(ROOT, "oracle", "_build", "liboracle_kernels.so")
    if not os.path.exists(so):
        import subprocess

        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    yield
