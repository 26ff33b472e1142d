import os
import sys

import pytest

# torch bundles its own ROCm runtime (same SONAME as the system one): when a process uses both torch and
# libvmaf_amd.so, torch has to be loaded FIRST so that the two share one HIP/HSA runtime.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model_path():
    return os.path.join(ROOT, "model", "vmaf_v0.6.1.pkl")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
