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


@pytest.fixture(scope="session", autouse=True)
def _warm_gpu_stack():
    """On a fresh GPU box the first process that touches ROCm pages the runtime, torch and the kernels in from a cold image,
    which can take a minute or two.  Do that once here, in-process and untimed, so that the bounded subprocesses of the CLI
    tests (timeouts exist to turn a hang into a failure) only ever see a warm machine.  No-op without a GPU."""
    if torch is not None and torch.cuda.is_available():
        import numpy as np
        import vmaf_amd
        frame = np.full((1, 48, 64), 100.0, np.float32)
        with vmaf_amd.Engine(64, 48, max_batch=1) as e:
            e.extract(frame, frame)
        exe = os.path.join(ROOT, "vmaf_amd", "vmafossexec")
        if os.path.exists(exe):
            import subprocess
            subprocess.run([exe], capture_output=True)          # usage + exit: maps the binary and the library
    yield


@pytest.fixture(scope="session")
def model_path():
    return os.path.join(ROOT, "model", "vmaf_v0.6.1.pkl")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
