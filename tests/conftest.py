import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _cap_threads():
    # the GPU boxes expose 256 logical CPUs; torch-CPU (the oracle) oversubscribes badly with that many
    try:
        import torch
        torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    except Exception:
        pass


_cap_threads()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than ~30 s")


def gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _host_io_library():
    """libsamrs_io.so (g++, ~2 s) is a build artefact: make sure it exists before tests that go through samrs_amd.tile_io
    (a clean checkout has no .so files; the HIP library is built by the tests that need it, tests/test_cabi_symbols.py)."""
    import subprocess
    csrc = os.path.join(ROOT, "samrs_amd", "csrc")
    if not os.path.exists(os.path.join(csrc, "libsamrs_io.so")):
        subprocess.run(["make", "-C", csrc, "libsamrs_io.so"], check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    yield
