import importlib
import os
import sys

import pytest

# torch bundles its own ROCm runtime: let it initialise HIP BEFORE libgl355.so pulls in the system
# libamdhip64, otherwise torch later reports "No HIP GPUs are available" in the same process.
# one hardware queue per prover context for the many-contexts tests (read by the HIP runtime when it starts; the product's own
# way to set it is gl355_runtime_config, which must run before HIP is initialised -- here torch initialises it first)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
try:
    import torch
    torch.cuda.is_available() and torch.cuda.init()
except Exception:  # CPU-only box / torch missing: the CPU tests do not need it
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def gl():
    return importlib.import_module("stark-verifier_amd")


@pytest.fixture(scope="session")
def ctx(gl):
    c = gl.Context(0)
    yield c
    c.close()
