import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; never imported by the product package)."""
    from oracle import oracle as orc
    orc.load()
    return orc


@pytest.fixture(scope="session")
def ndev():
    """rt_init() on the GPU box: the device count (GPU tests only; raises without a HIP device)."""
    import ray_tracer_s8_amd as rt
    return rt.init()
