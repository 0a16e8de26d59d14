import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spe_amd import lib
    lib.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _default_precision():
    """Every test starts from (and leaves behind) the library's default precision mode, bf16s - whatever mode it switched to."""
    yield
    if "spe_amd.kernels" in sys.modules:
        sys.modules["spe_amd.kernels"].set_precision("bf16s")
