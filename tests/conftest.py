import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def oracle_omp():
    """The OpenMP build of the oracle: identical integer outputs, parameter gradients summed in double."""
    from oracle.oracle import Oracle
    return Oracle(omp=True)


@pytest.fixture(scope="session")
def mc():
    """The product op surface; importing it on a box without the built HIP library must fail loudly."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mccnn_amd import build
    build.build()   # (no-op when the library and the extension are newer than their sources)
    import mccnn_amd.MCConvModule as M
    return M
