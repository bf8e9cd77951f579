import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale and hipcc is present) and load libcortex_hip.so."""
    import torch

    if torch.cuda.is_available():
        torch.cuda.init()  # torch's HIP context first, in the main thread (tests also hand torch tensors to the library)
    import cortex.jl_amd as cx
    from cortex.jl_amd import build as B

    if B.is_stale():
        B.build()
    return cx._lib.load()
