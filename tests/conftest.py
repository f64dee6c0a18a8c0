import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import wno

    wno.build()
    return wno


@pytest.fixture(scope="session")
def gpu(oracle):
    """GPU tier set-up.  torch first: it brings its own copy of the HIP runtime, and a process must not end up with
    two of them (the second one finds no device) -- once torch's is loaded, libwalnuts_hip.so binds to the same."""
    import torch

    assert torch.cuda.is_available(), "GPU tier needs a GPU"
    import walnuts_amd as wa

    wa.load_library()  # the in-tree HIP extension, or raise
    return wa
