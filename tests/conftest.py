import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "flash-attention-turing_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run the HIP path: no GPU or no compiled extension is a FAILURE, never a skip."""
    import torch

    assert torch.cuda.is_available(), "test marked gpu but no ROCm device is visible"
    import flash_attn_turing  # noqa: F401  (raises ImportError if the HIP build is missing)

    return torch.device("cuda:0")
