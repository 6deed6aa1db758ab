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


def pytest_sessionfinish(session, exitstatus):
    """worst RAW parity margins per test family (tests/_util.py MARGINS) -> gpurun_out/parity_margins.json (copied to profiles/)"""
    import json

    import _util as U

    if not U.MARGINS:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    doc = {"what": "worst raw max_abs / mean_abs / mean_rel (reference test_flash_attn.py:51-71 metrics, expectation rounded to the output "
                   "format, NO slack) per test family and tensor; plain_bound_cases = cases (sk >= 64) on which the reference's plain "
                   "bounds max_abs <= 5e-3, mean_abs <= 2e-4 (x8 for bf16) were asserted",
           "bounds": U.TOL, "exit_status": int(exitstatus), "families": U.MARGINS}
    with open(os.path.join(out, "parity_margins.json"), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
