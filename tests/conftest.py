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


_GPU_TESTS_RAN = {"n": 0}


def pytest_runtest_logreport(report):
    if report.when == "call" and "gpu" in report.keywords:
        _GPU_TESTS_RAN["n"] += 1


def pytest_sessionfinish(session, exitstatus):
    """worst RAW parity margins per test family (tests/_util.py MARGINS) and every mean_rel decision -> gpurun_out/parity_margins_<stamp>.json /
    mean_rel_table_<stamp>.json (tools/round_evidence_collect.py copies the newest pair to profiles/).  Written only by runs in which at least one
    `gpu`-marked test ran (a CPU-only `pytest -m "not gpu"` used to overwrite the GPU run's tables with the handful of rows the CPU tests of
    the tolerance rules produce), and to run-stamped files, so that no run overwrites another's."""
    import json
    import time

    import _util as U

    if not U.MARGINS or _GPU_TESTS_RAN["n"] == 0:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    stamp = time.strftime("%Y%m%d_%H%M%S") + f"_{os.getpid()}"
    if U.REL_TABLE:
        # every mean_rel decision of the run (tests/_util.py:check_mean_rel): kernel's and oracle's raw metric, the asserted quantity, its bound, the rule
        rules, fams = {}, {}
        for r in U.REL_TABLE:
            s = rules.setdefault(r["rule"], dict(cases=0, worst_kernel=0.0, worst_ratio_to_bound=0.0, unasserted=0))
            s["cases"] += 1
            s["worst_kernel"] = max(s["worst_kernel"], r["kernel"])
            if r.get("bound") and r.get("asserted") is not None:
                s["worst_ratio_to_bound"] = max(s["worst_ratio_to_bound"], r["asserted"] / r["bound"])
            else:
                s["unasserted"] += 1
            f = fams.setdefault(r["family"].split("[")[0], dict(cases=0, over_plain_bound=0, zero=0))
            f["cases"] += 1
            f["over_plain_bound"] += int(r["kernel"] > U.TOL[r["dtype"]]["mean_rel"])
            f["zero"] += int(r["rule"] == "zero")
        over = [r for r in U.REL_TABLE if r["kernel"] > U.TOL[r["dtype"]]["mean_rel"]]
        with open(os.path.join(out, f"mean_rel_table_{stamp}.json"), "w") as f:
            json.dump({"what": "raw mean_rel = mean(|x - e| / max(|e|, 1e-6)) (reference test_flash_attn.py:51-71,117,412) per asserted tensor; `asserted` is the quantity "
                               "the rule bounds (tests/_util.py:check_mean_rel); `over_plain_bound` lists EVERY case whose raw kernel value exceeds the plain bound, with "
                               "the oracle's value and the rule that applied",
                       "exit_status": int(exitstatus), "gpu_tests_run": _GPU_TESTS_RAN["n"], "n_cases": len(U.REL_TABLE), "unasserted_rows": sum(d["unasserted"] for d in rules.values()),
                       "by_rule": rules, "by_family": fams, "over_plain_bound": over}, f, indent=1)
    doc = {"what": "worst raw max_abs / mean_abs / mean_rel (reference test_flash_attn.py:51-71 metrics, expectation rounded to the output "
                   "format, NO slack) per test family and tensor; plain_bound_cases = cases (sk >= 64) on which the reference's plain "
                   "bounds max_abs <= 5e-3, mean_abs <= 2e-4 (x8 for bf16) were asserted",
           "bounds": U.TOL, "exit_status": int(exitstatus), "gpu_tests_run": _GPU_TESTS_RAN["n"], "families": U.MARGINS}
    with open(os.path.join(out, f"parity_margins_{stamp}.json"), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
