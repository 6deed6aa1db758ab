"""Committed evidence must not rot quietly: checks on the newest profiles/rNN_* summaries (CPU, no GPU needed)."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# reference-grid cases (of 4104 per tensor) on which the reference's PLAIN bounds were asserted in the round-3 run ("keep that count
# from shrinking"): a change to tests/_util.py that moves cases from the plain rule to a softer one shows up here
PLAIN_FLOOR = {"O/fp16": 3265, "dK/fp16": 2665, "dQ/fp16": 3347, "dV/fp16": 2014}


def _newest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=lambda p: int(re.search(r"r(\d+)_", os.path.basename(p)).group(1)))
    assert files, pattern
    return files[-1]


def test_plain_rule_case_count_has_not_shrunk():
    m = json.load(open(_newest("r*_parity_margins.json")))
    assert m["exit_status"] == 0
    got = {}
    for fam, tensors in m["families"].items():
        if "reference_grid" in fam:
            for t, d in tensors.items():
                got[t] = got.get(t, 0) + d.get("plain_bound_cases", 0)
    for t, floor in PLAIN_FLOOR.items():
        assert got.get(t, 0) >= floor, (t, got.get(t), floor)


def test_bench_line_and_profiler_agree_on_the_roofline_fraction():
    """`frac` must be reproducible from the line alone and from the committed rocprofv3 table within 1 %"""
    s = json.load(open(_newest("r*_evidence_summary.json")))
    b = json.load(open(_newest("r*_bench_stdout.json")))
    r = b["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    if r.get("frac_from_ms_per_step") is not None:                     # (round 4 on: the timed region IS the kernel's launches)
        assert abs(r["frac"] - r["frac_from_ms_per_step"]) / r["frac"] < 0.01
        assert abs(r["frac"] - b["value"] / r["peak"]) / r["frac"] < 0.01
    assert s["frac_bench_vs_rocprof_relative_difference"] < 0.01, s["frac_bench_vs_rocprof_relative_difference"]
    assert s["library_source_digest"] == r["library_source_digest"]
