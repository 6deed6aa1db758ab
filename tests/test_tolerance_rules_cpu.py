"""The tolerance rules of tests/_util.py are test infrastructure the whole parity claim rests on: each rule of check_mean_rel exercised on small synthetic
tensors (CPU, numpy only) - which rule fires, what it asserts, and that the loosening ones stay inside their caps."""
import os
import py_compile

import numpy as np
import pytest

import _util as U


def _last_row():
    return U.REL_TABLE[-1]


def test_plain_rule_asserts_the_references_bound_on_long_key_axes():
    rng = np.random.default_rng(0)
    e = rng.standard_normal((64, 32))
    U.check_mean_rel(e * (1 + 1e-3), e, "fp16", "t", 1.0, sk=128, oracle=None)
    assert _last_row()["rule"] == "plain" and _last_row()["bound"] == U.TOL["fp16"]["mean_rel"]
    with pytest.raises(AssertionError, match="PLAIN"):
        U.check_mean_rel(e * 1.05, e, "fp16", "t", 1.0, sk=128, oracle=None)


def test_zero_rule_asserts_the_kernels_own_magnitude_where_the_expectation_vanishes():
    e = np.zeros((8, 8))
    U.check_mean_rel(e + 1e-7, e, "fp16", "t", 1.0, sk=1, oracle=e)
    r = _last_row()
    assert r["rule"] == "zero" and r["bound"] == U.ZERO_ABS_TOL and r["asserted"] == pytest.approx(1e-7)
    with pytest.raises(AssertionError, match="vanishes identically"):
        U.check_mean_rel(e + 3 * U.ZERO_ABS_TOL, e, "fp16", "t", 1.0, sk=1, oracle=e)       # noise far above fp32 summation-order level is a failure now
    nz = np.full((8, 8), 1e-3)
    U.check_mean_rel(nz, nz, "fp16", "t", 1.0, sk=128, oracle=None)                          # a tensor that does NOT vanish never lands here
    assert _last_row()["rule"] == "plain"


def test_oracle_rule_allows_twice_the_reference_algorithms_own_error_and_no_more_than_its_cap():
    rng = np.random.default_rng(1)
    e = rng.standard_normal((16, 16))
    orc = e * (1 + 0.02)                                   # the reference algorithm itself is 2 % off on this (few-key) problem
    U.check_mean_rel(e * (1 + 0.035), e, "fp16", "t", 1.0, sk=8, oracle=orc)
    r = _last_row()
    assert r["rule"] == "oracle" and abs(r["bound"] - 0.04) < 1e-3 and not r["floored_denominator"] and abs(r["asserted"] - 0.035) < 1e-3
    with pytest.raises(AssertionError):
        U.check_mean_rel(e * (1 + 0.05), e, "fp16", "t", 1.0, sk=8, oracle=orc)
    # an oracle that is far off may widen the bound only to ORACLE_BOUND_CAP x plain (sk > ORACLE_TINY_SK) ...
    far = e * (1 + 0.2)
    U.check_mean_rel(e * (1 + 0.09), e, "fp16", "t", 1.0, sk=8, oracle=far)
    assert abs(_last_row()["bound"] - U.ORACLE_BOUND_CAP * U.TOL["fp16"]["mean_rel"]) < 1e-12
    # ... to ORACLE_OWN_CAP x plain on at most ORACLE_TINY_SK keys (round 5: twice that) ...
    U.check_mean_rel(e * (1 + 0.24), e, "fp16", "t", 1.0, sk=U.ORACLE_TINY_SK, oracle=far)
    assert abs(_last_row()["bound"] - U.ORACLE_OWN_CAP * U.TOL["fp16"]["mean_rel"]) < 1e-12 and _last_row()["floored_denominator"]
    with pytest.raises(AssertionError):
        U.check_mean_rel(e * (1 + 0.3), e, "fp16", "t", 1.0, sk=U.ORACLE_TINY_SK, oracle=far)
    # ... and an oracle beyond its own sanity cap fails the test outright instead of loosening anything
    with pytest.raises(AssertionError, match="oracle drift"):
        U.check_mean_rel(e, e, "fp16", "t", 1.0, sk=8, oracle=e * 1.5)


def test_oracle_rule_floors_the_denominators_on_tiny_key_counts_only():
    """one query over two keys: a single ~1e-6 expectation with an error of 1e-4 puts the RAW relative mean of the reference algorithm itself above any cap;
    on at most ORACLE_TINY_SK keys both sides are measured against max(|e|, 1 % RMS) - the same rule, not a fallback (round 5 had a rule of its own for it)"""
    rng = np.random.default_rng(2)
    e = rng.standard_normal((1, 64))
    e[0, 3] = 1e-6
    orc, x = e.copy(), e.copy()
    orc[0, 3] += 1e-4 * 3          # raw relative error 300 on one of 64 elements: raw mean 4.7
    x[0, 3] += 1e-4 * 4
    U.check_mean_rel(x, e, "fp16", "t", 1.0, sk=2, oracle=orc)
    r = _last_row()
    assert r["rule"] == "oracle" and r["floored_denominator"] and r["asserted"] <= r["bound"] <= U.ORACLE_OWN_CAP * U.TOL["fp16"]["mean_rel"]
    assert r["oracle"] > 4.0 and r["oracle_asserted"] < 1e-3          # the raw value stays on record, the floored one is what is asserted
    # the same tensors on a problem with more keys are measured raw: the oracle's sanity cap applies
    with pytest.raises(AssertionError, match="oracle drift"):
        U.check_mean_rel(x, e, "fp16", "t", 1.0, sk=16, oracle=orc)


def test_there_are_four_rules_and_every_one_asserts():
    """at most four rules in check_mean_rel, no row without an asserted quantity and a bound"""
    import inspect
    import re

    src = inspect.getsource(U.check_mean_rel)
    assert sorted(set(re.findall(r'rule="([a-z-]+)"', src))) == ["floor", "oracle", "plain", "zero"]
    n0 = len(U.REL_TABLE)
    rng = np.random.default_rng(5)
    e = rng.standard_normal((32, 32))
    U.check_mean_rel(e, e, "fp16", "t", 1.0, sk=128, oracle=None)
    U.check_mean_rel(e, e, "fp16", "t", 1.0, sk=8, oracle=None)
    U.check_mean_rel(e, e, "fp16", "t", 1.0, sk=8, oracle=e)
    U.check_mean_rel(e * 0, e * 0, "fp16", "t", 1.0, sk=1, oracle=None)
    rows = U.REL_TABLE[n0:]
    assert [r["rule"] for r in rows] == ["plain", "floor", "oracle", "zero"]
    assert all(r["asserted"] is not None and r["bound"] for r in rows)


def test_every_tool_script_byte_compiles():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    for f in sorted(os.listdir(root)):
        if f.endswith(".py"):
            py_compile.compile(os.path.join(root, f), doraise=True)
