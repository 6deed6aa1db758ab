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


def test_zero_rule_records_without_asserting_only_when_the_expectation_vanishes():
    e = np.zeros((8, 8))
    U.check_mean_rel(e + 1e-7, e, "fp16", "t", 1.0, sk=1, oracle=e)
    assert _last_row()["rule"] == "zero"


def test_oracle_rule_allows_twice_the_reference_algorithms_own_error_and_no_more_than_its_cap():
    rng = np.random.default_rng(1)
    e = rng.standard_normal((16, 16))
    orc = e * (1 + 0.02)                                   # the reference algorithm itself is 2 % off on this (tiny-sk) problem
    U.check_mean_rel(e * (1 + 0.035), e, "fp16", "t", 1.0, sk=8, oracle=orc)
    r = _last_row()
    assert r["rule"] == "oracle" and abs(r["bound"] - 0.04) < 1e-3
    with pytest.raises(AssertionError):
        U.check_mean_rel(e * (1 + 0.05), e, "fp16", "t", 1.0, sk=8, oracle=orc)
    # an oracle that is far off may widen the bound only to ORACLE_BOUND_CAP x plain (sk > ORACLE_TINY_SK) ...
    far = e * (1 + 0.2)
    U.check_mean_rel(e * (1 + 0.09), e, "fp16", "t", 1.0, sk=8, oracle=far)
    assert abs(_last_row()["bound"] - U.ORACLE_BOUND_CAP * U.TOL["fp16"]["mean_rel"]) < 1e-12
    # ... and an oracle beyond its own sanity cap fails the test outright instead of loosening anything
    with pytest.raises(AssertionError, match="oracle drift"):
        U.check_mean_rel(e, e, "fp16", "t", 1.0, sk=8, oracle=e * 1.5)


def test_oracle_floor_rule_takes_over_only_for_tiny_key_counts_with_an_ill_conditioned_raw_metric():
    """one query over two keys: a single ~1e-6 expectation with an error of 1e-4 puts the raw relative mean of the ORACLE above its cap"""
    rng = np.random.default_rng(2)
    e = rng.standard_normal((1, 64))
    e[0, 3] = 1e-6
    orc, x = e.copy(), e.copy()
    orc[0, 3] += 1e-4 * 3          # raw relative error 300 on one of 64 elements: mean 4.7 > ORACLE_OWN_CAP x 1e-2
    x[0, 3] += 1e-4 * 4
    U.check_mean_rel(x, e, "fp16", "t", 1.0, sk=2, oracle=orc)
    r = _last_row()
    assert r["rule"] == "oracle-floor" and r["floored"] <= r["bound"] <= 2 * U.ORACLE_OWN_CAP * U.TOL["fp16"]["mean_rel"]
    # the same tensors on a problem with more keys are NOT routed there: the oracle's sanity cap applies
    with pytest.raises(AssertionError, match="oracle drift"):
        U.check_mean_rel(x, e, "fp16", "t", 1.0, sk=16, oracle=orc)


def test_oracle_lazy_rule_consults_the_oracle_only_after_the_plain_bound_failed():
    rng = np.random.default_rng(3)
    e = rng.standard_normal((32, 32))
    calls = []

    def oracle_fn():
        calls.append(1)
        return e * (1 + 0.008)

    U.check_mean_rel(e * (1 + 1e-3), e, "fp16", "t", 1.0, sk=128, oracle=None, oracle_fn=oracle_fn)
    assert _last_row()["rule"] == "plain" and not calls                      # passed the plain bound: the oracle is never run
    U.check_mean_rel(e * (1 + 0.013), e, "fp16", "t", 1.0, sk=128, oracle=None, oracle_fn=oracle_fn)
    assert _last_row()["rule"] == "oracle-lazy" and len(calls) == 1 and abs(_last_row()["bound"] - 0.016) < 1e-3
    with pytest.raises(AssertionError, match="oracle consulted"):
        U.check_mean_rel(e * (1 + 0.03), e, "fp16", "t", 1.0, sk=128, oracle=None, oracle_fn=oracle_fn)


def test_every_tool_script_byte_compiles():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    for f in sorted(os.listdir(root)):
        if f.endswith(".py"):
            py_compile.compile(os.path.join(root, f), doraise=True)
