"""-m gpu: the reference's property tests (term-guard/tests/property_tests.rs) replayed through ValidationSuite.run
on the HIP path: completeness = 1 - round(n f)/n within 10 eps (:215-255), min / max exact and mean within 1e-4
(:378-480), uniqueness = distinct / total (:704-768).  Seeded generators stand in for proptest."""
import numpy as np
import pyarrow as pa
import pytest

from term_amd.suite import Assertion, Check, CompletenessOptions, Level, ValidationSuite

pytestmark = pytest.mark.gpu


def run(check, table):
    return ValidationSuite.builder("s").check(check.build()).build().run(table)


@pytest.mark.parametrize("seed", range(12))
def test_completeness_threshold_property(seed):
    rng = np.random.default_rng(seed)
    num_rows = int(rng.integers(10, 1000))
    null_fraction, threshold = float(rng.random()), float(rng.random())
    num_nulls = int(round(num_rows * null_fraction))
    mask = np.zeros(num_rows, dtype=bool)
    mask[rng.permutation(num_rows)[:num_nulls]] = True
    tbl = pa.table({"test_column": pa.array(rng.standard_normal(num_rows), pa.float64(), mask=mask)})
    r = run(Check.builder("c").level(Level.ERROR).completeness(["test_column"], CompletenessOptions.threshold(threshold)), tbl)
    expected = 1.0 - num_nulls / num_rows
    assert r.is_success() == (expected >= threshold)
    metric = r.report.metrics.custom_metrics.get("c.completeness", r.report.issues[0].metric if r.report.issues else None)
    assert abs(metric - expected) < 10 * np.finfo(float).eps


@pytest.mark.parametrize("seed", range(12))
def test_statistics_properties(seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(10, 2000))
    vals = rng.uniform(-1e6, 1e6, size=n)
    tbl = pa.table({"v": pa.array(vals, pa.float64())})
    c = (Check.builder("c").level(Level.ERROR)
         .has_min("v", Assertion.Equals(float(vals.min()))).has_max("v", Assertion.Equals(float(vals.max())))
         .has_mean("v", Assertion.Between(float(vals.mean()) - 1e-4, float(vals.mean()) + 1e-4)))
    r = run(c, tbl)
    assert r.is_success(), [i.message for i in r.report.issues]
    m = r.report.metrics.custom_metrics
    assert m["c.min"] == vals.min() and m["c.max"] == vals.max() and abs(m["c.mean"] - vals.mean()) < 1e-4


@pytest.mark.parametrize("seed", range(12))
def test_uniqueness_constraint_property(seed):
    rng = np.random.default_rng(200 + seed)
    total = int(rng.integers(10, 100))
    dup_fraction = float(rng.random())
    num_unique = max(1, int(np.ceil(total * (1.0 - dup_fraction))))
    values = list(range(num_unique)) + [int(x) for x in rng.integers(0, num_unique, size=total - num_unique)]
    rng.shuffle(values)
    tbl = pa.table({"id_column": pa.array(values, pa.int64())})
    r = run(Check.builder("c").level(Level.ERROR).validates_uniqueness(["id_column"], 1.0), tbl)
    actual = num_unique / total
    assert r.is_success() == (actual >= 1.0)
    metric = r.report.metrics.custom_metrics.get("c.full_uniqueness", r.report.issues[0].metric if r.report.issues else None)
    assert abs(metric - actual) < 1e-12
