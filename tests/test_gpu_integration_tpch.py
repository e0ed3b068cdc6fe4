"""-m gpu: the reference's integration tests on its TPC-H-like CUSTOMER table, restated from the closed-form
generator (term-guard/src/test_utils.rs:221-296, 1000 rows at SF0.1) and run through ValidationSuite.run on the HIP
path: completeness_integration.rs:10-63, distinctness_integration.rs:6-31, string_length_integration.rs:5-55,
approx_count_distinct / analysis-runner style metrics."""
import pyarrow as pa
import pytest

from term_amd import suite as S
from term_amd.suite import Assertion, Check, CompletenessOptions, Level, LogicalOperator, ValidationSuite

pytestmark = pytest.mark.gpu

SEGMENTS = ["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"]


def customer(rows=1000):
    r = range(1, rows + 1)
    return pa.table({
        "c_custkey": pa.array(list(r), pa.int64()),
        "c_name": pa.array(["Customer#%09d" % i for i in r], pa.string()),
        "c_address": pa.array(["Address %d" % (i % 100) for i in r], pa.string()),
        "c_nationkey": pa.array([i % 25 for i in r], pa.int64()),
        "c_phone": pa.array(["%d-%03d-%03d-%04d" % (10 + i % 25, i % 1000, (i * 7) % 1000, (i * 13) % 10000) for i in r]),
        "c_acctbal": pa.array([((i * 31) % 10000) / 100.0 for i in r], pa.float64()),
        "c_mktsegment": pa.array([SEGMENTS[i % 5] for i in r], pa.string()),
        "c_comment": pa.array([None if i % 10 == 0 else "Customer comment %d" % i for i in r], pa.string()),
    })


def test_completeness_integration():
    tbl = customer()
    s = (ValidationSuite.builder("single_column_completeness").check(
        Check.builder("customer_key_completeness").level(Level.ERROR)
        .completeness(["c_custkey"]).completeness(["c_name"], CompletenessOptions.threshold(0.99)).build()).build())
    assert s.run(tbl).is_success()
    s = (ValidationSuite.builder("backward_compatibility").check(
        Check.builder("old_api_test").level(Level.WARNING)
        .completeness(["c_custkey"]).completeness(["c_name"], CompletenessOptions.threshold(0.95))
        .completeness(["c_custkey", "c_name"], CompletenessOptions(1.0, LogicalOperator.All))
        .completeness(["c_phone", "c_comment"], CompletenessOptions(1.0, LogicalOperator.Any)).build()).build())
    r = s.run(tbl)
    assert r.is_success() and r.report.metrics.failed_checks == 0


def test_distinctness_integration():
    s = (ValidationSuite.builder("customer_distinctness")
         .check(Check.builder("customer_key_distinctness").level(Level.ERROR)
                .validates_distinctness(["c_custkey"], Assertion.GreaterThan(0.99)).build())
         .check(Check.builder("customer_segment_distinctness").level(Level.WARNING)
                .validates_distinctness(["c_mktsegment"], Assertion.LessThan(0.1)).build()).build())
    r = s.run(customer())
    assert r.is_success()
    m = r.report.metrics.custom_metrics
    assert m["customer_key_distinctness.distinctness"] == 1.0 and m["customer_segment_distinctness.distinctness"] == 0.005


def test_string_length_integration():
    s = (ValidationSuite.builder("string_length_validation")
         .check(Check.builder("customer_name_length").level(Level.ERROR)
                .has_min_length("c_name", 5).has_max_length("c_name", 25).build())
         .check(Check.builder("customer_address_length").level(Level.WARNING)
                .has_min_length("c_address", 10).has_max_length("c_address", 40).build())
         .check(Check.builder("customer_phone_length").level(Level.ERROR)
                .has_min_length("c_phone", 15).has_max_length("c_phone", 15).build()).build())
    r = s.run(customer())
    # "Address 0".."Address 9" are 9 characters: the Warning-level minimum fails, the suite still succeeds
    assert r.is_success() and r.report.metrics.failed_checks == 1
    issue = r.report.issues[0]
    assert issue.check_name == "customer_address_length" and issue.metric == 0.9
    assert issue.message == "Length constraint failed: 90.00% of values are at least 10 characters"


def test_mixed_suite_one_pass_and_analyzers():
    tbl = customer()
    s = (ValidationSuite.builder("customer_quality").check(
        Check.builder("all").level(Level.ERROR)
        .validates_primary_key(["c_custkey"]).validates_uniqueness(["c_custkey", "c_nationkey"], 1.0)
        .is_contained_in("c_mktsegment", SEGMENTS)
        .validates_regex("c_phone", r"^\d{2}-\d{3}-\d{3}-\d{4}$", 1.0)
        .has_min("c_acctbal", Assertion.GreaterThanOrEqual(0.0)).has_max("c_acctbal", Assertion.LessThan(100.0))
        .completeness(["c_comment"], CompletenessOptions.threshold(0.9)).build()).build())
    r = s.run(tbl)
    assert r.is_success(), [i.message for i in r.report.issues]
    ctx = (S.AnalysisRunner().add(S.SizeAnalyzer()).add(S.CompletenessAnalyzer("c_comment"))
           .add(S.ApproxCountDistinctAnalyzer("c_mktsegment")).add(S.DistinctnessAnalyzer("c_nationkey"))
           .add(S.MeanAnalyzer("c_acctbal")).add(S.MaxAnalyzer("c_custkey")).run(tbl))
    assert not ctx.has_errors()
    assert ctx.get_metric("size")["value"] == 1000 and ctx.get_metric("completeness.c_comment")["value"] == 0.9
    assert ctx.get_metric("approx_count_distinct.c_mktsegment")["value"] == 5
    assert ctx.get_metric("distinctness.c_nationkey")["value"] == 0.025 and ctx.get_metric("max.c_custkey")["value"] == 1000.0
    want_mean = sum(((i * 31) % 10000) / 100.0 for i in range(1, 1001)) / 1000
    assert abs(ctx.get_metric("mean.c_acctbal")["value"] - want_mean) < 1e-9


def test_performance_regression_thresholds():
    """term-guard/tests/performance_regression_test.rs: 10 k rows -- completeness < 300 ms (:146), statistics
    < 300 ms (:182), six constraints < 400 ms (:225); 50 k rows, ten constraints < 1000 ms (:317).  (Wall clock of
    ValidationSuite.run including planning, pattern compilation and the JSON bridge; first call warmed up.)"""
    import time

    def timed(suite, tbl):
        suite.run(tbl)
        t0 = time.perf_counter()
        r = suite.run(tbl)
        return (time.perf_counter() - t0) * 1e3, r

    t10 = customer(10_000)
    ms, _ = timed(ValidationSuite.builder("s").check(Check.builder("c").level(Level.ERROR)
                                                  .completeness(["c_custkey"]).build()).build(), t10)
    assert ms < 300
    ms, _ = timed(ValidationSuite.builder("s").check(
        Check.builder("c").level(Level.ERROR).has_min("c_acctbal", Assertion.GreaterThanOrEqual(0.0))
        .has_max("c_acctbal", Assertion.LessThan(1000.0)).has_mean("c_acctbal", Assertion.Between(0.0, 100.0)).build()).build(), t10)
    assert ms < 300
    six = (Check.builder("c").level(Level.ERROR).completeness(["c_custkey"]).validates_uniqueness(["c_custkey"], 1.0)
           .has_min("c_acctbal", Assertion.GreaterThanOrEqual(0.0)).has_max("c_acctbal", Assertion.LessThan(1000.0))
           .validates_regex("c_phone", r"^\d{2}-\d{3}-\d{3}-\d{4}$", 1.0).has_min_length("c_name", 5))
    ms, r = timed(ValidationSuite.builder("s").check(six.build()).build(), t10)
    assert ms < 400 and r.is_success()
    ten = (Check.builder("c").level(Level.ERROR).completeness(["c_custkey"]).completeness(["c_name"])
           .validates_uniqueness(["c_custkey"], 1.0).validates_distinctness(["c_mktsegment"], Assertion.LessThan(0.1))
           .has_min("c_acctbal", Assertion.GreaterThanOrEqual(0.0)).has_max("c_acctbal", Assertion.LessThan(1000.0))
           .has_mean("c_acctbal", Assertion.Between(0.0, 100.0)).validates_regex("c_phone", r"^\d{2}-", 1.0)
           .has_max_length("c_name", 25).is_contained_in("c_mktsegment", SEGMENTS))
    ms, r = timed(ValidationSuite.builder("s").check(ten.build()).build(), customer(50_000))
    assert ms < 1000 and r.is_success()
