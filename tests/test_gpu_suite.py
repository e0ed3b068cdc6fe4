"""-m gpu: ValidationSuite.run end to end (builder -> one fused tgx_plan -> HIP kernels -> verdicts), on the
reference's known-answer vectors and on configs[0] of BASELINE.json (10 k-row users.csv)."""
import csv
import io
import json

import numpy as np
import pyarrow as pa
import pytest

import oracle_binding as orc
import term_amd as T
from term_amd.suite import (Assertion, Check, CompletenessOptions, FormatOptions, Level, NullHandling,
                            ValidationSuite)

pytestmark = pytest.mark.gpu


def arrow_table(**cols):
    return pa.table({k: pa.array(v[1], type=v[0]) for k, v in cols.items()})


def test_reference_vectors_end_to_end(golden):
    # completeness
    for case in golden["completeness"]:
        from test_host_logic import parse_op

        tbl = arrow_table(**{c: (pa.int64(), v) for c, v in case["columns"].items()})
        suite = (ValidationSuite.builder("s").check(
            Check.builder("chk").level(Level.ERROR)
            .completeness(case["cols"], CompletenessOptions(case["threshold"], parse_op(case["operator"]))).build()).build())
        r = suite.run(tbl)
        m = r.report.metrics
        if case["status"] == "success":
            assert r.is_success() and m.passed_checks == 1, case["ref"]
            if "metric" in case:
                assert m.custom_metrics["chk.completeness"] == case["metric"]
        elif case["status"] == "failure":
            assert r.is_failure() and m.failed_checks == 1
            if "message_contains" in case:
                assert case["message_contains"] in r.report.issues[0].message
        else:
            assert m.skipped_checks == 1 and r.is_success()
    # statistics
    for case in golden["statistics"]:
        tbl = arrow_table(value=(pa.float64(), case["values"]))
        a = Assertion(case["assertion"][0], case["assertion"][1])
        r = (ValidationSuite.builder("s").check(Check.builder("chk").level(Level.ERROR)
                                                .statistic("value", case["stat"], a).build()).build()).run(tbl)
        if case["status"] == "success":
            assert r.is_success() and r.report.metrics.custom_metrics["chk." + case["stat"]] == case["metric"]
        else:
            assert r.is_failure() and case["message_contains"] in r.report.issues[0].message
    # uniqueness (Utf8 columns)
    for case in golden["uniqueness"]:
        tbl = arrow_table(test_col=(pa.string(), case["values"]))
        b = Check.builder("chk").level(Level.ERROR)
        kind = case["kind"]
        if kind == "full_uniqueness":
            b.validates_uniqueness(["test_col"], case["threshold"])
        elif kind == "distinctness":
            b.validates_distinctness(["test_col"], Assertion(*case["assertion"]))
        elif kind == "unique_value_ratio":
            b.validates_unique_value_ratio(["test_col"], Assertion(*case["assertion"]))
        elif kind == "primary_key":
            b.validates_primary_key(["test_col"])
        else:
            b.validates_uniqueness_with_nulls(["test_col"], case["threshold"], NullHandling.Include)
        r = ValidationSuite.builder("s").check(b.build()).build().run(tbl)
        m = r.report.metrics
        if case["status"] == "success":
            assert r.is_success() and m.passed_checks == 1, case["ref"]
            assert list(m.custom_metrics.values())[0] == case["metric"]
        elif case["status"] == "failure":
            assert r.is_failure() and case["message_contains"] in r.report.issues[0].message
        else:
            assert m.skipped_checks == 1
    # multi-column uniqueness: COUNT(DISTINCT (col1, col2)) as ONE tuple check (constraints/uniqueness.rs:1009-1041)
    for case in golden["uniqueness_multi"]:
        tbl = arrow_table(col1=(pa.string(), case["col1"]), col2=(pa.string(), case["col2"]))
        b = Check.builder("chk").level(Level.ERROR)
        if case["kind"] == "full_uniqueness":
            b.validates_uniqueness(["col1", "col2"], case["threshold"])
        else:
            b.validates_distinctness(["col1", "col2"], Assertion(*case["assertion"]))
        r = ValidationSuite.builder("s").check(b.build()).build().run(tbl)
        assert r.is_success(), case["ref"]
        assert abs(r.report.metrics.custom_metrics["chk." + case["kind"]] - case["metric"]) < 1e-12
    # length (constraints/length.rs:246-438)
    for case in golden["length"]:
        tbl = arrow_table(text=(pa.string(), case["values"]))
        b = Check.builder("chk").level(Level.ERROR).length("text", case["kind"], case.get("a", 0), case.get("b", 0))
        r = ValidationSuite.builder("s").check(b.build()).build().run(tbl)
        m = r.report.metrics
        if case["status"] == "success":
            assert r.is_success() and m.passed_checks == 1, (case["ref"], [i.message for i in r.report.issues])
            assert list(m.custom_metrics.values()) == [case["metric"]]
        elif case["status"] == "failure":
            assert r.is_failure() and case["message_contains"] in r.report.issues[0].message, case["ref"]
            assert r.report.issues[0].metric == case["metric"]
        else:
            assert m.skipped_checks == 1
        if "name" in case and case["status"] == "success":
            assert list(m.custom_metrics.keys()) == ["chk." + case["name"]]
    # approx_count_distinct (constraints/approx_count_distinct.rs:186-347): the metric is the HyperLogLog estimate of
    # the scan's lane -- byte-identical registers to the oracle's sketch, so the same estimate -- inside the bounds the
    # reference's tests put on DataFusion's estimate; string columns are answered by the exact key set
    for case in golden["approx_count_distinct"]:
        typ = pa.int64() if case["dtype"] == "int64" else pa.string()
        tbl = arrow_table(test_col=(typ, case["values"]))
        b = Check.builder("chk").level(Level.ERROR).has_approx_count_distinct("test_col", Assertion(*case["assertion"]))
        r = ValidationSuite.builder("s").check(b.build()).build().run(tbl)
        if case["status"] == "success":
            assert r.is_success(), (case["ref"], [i.message for i in r.report.issues])
            metric = r.report.metrics.custom_metrics["chk.approx_count_distinct"]
        else:
            assert r.is_failure() and r.report.issues[0].message == case["message"], case["ref"]
            metric = r.report.issues[0].metric
        if case["dtype"] == "int64":
            vals = np.array([0 if v is None else v for v in case["values"]], dtype=np.int64)
            mask = np.array([v is not None for v in case["values"]], dtype=bool)
            want = orc.hll_estimate(orc.hll_registers(vals, orc.pack_validity(mask) if len(vals) else None, n=len(vals)))
        else:
            want = case["exact"]
        assert metric == want and case["bounds"][0] <= metric <= case["bounds"][1], case["ref"]
    # containment (constraints/values.rs:520-601): the IN-list runs as an anchored alternation on the pattern kernel
    for case in golden["containment"]:
        tbl = arrow_table(text_col=(pa.string(), case["values"]))
        b = Check.builder("chk").level(Level.ERROR).is_contained_in("text_col", case["allowed"])
        r = ValidationSuite.builder("s").check(b.build()).build().run(tbl)
        if case["status"] == "success":
            assert r.is_success() and r.report.metrics.custom_metrics["chk.containment"] == case["metric"], case["ref"]
        else:
            assert r.is_failure() and case["message_contains"] in r.report.issues[0].message
            assert r.report.issues[0].metric == case["metric"]
    tbl = arrow_table(t=(pa.string(), ["a.b", "a|b", "axb", "it's", "", None]))
    r = ValidationSuite.builder("s").check(Check.builder("chk").level(Level.ERROR)
                                           .is_contained_in("t", ["a.b", "a|b", "it's", ""]).build()).build().run(tbl)
    assert r.is_failure() and r.report.issues[0].message == "1 values are not in the allowed set"  # only "axb"
    assert r.report.issues[0].metric == 0.8
    # formats
    for case in golden["format"]:
        fmt = case["format"]
        kw = {k: case[k] for k in ("pattern", "allow_localhost", "detect_only", "country") if k in case}
        opts = FormatOptions(case.get("case_sensitive", True), bool(case.get("trim")), case.get("null_is_valid", True))
        tbl = arrow_table(text_col=(pa.string(), case["values"]))
        r = (ValidationSuite.builder("s").check(Check.builder("chk").level(Level.ERROR)
                                                .has_format("text_col", fmt, case["threshold"], opts, **kw).build())
             .build()).run(tbl)
        m = r.report.metrics
        if case["status"] == "skipped":
            assert m.skipped_checks == 1
            continue
        assert (case["status"] == "success") == r.is_success(), case["ref"]
        got = list(m.custom_metrics.values())[0]
        assert got == case["metric"], case["ref"]


def make_users_csv(n):
    """docs/tutorials/02-validating-csv-files.md:22-29 schema"""
    rng = np.random.default_rng(42)
    buf = io.StringIO()
    w = csv.writer(buf)
    w.writerow(["user_id", "name", "email", "age", "signup_date"])
    for i in range(n):
        email = "user%d@example.com" % i if i % 20 else ("" if i % 40 else "not-an-email")
        age = int(rng.integers(18, 90)) if i % 50 else ""
        w.writerow([i + 1, "User %d" % i, email, age, "2024-%02d-%02d" % (1 + i % 12, 1 + i % 28)])
    return buf.getvalue()


def test_constraint_variant_vectors_end_to_end(golden):
    """MultiStatisticalConstraint, QuantileConstraint (Single / Multiple / Monotonic) and CorrelationConstraint
    (Pairwise / Range / Independence): the reference's unit tests (statistics.rs:642-682, quantile.rs:527-593,
    correlation.rs:586-631) through ValidationSuite.run on the device"""

    class Raw:  # CheckBuilder.constraint() takes any object with a `.spec`
        def __init__(self, spec): self.spec = spec

    for case in golden["constraint_variants"]:
        tbl = arrow_table(**{c: (pa.float64(), v) for c, v in case["table"].items()})
        r = (ValidationSuite.builder("s").check(Check.builder("chk").level(Level.ERROR)
                                                .constraint(Raw(case["constraint"])).build()).build()).run(tbl)
        m = r.report.metrics
        if case["status"] == "success":
            assert r.is_success() and m.passed_checks == 1, (case["ref"], r.to_json())
            if "metric_gt" in case:
                assert m.custom_metrics["chk." + case["name"]] > case["metric_gt"]
        else:
            assert r.is_failure() and m.failed_checks == 1, case["ref"]
            assert r.report.issues[0].constraint_name == case["name"]
            assert case["message_contains"] in r.report.issues[0].message
    # all variants in ONE suite over one table: the KLL sketch and the co-moments are planned once
    x = np.arange(1000, dtype=np.float64)
    tbl = arrow_table(x=(pa.float64(), x), y=(pa.float64(), 2 * x + (np.arange(1000) % 10) - 5.0))
    from term_amd.suite import (CorrelationConstraint, CorrelationType, MultiStatisticalConstraint, QuantileCheck,
                                QuantileConstraint, StatisticType)
    A = Assertion
    chk = (Check.builder("chk").level(Level.ERROR)
           .constraint(MultiStatisticalConstraint("x", [(StatisticType.Mean, A.Equals(499.5)), (StatisticType.Max, A.Equals(999)),
                                                        (StatisticType.Median, A.Between(480, 520)),
                                                        (StatisticType.StandardDeviation, A.Between(288, 289))]))
           .constraint(QuantileConstraint.multiple("x", [QuantileCheck(0.25, A.Between(240, 260)),
                                                         QuantileCheck(0.75, A.Between(740, 760))]))
           .constraint(QuantileConstraint.monotonic("x", [0.1, 0.5, 0.9], True))
           .constraint(CorrelationConstraint.covariance("x", "y", A.GreaterThan(0)))
           .constraint(CorrelationConstraint.range("x", "y", CorrelationType.Pearson, 0.99, 1.0))
           .constraint(CorrelationConstraint.independence("x", "y", 0.5))
           .constraint(CorrelationConstraint.spearman("x", "y", A.GreaterThan(0))).build())
    r = ValidationSuite.builder("s").check(chk).build().run(tbl)
    m = r.report.metrics
    assert (m.total_checks, m.passed_checks, m.failed_checks, m.skipped_checks) == (7, 5, 1, 1), r.to_json()
    assert m.custom_metrics["chk.multi_statistical"] == 499.5  # the first statistic
    st = orc.comoments(x, 2 * x + (np.arange(1000) % 10) - 5.0)
    assert abs(m.custom_metrics["chk.covariance"] - orc.covariance(st)) <= 1e-9 * abs(orc.covariance(st))
    (issue,) = r.report.issues
    assert issue.constraint_name == "independence" and "exceeding independence threshold 0.5" in issue.message


def test_config1_users_csv_plumbing():
    """BASELINE.json configs[0]: is_complete + has_min on a 10 k-row users.csv (host buffers in, verdict out)"""
    import pyarrow.csv as pcsv

    text = make_users_csv(10_000)
    tbl = pcsv.read_csv(io.BytesIO(text.encode()))
    assert tbl.schema.field("age").type == pa.int64() and tbl.column("age").null_count == 200
    suite = (ValidationSuite.builder("user_data_validation").table_name("users")
             .check(Check.builder("critical").level(Level.ERROR)
                    .completeness("user_id", CompletenessOptions.full())
                    .has_min("age", Assertion.GreaterThanOrEqual(0.0))
                    .validates_uniqueness(["user_id"], 1.0)
                    .has_size(Assertion.Equals(10_000)).build())
             .check(Check.builder("quality").level(Level.WARNING)
                    .completeness("age", CompletenessOptions.threshold(0.99))
                    .email("email", 0.96)
                    .has_mean("age", Assertion.Between(40.0, 65.0))
                    .has_approx_quantile("age", 0.5, Assertion.Between(45.0, 62.0)).build())
             .build())
    r = suite.run(tbl)
    assert r.is_success()  # only Warning-level failures
    m = r.report.metrics
    assert (m.total_checks, m.passed_checks, m.failed_checks) == (8, 6, 2)
    cm = m.custom_metrics
    assert cm["critical.completeness"] == 1.0 and cm["critical.full_uniqueness"] == 1.0 and cm["critical.size"] == 10_000
    ages = np.array([a for a in tbl.column("age").to_pylist() if a is not None], dtype=np.int64)
    assert cm["critical.min"] == float(ages.min()) and abs(cm["quality.mean"] - ages.mean()) < 1e-9
    assert cm["quality.completeness"] == 0.98
    failed = {i.constraint_name: i for i in r.report.issues}
    assert set(failed) == {"completeness", "email"}
    assert failed["completeness"].message == "Column 'age' completeness 98.00% is below threshold 99.00%"
    assert failed["email"].metric == 0.95 and failed["email"].level == "warning"
    assert abs(cm["quality.quantile"] - np.median(ages)) <= 2


def test_suite_tally_rules():
    """tests/integration_test_suite.rs:412-437, 1157-1207, 1245-1277"""
    tbl = arrow_table(id=(pa.int64(), [1, 2, 3, None]), v=(pa.float64(), [1.0, 2.0, 3.0, 4.0]))
    # missing column at Error level => failure, has_errors, the other constraints still evaluated
    r = (ValidationSuite.builder("s").check(Check.builder("c").level(Level.ERROR)
                                            .completeness("nope", CompletenessOptions.full())
                                            .has_max("v", Assertion.Equals(4.0)).build()).build()).run(tbl)
    assert r.is_failure() and r.report.has_errors() and r.report.metrics.passed_checks == 1
    assert r.report.issues[0].message.startswith("Error evaluating constraint: DataFusion error: Schema error: No field named nope")
    # a Warning-level failure next to Error / Info passes keeps the suite successful
    r = (ValidationSuite.builder("json_test")
         .check(Check.builder("e").level(Level.ERROR).has_size(Assertion.Equals(4)).build())
         .check(Check.builder("w").level(Level.WARNING).completeness("id", CompletenessOptions.full()).build())
         .check(Check.builder("i").level(Level.INFO).has_sum("v", Assertion.Equals(10.0)).build()).build()).run(tbl)
    assert r.is_success() and r.report.has_warnings() and not r.report.has_errors()
    assert (r.metrics().passed_checks, r.metrics().failed_checks) == (2, 1)
    text = r.to_json()
    assert '"suite_name": "json_test"' in text and '"status": "success"' in text
    assert json.loads(text)["metrics"]["custom_metrics"]["i.sum"] == 10.0
    # later duplicates of "{check}.{constraint}" overwrite (suite.rs:203-209)
    r = (ValidationSuite.builder("s").check(Check.builder("c").has_min("v", Assertion.Equals(1.0))
                                            .has_min("id", Assertion.Equals(1.0)).build()).build()).run(tbl)
    assert r.metrics().custom_metrics == {"c.min": 1.0}


def test_multi_batch_table_and_device_columns():
    import torch

    rng = np.random.default_rng(1)
    n = 300_000
    v = rng.standard_normal(n)
    k = rng.integers(0, 1000, size=n, dtype=np.int64)
    batches = []
    for lo in range(0, n, 8192):  # DataFusion's default batch size (core/context.rs:31)
        hi = min(n, lo + 8192)
        batches.append({"v": T.Column.float64(torch.from_numpy(v[lo:hi]).cuda()),
                        "k": T.Column.int64(torch.from_numpy(k[lo:hi]).cuda())})
    r = (ValidationSuite.builder("s").check(Check.builder("c").level(Level.ERROR)
                                            .has_mean("v", Assertion.Between(-0.01, 0.01))
                                            .has_standard_deviation("v", Assertion.Between(0.99, 1.01))
                                            .validates_distinctness(["k"], Assertion.LessThan(0.01))
                                            .has_correlation("v", "k", Assertion.Between(-0.01, 0.01))
                                            .has_approx_quantile("v", 0.95, Assertion.Between(1.6, 1.7)).build())
         .build()).run(batches)
    assert r.is_success(), [i.message for i in r.report.issues]
    cm = r.metrics().custom_metrics
    assert abs(cm["c.mean"] - v.mean()) < 1e-12 and abs(cm["c.standard_deviation"] - v.std(ddof=1)) < 1e-9
    assert cm["c.distinctness"] == len(np.unique(k)) / n
    assert abs(cm["c.correlation"] - np.corrcoef(v, k)[0, 1]) < 1e-9
