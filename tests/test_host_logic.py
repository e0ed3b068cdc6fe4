"""CPU tests of the host-side mirror of the reference's builder / verdict layer (term_amd/csrc/host).

Aggregates are produced by the oracle and handed to the product's verdict functions through
tgx_host_constraint_verdict_json, so statuses, metrics and message texts of the reference's own unit tests are
checked without a GPU (the GPU suites in test_gpu_suite.py run the same vectors end to end)."""
import json

import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from term_amd import suite as S
from term_amd.suite import (Assertion, Check, CompletenessOptions, FormatOptions, Level, LogicalOperator, NullHandling,
                            ValidationSuite)


def only(builder):
    return builder.build().spec["constraints"][0]


def parse_op(op):
    if op in ("all", "any"):
        return op
    k, n = op.split(":")
    return {k: int(n)}


def test_assertion_vectors(golden):
    """constraints/assertion.rs:48-76"""
    for a in golden["assertion"]:
        A = Assertion(a["kind"], *a["args"])
        assert A.evaluate(a["value"]) == a["ok"], a
        assert A.description() == a["text"]
    assert Assertion.Equals(0.1).description() == "equals 0.1"
    assert Assertion.GreaterThan(-2.5e-5).description() == "greater than -0.000025"
    assert Assertion.Between(1e15, 1.5e300).description().startswith("between 1000000000000000 and 15000000")


def test_completeness_vectors(golden):
    for case in golden["completeness"]:
        c = only(Check.builder("c").completeness(case["cols"], CompletenessOptions(case["threshold"], parse_op(case["operator"]))))
        results = []
        for name in case["cols"]:
            vals, validity = orc.column_from_list(case["columns"][name], np.int64)
            cnt = orc.count(validity, len(vals))
            results.append({"total": cnt.total, "non_null": cnt.non_null})
        v = S.constraint_verdict(c, results)
        assert v["status"] == case["status"], case["ref"]
        assert v["name"] == "completeness"
        if "metric" in case:
            assert v["metric"] == case["metric"]
        if "message_contains" in case:
            assert case["message_contains"] in v["message"]
    # message formats: completeness.rs:239-243, unified.rs:77-107
    c = only(Check.builder("c").completeness(["a", "b"], CompletenessOptions.full()))
    v = S.constraint_verdict(c, [{"total": 3, "non_null": 3}, {"total": 3, "non_null": 2}])
    assert v["message"] == "Constraint failed for columns: b. Required: all"
    v = S.constraint_verdict(c, [{"total": 3, "non_null": 3}, {"total": 3, "non_null": 3}])
    assert v["message"] == "All 2 columns satisfy the constraint" and v["metric"] == 1.0
    c = only(Check.builder("c").any_complete(["a", "b"]))
    v = S.constraint_verdict(c, [{"total": 3, "non_null": 3}, {"total": 3, "non_null": 0}])
    assert v["message"] == "Columns a satisfy the constraint" and v["metric"] == 0.5
    with pytest.raises(T.TgxError) as e:
        only(Check.builder("c").completeness("col", CompletenessOptions.threshold(1.5)))
        S.constraint_plan(only(Check.builder("c").completeness("col", CompletenessOptions.threshold(1.5))))
    assert "Threshold must be between 0.0 and 1.0" in str(e.value)  # completeness.rs:542-546


def stats_result(vals, validity):
    st = orc.stats(vals, validity)
    return {"total": st.total, "non_null": st.non_null, "has_value": st.has_value, "is_float": st.is_float,
            "min_f": st.min_f if st.has_value else 0.0, "max_f": st.max_f if st.has_value else 0.0,
            "sum_f": st.sum_f, "sum_i": st.sum_i_wrapping, "mean": st.mean if st.has_value else 0.0,
            "has_variance": st.has_variance, "var_samp": st.var_samp if st.has_variance else 0.0,
            "stddev_samp": st.stddev_samp if st.has_variance else 0.0}


def test_statistics_vectors(golden):
    for case in golden["statistics"]:
        vals, validity = orc.column_from_list(case["values"], np.float64)
        a = Assertion(case["assertion"][0], case["assertion"][1])
        c = only(Check.builder("c").statistic("value", case["stat"], a))
        v = S.constraint_verdict(c, [stats_result(vals, validity)])
        assert v["status"] == case["status"], case["ref"]
        assert v["name"] == case["stat"]
        if "metric" in case:
            assert v["metric"] == case["metric"]
        if "message_contains" in case:
            assert case["message_contains"] in v["message"]
    vals, validity = orc.column_from_list([None, None], np.float64)
    v = S.constraint_verdict(only(Check.builder("c").has_min("v", Assertion.Equals(0))), [stats_result(vals, validity)])
    assert v["message"] == "minimum is null (no non-null values)"  # statistics.rs:284-301
    vals = np.array([1.0, 2.0, 4.5])
    v = S.constraint_verdict(only(Check.builder("c").has_max("v", Assertion.LessThan(4))), [stats_result(vals, None)])
    assert v == {"status": "failure", "metric": 4.5, "message": "maximum 4.5 does not less than 4", "name": "max"}
    v = S.constraint_verdict(only(Check.builder("c").has_standard_deviation("v", Assertion.Between(1, 2))),
                             [stats_result(vals, None)])
    assert v["status"] == "success" and abs(v["metric"] - np.std(vals, ddof=1)) < 1e-12


def distinct_result(values):
    offs, data, validity = orc.utf8_from_list(values)
    d = orc.distinct_utf8(offs, data, validity)
    return {"total": d.total, "non_null": d.non_null, "distinct": d.distinct, "groups_once": d.groups_once}


def test_uniqueness_vectors(golden):
    for case in golden["uniqueness"]:
        kind = case["kind"]
        b = Check.builder("c")
        if kind == "full_uniqueness":
            b.validates_uniqueness(["test_col"], case["threshold"])
        elif kind == "distinctness":
            b.validates_distinctness(["test_col"], Assertion(*case["assertion"]))
        elif kind == "unique_value_ratio":
            b.validates_unique_value_ratio(["test_col"], Assertion(*case["assertion"]))
        elif kind == "primary_key":
            b.validates_primary_key(["test_col"])
        elif kind == "unique_with_nulls_include":
            b.validates_uniqueness_with_nulls(["test_col"], case["threshold"], NullHandling.Include)
        v = S.constraint_verdict(only(b), [distinct_result(case["values"])])
        assert v["status"] == case["status"], case["ref"]
        if "metric" in case:
            assert v["metric"] == case["metric"], case["ref"]
        if "message_contains" in case:
            assert case["message_contains"] in v["message"]
    v = S.constraint_verdict(only(Check.builder("c").validates_uniqueness(["k"], 0.9)), [distinct_result(["A", "B", "A", "A"])])
    assert v["message"] == "Uniqueness ratio 0.500 is below threshold 0.900 for columns: k"  # uniqueness.rs:751-754
    v = S.constraint_verdict(only(Check.builder("c").validates_primary_key(["k"])), [distinct_result(["A", "B", "A"])])
    assert v["message"] == "Primary key columns contain 1 duplicate values: k" and abs(v["metric"] - 1 / 3) < 1e-15
    v = S.constraint_verdict(only(Check.builder("c").validates_primary_key(["k"])), [distinct_result(["A", None, None])])
    assert v["message"] == "Primary key columns contain 2 NULL values: k"
    v = S.constraint_verdict(only(Check.builder("c").validates_distinctness(["k"], Assertion.GreaterThan(0.9))),
                             [distinct_result(["A", "B", "A", "A"])])
    assert v["message"] == "distinctness ratio 0.500 does not satisfy greater than 0.9 for columns: k"
    with pytest.raises(T.TgxError) as e:
        S.constraint_plan(only(Check.builder("c").validates_uniqueness(["col"], 1.5)))
    assert "Threshold must be between 0.0 and 1.0" in str(e.value)
    with pytest.raises(T.TgxError) as e:
        S.constraint_plan(only(Check.builder("c").validates_uniqueness([], 1.0)))
    assert "At least one column must be specified" in str(e.value)


def test_format_vectors(golden):
    from test_regex_host import pattern_of

    for case in golden["format"]:
        fmt = case["format"]
        opts = FormatOptions(case.get("case_sensitive", True), bool(case.get("trim")), case.get("null_is_valid", True))
        kw = {}
        if fmt == "regex":
            kw["pattern"] = case["pattern"]
        if "allow_localhost" in case:
            kw["allow_localhost"] = case["allow_localhost"]
        if "detect_only" in case:
            kw["detect_only"] = case["detect_only"]
        if "country" in case:
            kw["country"] = case["country"]
        c = only(Check.builder("c").has_format("text_col", fmt, case["threshold"], opts, **kw))
        plan = S.constraint_plan(c)
        assert plan["requests"][0]["pattern"] == pattern_of(case, golden["patterns"])
        if "name" in case:
            assert plan["name"] == case["name"]
        vals = case["values"]
        offs, data, validity = orc.utf8_from_list(vals)
        m = orc.Regex(plan["requests"][0]["pattern"], not opts.case_sensitive_).count_utf8(
            offs, data, validity, trim=opts.trim_, null_is_valid=opts.null_is_valid_)
        v = S.constraint_verdict(c, [{"total": m.total, "matches": m.matches}])
        assert v["status"] == case["status"], case["ref"]
        if "metric" in case:
            assert v["metric"] == case["metric"]
    c = only(Check.builder("c").validates_email("e", 0.9))
    v = S.constraint_verdict(c, [{"total": 4, "matches": 1}])
    assert v["message"] == ("Format validation ratio 0.250 is below threshold 0.900 - values that are valid email addresses")
    c = only(Check.builder("c").validates_credit_card("cc", 0.1, detect_only=True))
    v = S.constraint_verdict(c, [{"total": 4, "matches": 3}])
    assert v["status"] == "failure" and v["message"] == "Credit card detection ratio 0.750 exceeds threshold 0.100"
    with pytest.raises(T.TgxError) as e:
        S.constraint_plan(only(Check.builder("c").validates_email("col", 1.5)))
    assert "Threshold must be between 0.0 and 1.0" in str(e.value)  # format.rs:1273-1281
    # builder defaults: phone / postal code / ssn trim, email() trims and rejects NULLs
    assert S.constraint_plan(only(Check.builder("c").validates_phone("p", 0.5, "US")))["requests"][0]["flags"] == \
        T.FLAG_TRIM | T.FLAG_NULL_IS_VALID
    assert S.constraint_plan(only(Check.builder("c").email("p", 0.5)))["requests"][0]["flags"] == T.FLAG_TRIM


def test_size_quantile_correlation_verdicts():
    v = S.constraint_verdict(only(Check.builder("c").has_size(Assertion.Equals(0))), [{"total": 0}])
    assert v["status"] == "success" and v["metric"] == 0  # size evaluates on an empty table (size.rs:66-116)
    c = only(Check.builder("c").has_approx_quantile("x", 0.9, Assertion.LessThan(5)))
    v = S.constraint_verdict(c, [{"kll_n": 10, "quantiles": {"0.9": 7.5}}])
    assert v["message"] == "Quantile 0.9 is 7.5 which does not less than 5" and v["name"] == "quantile"
    x = np.arange(100, dtype=np.float64)
    st = orc.comoments(x, 2 * x + 1)
    c = only(Check.builder("c").has_correlation("x", "y", Assertion.GreaterThan(0.99)))
    v = S.constraint_verdict(c, [{"total": 100, "non_null": st.n, "sum_x": st.sum_x, "sum_y": st.sum_y,
                                  "sum_x2": st.sum_x2, "sum_y2": st.sum_y2, "sum_xy": st.sum_xy}])
    assert v["status"] == "success" and abs(v["metric"] - 1.0) < 1e-9 and v["name"] == "correlation"


def _variant_results(constraint, table):
    """oracle aggregates for every request the constraint plans (SpecRequest kinds of include/tgx.h)"""
    phis = {0.5, constraint.get("quantile", 0.5), *constraint.get("quantiles", [])}
    phis |= {c["quantile"] for c in constraint.get("checks", [])}
    phis |= {st.get("p", 0.5) for st in constraint.get("statistics", [])}
    results = []
    for req in S.constraint_plan(constraint)["requests"]:
        col = np.array(table[req["column"]], dtype=np.float64)
        if req["kind"] == T.NUMERIC_STATS:
            results.append(stats_result(col, None))
        elif req["kind"] == T.KLL:
            sk = orc.Kll(req["kll_k"])
            sk.update_many(col)
            results.append({"kll_n": len(col), "quantiles": {repr(float(p)): sk.quantile(p) for p in phis}})
        elif req["kind"] == T.COMOMENTS:
            st = orc.comoments(col, np.array(table[req["column2"]], dtype=np.float64))
            results.append({"total": len(col), "non_null": st.n, "sum_x": st.sum_x, "sum_y": st.sum_y,
                            "sum_x2": st.sum_x2, "sum_y2": st.sum_y2, "sum_xy": st.sum_xy})
        else:
            raise AssertionError(req)
    return results


def test_constraint_variant_vectors(golden):
    """MultiStatisticalConstraint (statistics.rs:642-682), QuantileConstraint Single / Multiple / Monotonic
    (quantile.rs:527-593), CorrelationConstraint Pairwise / Independence / Range (correlation.rs:586-631): the
    reference's own unit tests, aggregates from the oracle, verdicts from the product's host layer"""
    for case in golden["constraint_variants"]:
        v = S.constraint_verdict(case["constraint"], _variant_results(case["constraint"], case["table"]))
        assert v["status"] == case["status"], (case["ref"], v)
        assert v["name"] == case["name"]
        if "message_contains" in case:
            assert case["message_contains"] in v["message"], v
        if "metric_gt" in case:
            assert v["metric"] > case["metric_gt"]
    for case in golden["constraint_variant_errors"]:
        with pytest.raises(T.TgxError) as e:
            S.constraint_plan(case["constraint"])
        assert case["error_contains"] in str(e.value), case["ref"]


def test_constraint_variant_rules():
    """verdict rules of the variants beyond what the reference's unit tests pin (cited lines)"""
    A = Assertion
    tbl = {"value": [10.0, 20.0, 30.0, 40.0]}
    # statistics.rs:498-503: the metric of a success is the FIRST statistic; failures are joined by "; ", no metric
    c = S.MultiStatisticalConstraint("value", [(S.StatisticType.Max, A.LessThan(50)), (S.StatisticType.Min, A.Equals(10))]).spec
    v = S.constraint_verdict(c, _variant_results(c, tbl))
    assert v == {"status": "success", "metric": 40.0, "message": None, "name": "multi_statistical"}
    c = S.MultiStatisticalConstraint("value", [(S.StatisticType.Min, A.Equals(5)), (S.StatisticType.Sum, A.LessThan(50)),
                                               (S.StatisticType.Median, A.Between(20, 30))]).spec
    v = S.constraint_verdict(c, _variant_results(c, tbl))
    assert v["status"] == "failure" and v["metric"] is None
    assert v["message"] == "minimum is 10 which does not equals 5; sum is 100 which does not less than 50"
    # an all-NULL column: every statistic "is null" (statistics.rs:466-470)
    c = S.MultiStatisticalConstraint("value", [(S.StatisticType.Min, A.Equals(5)), (S.StatisticType.Mean, A.Equals(1))]).spec
    v = S.constraint_verdict(c, [{"total": 3, "non_null": 0, "has_value": 0}])
    assert v["message"] == "minimum is null; mean is null"
    # quantile.rs:403-417: "Q{pct} is {value} which does not {assertion}", joined by "; "
    c = S.QuantileConstraint.multiple("value", [S.QuantileCheck(0.25, A.LessThan(1)), S.QuantileCheck(0.999, A.GreaterThan(99))]).spec
    v = S.constraint_verdict(c, [{"kll_n": 4, "quantiles": {"0.25": 10.0, "0.999": 40.5}}])
    assert v["message"] == "Q25 is 10 which does not less than 1; Q99 is 40.5 which does not greater than 99"
    assert v["metric"] is None
    # quantile.rs:455-478: strict needs values[i] > values[i-1]; the message prints the Vec<f64> with {:?}
    c = S.QuantileConstraint.monotonic("value", [0.1, 0.5, 0.9], True).spec
    q = {"kll_n": 4, "quantiles": {"0.1": 10.0, "0.5": 10.0, "0.9": 2.5e-5}}
    v = S.constraint_verdict(c, [q])
    assert v["status"] == "failure" and v["message"] == "Quantiles are not strictly monotonic: [10.0, 10.0, 2.5e-5]"
    c = S.QuantileConstraint.monotonic("value", [0.1, 0.5], False).spec
    assert S.constraint_verdict(c, [q])["status"] == "success"
    c = S.QuantileConstraint.monotonic("value", [0.5, 0.9], False).spec
    assert S.constraint_verdict(c, [q])["message"] == "Quantiles are not  monotonic: [10.0, 2.5e-5]"
    v = S.constraint_verdict(S.QuantileConstraint.distribution("value").spec, [{"kll_n": 4}])
    assert v == {"status": "skipped", "metric": None, "message": "Validation type not yet implemented", "name": "quantile"}
    with pytest.raises(T.TgxError) as e:
        S.constraint_plan(S.QuantileConstraint.multiple("value", []).spec)
    assert "At least one quantile check is required" in str(e.value)  # quantile.rs:196-200
    # correlation.rs: covariance = COVAR_SAMP, names, messages
    x = np.arange(10, dtype=np.float64)
    tbl = {"x": x, "y": 3 * x + 1}
    c = S.CorrelationConstraint.covariance("x", "y", A.Between(27.0, 28.0)).spec
    v = S.constraint_verdict(c, _variant_results(c, tbl))
    assert v["status"] == "success" and v["name"] == "covariance" and abs(v["metric"] - 3 * np.var(x, ddof=1)) < 1e-12
    c = S.CorrelationConstraint.covariance("x", "y", A.LessThan(1.0)).spec
    v = S.constraint_verdict(c, _variant_results(c, tbl))
    assert v["message"] == "covariance between x and y is 27.5 which does not less than 1"
    c = S.CorrelationConstraint.range("x", "y", S.CorrelationType.Pearson, -0.5, 0.5).spec
    v = S.constraint_verdict(c, _variant_results(c, tbl))
    assert v["name"] == "correlation_range" and v["status"] == "failure"
    assert v["message"].startswith("Pearson correlation between x and y is ") and v["message"].endswith(
        " which does not between -0.5 and 0.5")
    c = S.CorrelationConstraint.independence("x", "y", 0.25).spec
    tbl_neg = {"x": x, "y": -3 * x}
    v = S.constraint_verdict(c, _variant_results(c, tbl_neg))
    assert v["status"] == "failure" and abs(v["metric"] - 1.0) < 1e-12 and v["name"] == "independence"  # ABS(CORR)
    assert v["message"].startswith("Columns x and y have correlation ") and v["message"].endswith(
        " exceeding independence threshold 0.25")
    # types / validations the reference leaves unimplemented are Skipped with its texts (:336-341, :440-442)
    v = S.constraint_verdict(S.CorrelationConstraint.spearman("x", "y", A.GreaterThan(0)).spec, [])
    assert v == {"status": "skipped", "metric": None, "message": "Correlation type not yet implemented",
                 "name": "spearman_correlation"}
    v = S.constraint_verdict(S.CorrelationConstraint.multi_column(["a", "b"]).spec, [])
    assert v["message"] == "Validation type not yet implemented" and v["name"] == "multi_correlation"
    # Custom SQL: the unsafe-content screen is the reference's (:324-330); anything else cannot run on this path
    bad = S.CorrelationConstraint.pairwise("x", "y", S.CorrelationType.Custom, A.GreaterThan(0), "CORR(x, y); DROP TABLE t").spec
    v = S.constraint_verdict(bad, [])
    assert v["status"] == "failure" and v["message"] == "Custom SQL expression contains potentially unsafe content"
    with pytest.raises(T.TgxError) as e:
        S.constraint_plan(S.CorrelationConstraint.pairwise("x", "y", S.CorrelationType.Custom, A.GreaterThan(0),
                                                           "CORR({column1}, {column2})").spec)
    assert "Operation not supported" in str(e.value)


def test_identifier_rules():
    """security.rs:103-146, 212-255"""
    for ok in ["id", "user_id", "_private", "schema.table", '"quoted"', "created_at", "updated_by", "a1", "t.c1"]:
        S.validate_identifier(ok)
    bad = {"": "cannot be empty", "a" * 129: "too long", "1abc": "Invalid SQL identifier format",
           "col name": "Invalid SQL identifier format", "a;b": "Invalid SQL identifier format", "a.": "Invalid",
           "xp_cmdshell": "system stored procedure", "sp_help": "system stored procedure",
           "union_all": "suspicious SQL keyword pattern: 'union'", "drop_table": "suspicious SQL keyword pattern: 'drop'",
           "a--b": "Invalid"}
    for ident, needle in bad.items():
        with pytest.raises(T.TgxError) as e:
            S.validate_identifier(ident)
        assert needle in str(e.value), ident
    with pytest.raises(T.TgxError):
        S.constraint_plan(only(Check.builder("c").has_min("drop table x", Assertion.Equals(0))))


def test_suite_without_table_reports_errors_and_json_shape():
    """tests/integration_test_suite.rs:391-408: missing table at Error level => failure with an error issue"""
    suite = (ValidationSuite.builder("json_test").table_name("missing")
             .check(Check.builder("c").level(Level.ERROR).has_size(Assertion.GreaterThan(0)).build()).build())
    r = suite.run(None)
    assert r.is_failure() and r.report.has_errors()
    assert r.report.issues[0].message.startswith("Error evaluating constraint: ")
    d = json.loads(r.to_json())
    assert d["status"] == "failure" and d["report"]["suite_name"] == "json_test" and "metrics" not in d
    assert '"suite_name": "json_test"' in r.to_json()
    # Warning-level check: same error, but the suite still succeeds (integration_test_suite.rs:441-465)
    suite = (ValidationSuite.builder("w").check(Check.builder("c").has_size(Assertion.GreaterThan(0)).build()).build())
    r = suite.run(None)
    assert r.is_success() and r.report.has_warnings() and r.metrics().failed_checks == 1
    assert '"status": "success"' in r.to_json()


def test_statistics_follow_the_reference_result_type_rule():
    """constraints/statistics.rs:277-308: the aggregate's column is read as Int64Array, then Float64Array, else
    Err("Failed to extract statistic value") -- DataFusion's MIN / MAX / APPROX_PERCENTILE_CONT keep the input type, SUM
    widens signed integers to Int64 and floats to Float64 (unsigned: UInt64), AVG / STDDEV / VARIANCE are Float64.
    `strict_reference_types` (default) mirrors that; off, the widened column's value answers.  MultiStatistical pushes
    "Failed to compute {name}" (:478-482); QuantileConstraint also reads Int32 and fails with TypeMismatch (quantile.rs:
    308-324)."""
    vals, validity = orc.column_from_list([3, 1, 2, None], np.int64)
    res = stats_result(vals, validity)
    gt0 = Assertion.GreaterThan(0.0)
    ok = {  # statistic -> the Arrow types the reference reads a value for
        "min": ["Int64", "Float64"], "max": ["Int64", "Float64"],
        "sum": ["Int64", "Float64", "Int8", "Int16", "Int32", "Float32"],
        "mean": ["Int64", "Float64", "Int8", "Int16", "Int32", "Float32", "UInt8", "UInt16", "UInt32", "UInt64"],
        "standard_deviation": ["Int64", "Float64", "Int8", "Int16", "Int32", "Float32", "UInt8", "UInt16", "UInt32", "UInt64"],
    }
    every = ["Int64", "Float64", "Int8", "Int16", "Int32", "Float32", "UInt8", "UInt16", "UInt32", "UInt64", "Date32",
             "Timestamp(Nanosecond, None)"]
    for stat, good in ok.items():
        for t in every:
            if stat in ("sum", "mean", "standard_deviation") and (t.startswith("Date") or t.startswith("Timestamp")):
                continue  # (DataFusion refuses to plan these: the shim falls back to the stock constraint)
            c = dict(only(Check.builder("c").statistic("value", stat, gt0)), column_type=t)
            if t in good:
                assert S.constraint_verdict(c, [res])["status"] == "success", (stat, t)
            else:
                with pytest.raises(T.TgxError) as e:
                    S.constraint_verdict(c, [res])
                assert "Internal error: Failed to extract statistic value" in str(e.value), (stat, t)
                loose = dict(c, strict_reference_types=False)
                assert S.constraint_verdict(loose, [res])["status"] == "success", (stat, t)
    # no declared type: the column is what its tgx_type says (Int64 / Float64 here)
    assert S.constraint_verdict(only(Check.builder("c").statistic("value", "min", gt0)), [res])["status"] == "success"
    # MultiStatisticalConstraint: a failure entry per unreadable statistic, no error
    multi = dict(S.MultiStatisticalConstraint("value", [("min", gt0), ("mean", gt0), ("max", gt0)]).spec,
                 column_type="Int32")
    v = S.constraint_verdict(multi, [res])
    assert v["status"] == "failure" and v["message"] == "Failed to compute minimum; Failed to compute maximum"
    v = S.constraint_verdict(dict(multi, column_type="Int64"), [res])
    assert v["status"] == "success" and v["metric"] == 1.0
    # QuantileConstraint: Float64 / Int64 / Int32 are read, the rest is a TypeMismatch
    q = only(Check.builder("c").has_approx_quantile("value", 0.5, gt0))
    kll = [{"kll_n": 3, "quantiles": {"0.5": 2.0}}]
    for t in ("Int32", "Int64", "Float64"):
        assert S.constraint_verdict(dict(q, column_type=t), kll)["status"] == "success"
    with pytest.raises(T.TgxError) as e:
        S.constraint_verdict(dict(q, column_type="Float32"), kll)
    assert "Type mismatch: expected Float64, Int64, or Int32, found Float32" in str(e.value)
    assert S.constraint_verdict(dict(q, column_type="Float32", strict_reference_types=False), kll)["status"] == "success"
