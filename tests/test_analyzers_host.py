"""CPU: the device-free halves of the Analyzer mirror -- AnalyzerState::merge and compute_metric_from_state -- against
the reference's own analyzer tests (TG/analyzers/basic/tests.rs, advanced/standard_deviation.rs,
advanced/correlation.rs; vectors in tests/golden/reference_vectors.json)."""
import math

import pytest

import term_amd as T
from term_amd import suite as S


def test_state_merges_match_reference_tests(golden):
    for case in golden["analyzers"]["merges"]:
        a, st, want = case["analyzer"], case["states"], case["merged"]
        if a == "size":
            m = S.SizeAnalyzer().merge_states([{"count": c} for c in st])
            assert m == {"count": want}, case["ref"]
        elif a == "completeness":
            m = S.CompletenessAnalyzer("c").merge_states([{"total_count": t, "non_null_count": n} for t, n in st])
            assert (m["total_count"], m["non_null_count"]) == tuple(want)
            assert S.CompletenessAnalyzer("c").compute_metric_from_state(m)["value"] == want[1] / want[0]
        elif a == "mean":
            an = S.MeanAnalyzer("c")
            m = an.merge_states([{"sum": s_, "count": c} for s_, c in st])
            assert (m["sum"], m["count"]) == tuple(want)
            assert an.compute_metric_from_state(m) == {"type": "Double", "value": case["metric"]}
        elif a == "minmax":
            m = S.MinAnalyzer("c").merge_states([{"min": lo, "max": hi} for lo, hi in st])
            assert (m["min"], m["max"]) == tuple(want)
            assert S.MaxAnalyzer("c").compute_metric_from_state(m)["value"] == want[1]
        elif a == "sum":
            m = S.SumAnalyzer("c").merge_states([{"sum": s_, "has_values": True} for s_ in st])
            assert m == {"sum": want, "has_values": True}


def test_metric_rules_and_errors():
    # empty dataset is complete / fully distinct (completeness.rs:62-68, distinctness.rs:35-41)
    assert S.CompletenessAnalyzer("c").compute_metric_from_state({"total_count": 0, "non_null_count": 0})["value"] == 1.0
    assert S.DistinctnessAnalyzer("c").compute_metric_from_state({"total_count": 0, "distinct_count": 0})["value"] == 1.0
    # distinct counts merge as a clamped sum: an upper bound (distinctness.rs:77-92)
    m = S.DistinctnessAnalyzer("c").merge_states([{"total_count": 4, "distinct_count": 3}, {"total_count": 2, "distinct_count": 2}])
    assert m == {"total_count": 6, "distinct_count": 5}
    m = S.DistinctnessAnalyzer("c").merge_states([{"total_count": 2, "distinct_count": 9}])
    assert m["distinct_count"] == 2
    # approx_count_distinct.rs:45-61, 126-128: counts merge by max, the metric is a Long
    a = S.ApproxCountDistinctAnalyzer("c")
    m = a.merge_states([{"approx_distinct_count": 7, "total_count": 10}, {"approx_distinct_count": 9, "total_count": 5}])
    assert m == {"approx_distinct_count": 9, "total_count": 15}
    assert a.compute_metric_from_state(m) == {"type": "Long", "value": 9} and a.metric_key() == "approx_count_distinct.c"
    # NoData (mean.rs:147-152, sum.rs:145-151, min_max.rs:167-172)
    for an, st in ((S.MeanAnalyzer("c"), {"sum": 0.0, "count": 0}), (S.SumAnalyzer("c"), {"sum": 0.0, "has_values": False}),
                   (S.MinAnalyzer("c"), {"min": None, "max": None}), (S.MaxAnalyzer("c"), {"min": None, "max": None})):
        with pytest.raises(T.TgxError, match="No data available for analysis"):
            an.compute_metric_from_state(st)
    # min/max merge ignores empty states
    assert S.MinAnalyzer("c").merge_states([{"min": None, "max": None}, {"min": 3.0, "max": 4.0}]) == {"min": 3.0, "max": 4.0}
    # metric keys (traits.rs:133-135 and the per-analyzer overrides)
    assert S.SizeAnalyzer().metric_key() == "size"
    assert S.CompletenessAnalyzer("value").metric_key() == "completeness.value"
    assert S.StandardDeviationAnalyzer("value").metric_key() == "standard_deviation"
    assert S.CorrelationAnalyzer("x", "y", "spearman").metric_key() == "correlation_spearman_x_y"
    with pytest.raises(T.TgxError, match="unknown analyzer type"):
        S._Analyzer({"type": "entropy", "column": "c"}).merge_states([])


def test_standard_deviation_state_algebra():
    """advanced/standard_deviation.rs:37-130, 239-279 on [2, 4, 4, 4, 5, 5, 7, 9] (mean 5, population std 2)"""
    vals = [2.0, 4.0, 4.0, 4.0, 5.0, 5.0, 7.0, 9.0]
    an = S.StandardDeviationAnalyzer("v")

    def state(xs):
        return {"count": len(xs), "sum": sum(xs), "sum_squared": sum(x * x for x in xs), "mean": sum(xs) / len(xs)}

    m = an.merge_states([state(vals[:3]), state(vals[3:])])
    assert m == state(vals)
    met = an.compute_metric_from_state(m)
    assert met["type"] == "Map"
    v = {k: x["value"] for k, x in met["value"].items()}
    assert v["count"] == 8 and v["mean"] == 5.0
    assert abs(v["std_dev"] - 2.0) < 1e-12 and abs(v["variance"] - 4.0) < 1e-12
    assert abs(v["sample_variance"] - 32.0 / 7.0) < 1e-12 and abs(v["sample_std_dev"] - math.sqrt(32.0 / 7.0)) < 1e-12
    assert abs(v["coefficient_of_variation"] - 0.4) < 1e-12
    # one value: no sample statistics; zero mean: no coefficient of variation
    v1 = an.compute_metric_from_state(state([3.0]))["value"]
    assert "sample_std_dev" not in v1 and v1["std_dev"]["value"] == 0.0
    v0 = an.compute_metric_from_state(state([-1.0, 1.0]))["value"]
    assert "coefficient_of_variation" not in v0
    with pytest.raises(T.TgxError, match="No states to merge"):
        an.merge_states([])


def test_correlation_state_algebra(golden):
    """advanced/correlation.rs:64-110 (merge), 407-435 (metric) on y = 2x + 1"""
    xs = list(range(100))
    ys = [2 * x + 1 for x in xs]

    def state(x, y, t):
        return {"n": len(x), "sum_x": float(sum(x)), "sum_y": float(sum(y)), "sum_x2": float(sum(a * a for a in x)),
                "sum_y2": float(sum(b * b for b in y)), "sum_xy": float(sum(a * b for a, b in zip(x, y))),
                "x_ranks": None, "y_ranks": None, "correlation_type": t}

    p = S.CorrelationAnalyzer("x", "y", "pearson")
    m = p.merge_states([state(xs[:40], ys[:40], "Pearson"), state(xs[40:], ys[40:], "Pearson")])
    assert m["n"] == 100 and m["sum_xy"] == state(xs, ys, "Pearson")["sum_xy"]
    assert abs(p.compute_metric_from_state(m)["value"] - 1.0) < 1e-12
    c = S.CorrelationAnalyzer("x", "y", "covariance")
    want_cov = (m["sum_xy"] - m["sum_x"] * m["sum_y"] / 100) / 99
    assert abs(c.compute_metric_from_state(dict(m, correlation_type="Covariance"))["value"] - want_cov) < 1e-9
    # n < 2 -> NaN (serialised as null); zero variance -> 0.0; rank-based states do not merge
    assert p.compute_metric_from_state(state([1], [2], "Pearson"))["value"] is None
    assert p.compute_metric_from_state(state([1, 1, 1], [2, 3, 4], "Pearson"))["value"] == 0.0
    with pytest.raises(T.TgxError, match="Cannot merge rank-based correlation states"):
        S.CorrelationAnalyzer("x", "y", "spearman").merge_states([state(xs, ys, "Spearman")])


def test_incremental_runner_merges_stored_states_without_data(tmp_path):
    """TG/analyzers/incremental/runner.rs:320-414 over a FileSystemStateStore laid out like the reference's
    (state_store.rs:37-127: <base>/<partition>/<metric_key>.json, serde field names) -- no device involved"""
    store = S.FileSystemStateStore(tmp_path / "states")
    store.save_state("2024-01-01", {"size": {"count": 10}, "completeness.v": {"total_count": 10, "non_null_count": 8},
                                    "mean.v": {"sum": 100.0, "count": 4}, "min.v": {"min": 10.0, "max": 30.0}})
    store.save_state("2024-01-02", {"size": {"count": 20}, "completeness.v": {"total_count": 20, "non_null_count": 18},
                                    "mean.v": {"sum": 50.0, "count": 2}, "min.v": {"min": 5.0, "max": 40.0}})
    assert store.list_partitions() == ["2024-01-01", "2024-01-02"]
    import json, os
    assert json.load(open(os.path.join(tmp_path, "states", "2024-01-01", "mean.v.json"))) == {"sum": 100.0, "count": 4}
    r = (S.IncrementalAnalysisRunner(store).add_analyzer(S.SizeAnalyzer()).add_analyzer(S.CompletenessAnalyzer("v"))
         .add_analyzer(S.MeanAnalyzer("v")).add_analyzer(S.MinAnalyzer("v")).add_analyzer(S.SumAnalyzer("v")))
    ctx = r.analyze_partitions(["2024-01-01", "2024-01-02"])
    assert ctx.get_metric("size") == {"type": "Long", "value": 30}
    assert ctx.get_metric("completeness.v")["value"] == 26 / 30
    assert ctx.get_metric("mean.v")["value"] == 25.0 and ctx.get_metric("min.v")["value"] == 5.0
    assert ctx.get_metric("sum.v") is None  # no stored state for that analyzer: skipped (runner.rs:357-362)
    assert S.IncrementalAnalysisRunner(store).analyze_partitions([]).metrics == {}
    r.delete_partition("2024-01-01")
    assert r.list_partitions() == ["2024-01-02"]
    assert r.analyze_partitions(["2024-01-01", "2024-01-02"]).get_metric("size")["value"] == 20
