"""-m gpu: AnalysisRunner (TG/analyzers/runner.rs) end to end -- every analyzer of a run planned into ONE pass --
against the reference's analyzer tests (TG/analyzers/basic/tests.rs, runner.rs:204-396)."""
import math

import pytest

pa = pytest.importorskip("pyarrow")

import term_amd as T
from term_amd import suite as S

pytestmark = pytest.mark.gpu


def table_of(golden):
    t = golden["analyzers"]["table"]
    return pa.table({"id": pa.array(t["id"], pa.int64()), "value": pa.array(t["value"], pa.float64()),
                     "name": pa.array(t["name"], pa.string())})


def test_reference_analyzer_vectors(golden):
    tbl = table_of(golden)
    runner = S.AnalysisRunner()
    makers = {"completeness": S.CompletenessAnalyzer, "distinctness": S.DistinctnessAnalyzer, "mean": S.MeanAnalyzer,
              "min": S.MinAnalyzer, "max": S.MaxAnalyzer, "sum": S.SumAnalyzer}
    for e in golden["analyzers"]["expect"]:
        runner.add(S.SizeAnalyzer() if e["analyzer"] == "size" else makers[e["analyzer"]](e["column"]))
    ctx = runner.run(tbl)
    assert not ctx.has_errors(), ctx.errors()
    for e in golden["analyzers"]["expect"]:
        key = "size" if e["analyzer"] == "size" else "%s.%s" % (e["analyzer"], e["column"])
        m = ctx.get_metric(key)
        assert m["value"] == e["metric"], e["ref"]
        assert m["type"] == ("Long" if e["analyzer"] == "size" else "Double")
        for k, v in e.get("state", {}).items():
            assert ctx.states[key][k] == v, (e["ref"], k)


def test_runner_tests_of_the_reference():
    """runner.rs:213-236 table; :238-262 basic; :264-289 two analyzers; :316-340 errors; :342-380 twelve analyzers"""
    tbl = pa.table({"id": pa.array([1, 2, 3, 4, 5], pa.int64()),
                    "value": pa.array([10.0, None, 30.0, 40.0, 50.0], pa.float64())})
    ctx = S.AnalysisRunner().add(S.SizeAnalyzer()).add(S.CompletenessAnalyzer("value")).run(tbl)
    assert ctx.get_metric("size") == {"type": "Long", "value": 5}
    assert abs(ctx.get_metric("completeness.value")["value"] - 0.8) < 0.001
    # a failing analyzer is recorded and the others still run (continue_on_error defaults to true)
    ctx = (S.AnalysisRunner().add(S.CompletenessAnalyzer("non_existent_column")).add(S.SizeAnalyzer())
           .continue_on_error(True).run(tbl))
    assert ctx.has_errors() and len(ctx.errors()) == 1 and ctx.errors()[0]["analyzer_name"] == "completeness"
    assert "non_existent_column" in ctx.errors()[0]["error"]
    assert ctx.get_metric("size")["value"] == 5
    with pytest.raises(T.TgxError, match="Analyzer completeness failed"):
        (S.AnalysisRunner().add(S.CompletenessAnalyzer("non_existent_column")).add(S.SizeAnalyzer())
         .continue_on_error(False).run(tbl))
    r = S.AnalysisRunner()
    for a in (S.SizeAnalyzer(), S.CompletenessAnalyzer("id"), S.CompletenessAnalyzer("value"), S.DistinctnessAnalyzer("id"),
              S.DistinctnessAnalyzer("value"), S.MeanAnalyzer("value"), S.MinAnalyzer("value"), S.MaxAnalyzer("value"),
              S.SumAnalyzer("value"), S.MinAnalyzer("id"), S.MaxAnalyzer("id"), S.SumAnalyzer("id")):
        r.add(a)
    assert r.analyzer_count() == 12
    ctx = r.run(tbl)
    for key in ("size", "completeness.id", "completeness.value", "distinctness.id", "distinctness.value", "mean.value",
                "min.value", "max.value", "sum.value", "min.id", "max.id", "sum.id"):
        assert ctx.get_metric(key) is not None, key
    assert ctx.get_metric("sum.id")["value"] == 15.0 and ctx.get_metric("max.id")["value"] == 5.0
    assert ctx.get_metric("mean.value")["value"] == 32.5 and ctx.get_metric("distinctness.value")["value"] == 1.0
    ctx = S.AnalysisRunner().add(S.ApproxCountDistinctAnalyzer("value")).run(tbl)
    assert ctx.get_metric("approx_count_distinct.value") == {"type": "Long", "value": 4}
    assert ctx.states["approx_count_distinct.value"] == {"approx_distinct_count": 4, "total_count": 4}


def test_edge_tables_and_type_quirks(golden):
    for e in golden["analyzers"]["edge"]:
        vals = [None] * e.get("rows", 0)
        tbl = pa.table({"value": pa.array(vals, pa.float64())})
        ctx = (S.AnalysisRunner().add(S.SizeAnalyzer()).add(S.CompletenessAnalyzer("value")).add(S.MeanAnalyzer("value"))
               .add(S.SumAnalyzer("value")).run(tbl))
        assert ctx.get_metric("size")["value"] == e["size"], e["ref"]
        assert ctx.get_metric("completeness.value")["value"] == e["completeness"]
        errs = {x["analyzer_name"]: x["error"] for x in ctx.errors()}
        assert errs == {"mean": "No data available for analysis", "sum": "No data available for analysis"}
    # MeanAnalyzer reads SUM(col) as Float64 only (mean.rs:117-126): an Int64 column is an InvalidData error;
    # SumAnalyzer / MinAnalyzer accept Int64 (`as f64`)
    tbl = pa.table({"i": pa.array([1, 2, 3, None], pa.int64())})
    ctx = (S.AnalysisRunner().add(S.MeanAnalyzer("i")).add(S.SumAnalyzer("i")).add(S.MinAnalyzer("i"))
           .add(S.StandardDeviationAnalyzer("i")).run(tbl))
    errs = {x["analyzer_name"]: x["error"] for x in ctx.errors()}
    assert errs == {"mean": "Invalid data: Expected Float64 array for sum",
                    "standard_deviation": "Invalid data: Expected Float64 for sum"}
    assert ctx.get_metric("sum.i")["value"] == 6.0 and ctx.get_metric("min.i")["value"] == 1.0


def test_standard_deviation_and_correlation_analyzers():
    vals = [2.0, 4.0, 4.0, 4.0, 5.0, 5.0, 7.0, 9.0, None]
    xs = [float(i) for i in range(100)]
    tbl = pa.table({"v": pa.array(vals + [None] * 91, pa.float64()), "x": pa.array(xs, pa.float64()),
                    "y": pa.array([2 * x + 1 for x in xs], pa.float64())})
    ctx = (S.AnalysisRunner().add(S.StandardDeviationAnalyzer("v")).add(S.CorrelationAnalyzer("x", "y", "pearson"))
           .add(S.CorrelationAnalyzer("x", "y", "covariance")).add(S.CorrelationAnalyzer("x", "y", "spearman")).run(tbl))
    assert not ctx.has_errors(), ctx.errors()
    sd = {k: v["value"] for k, v in ctx.get_metric("standard_deviation")["value"].items()}
    assert sd["count"] == 8 and sd["mean"] == 5.0 and abs(sd["std_dev"] - 2.0) < 1e-9
    assert abs(sd["sample_variance"] - 32.0 / 7.0) < 1e-9
    st = ctx.states["standard_deviation"]
    assert st["count"] == 8 and st["sum"] == 40.0 and abs(st["sum_squared"] - 232.0) < 1e-9
    assert abs(ctx.get_metric("correlation_pearson_x_y")["value"] - 1.0) < 1e-12    # correlation.rs:505-525
    assert abs(ctx.get_metric("correlation_spearman_x_y")["value"] - 1.0) < 1e-12   # :527-548
    cov = sum((x - 49.5) * (2 * x + 1 - 100.0) for x in xs) / 99
    assert abs(ctx.get_metric("correlation_covariance_x_y")["value"] - cov) < 1e-9 * cov
    # the states of two shards merge into the state of the whole table (Pearson / covariance)
    half = tbl.slice(0, 50), tbl.slice(50)
    an = S.CorrelationAnalyzer("x", "y", "pearson")
    parts = [S.AnalysisRunner().add(an).run(h).states[an.metric_key()] for h in half]
    merged = an.merge_states(parts)
    whole = ctx.states[an.metric_key()]
    assert merged["n"] == whole["n"] and abs(merged["sum_xy"] - whole["sum_xy"]) <= 1e-9 * whole["sum_xy"]
    assert abs(an.compute_metric_from_state(merged)["value"] - 1.0) < 1e-12


def test_incremental_runner_partitions_equal_the_whole_table(tmp_path):
    """TG/analyzers/incremental/runner.rs: analyze_partition per day + analyze_partitions over the stored states ==
    one run over all rows; analyze_incremental grows a partition in place"""
    import numpy as np

    rng = np.random.default_rng(3)
    n = 30_000
    v = rng.standard_normal(n) * 50
    mask = rng.random(n) >= 0.1
    x = rng.random(n)
    tbl = pa.table({"v": pa.array(np.where(mask, v, np.nan), pa.float64(), mask=~mask), "x": pa.array(x, pa.float64()),
                    "y": pa.array(3 * x + rng.standard_normal(n) * 0.1, pa.float64())})
    analyzers = [S.SizeAnalyzer(), S.CompletenessAnalyzer("v"), S.MeanAnalyzer("v"), S.MinAnalyzer("v"), S.MaxAnalyzer("v"),
                 S.SumAnalyzer("v"), S.StandardDeviationAnalyzer("v"), S.CorrelationAnalyzer("x", "y", "pearson")]
    whole = S.AnalysisRunner()
    for a in analyzers:
        whole.add(a)
    want = whole.run(tbl)
    for store in (S.InMemoryStateStore(), S.FileSystemStateStore(tmp_path / "s")):
        r = S.IncrementalAnalysisRunner(store)
        for a in analyzers:
            r.add_analyzer(a)
        r.analyze_partition(tbl.slice(0, 10_000), "p0")
        r.analyze_partition(tbl.slice(10_000, 5_000), "p1")
        r.analyze_incremental(tbl.slice(15_000, 7_000), "p1")      # p1 now covers rows 10000..22000
        ctx_last = r.analyze_incremental(tbl.slice(22_000), "p2")  # a partition that did not exist yet
        assert ctx_last.get_metric("size")["value"] == 8000
        got = r.analyze_partitions(r.list_partitions())
        assert got.get_metric("size") == want.get_metric("size")
        assert got.get_metric("completeness.v") == want.get_metric("completeness.v")
        assert got.get_metric("min.v") == want.get_metric("min.v") and got.get_metric("max.v") == want.get_metric("max.v")
        for key in ("mean.v", "sum.v", "correlation_pearson_x_y"):
            a, b = got.get_metric(key)["value"], want.get_metric(key)["value"]
            assert abs(a - b) <= 1e-9 * abs(b), key
        sa = {k: x_["value"] for k, x_ in got.get_metric("standard_deviation")["value"].items()}
        sb = {k: x_["value"] for k, x_ in want.get_metric("standard_deviation")["value"].items()}
        assert sa["count"] == sb["count"]
        for k in ("mean", "std_dev", "sample_variance"):
            assert abs(sa[k] - sb[k]) <= 1e-9 * abs(sb[k]), k


def test_an_analyzer_the_library_refuses_does_not_cost_the_others_their_metrics():
    """In the reference every analyzer is a query of its own (runner.rs:141-201): one that fails is recorded, the others
    run.  Here all analyzers share ONE pass, and a request the library refuses (MIN of a Boolean column: COUNT and
    DISTINCT checks only) used to fail that pass for everybody.  Now the refused analyzer keeps the refusal as its error
    and the rest run again as one pass."""
    n = 20_000
    tbl = pa.table({"id": pa.array(range(n), pa.int64()),
                    "flag": pa.array([i % 3 == 0 for i in range(n)], pa.bool_()),
                    "value": pa.array([float(i % 100) for i in range(n)], pa.float64())})
    ctx = (S.AnalysisRunner().add(S.SizeAnalyzer()).add(S.MinAnalyzer("flag")).add(S.MeanAnalyzer("value"))
           .add(S.CompletenessAnalyzer("flag")).add(S.DistinctnessAnalyzer("id")).run(tbl))
    assert [e["analyzer_name"] for e in ctx.errors()] == ["min"] and "TGX_UNSUPPORTED" in ctx.errors()[0]["error"]
    assert ctx.get_metric("size")["value"] == n and ctx.get_metric("mean.value")["value"] == 49.5
    assert ctx.get_metric("completeness.flag")["value"] == 1.0 and ctx.get_metric("distinctness.id")["value"] == 1.0
    assert ctx.get_metric("min.flag") is None
