"""-m gpu: CorrelationConstraint (TG/constraints/correlation.rs:260-275, 355-362) emits `CORR(a, b)` /
`COVAR_SAMP(a, b)`: DataFusion's online co-moment accumulators.  The kernels sum about a pivot near the data
(kernels/device_types.h, ComomentAcc), so the Pearson / Covariance / Independence metrics must match the oracle's
restatement of those accumulators (oracle/tgx_oracle.c, orc_corr_online) to 1e-6 relative -- BASELINE.json's bound for
float aggregates -- on OFFSET data too, where the raw-sum form n Sxy - Sx Sy has lost its digits: epoch seconds,
Int64 epoch milliseconds, mean / sigma = 1e8; in one batch, in ragged batches, over threaded ranks, through the
stand-alone kernel and riding on the scan."""
import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
import term_amd.suite as S
from _lib_spec import spec
from gpu_util import numeric_column, rel_err
from term_amd.suite import Assertion, Check, CorrelationConstraint, Level, ValidationSuite

pytestmark = pytest.mark.gpu

TOL = 1e-6  # north_star: "within 1e-6 relative for float aggregates"


def offset_pair(kind, n, seed=3):
    rng = np.random.default_rng(seed)
    if kind == "epoch_seconds":
        x = 1.7e9 + rng.random(n) * 1000.0
        y = 3e8 + 0.5 * (x - 1.7e9) + rng.standard_normal(n) * 100.0
    elif kind == "epoch_millis_int64":
        x = 1_700_000_000_000 + rng.integers(0, 1000, size=n, dtype=np.int64)
        y = 300_000_000_000 + (x - 1_700_000_000_000) // 2 + rng.integers(-100, 100, size=n, dtype=np.int64)
    elif kind == "mean_over_sigma_1e8":
        x = 1e8 + rng.standard_normal(n)
        y = 5e8 + 0.3 * (x - 1e8) + rng.standard_normal(n)
    elif kind == "independent_offset":
        x = 1.7e9 + rng.random(n) * 1000.0
        y = -4e9 + rng.standard_normal(n) * 3.0
    else:
        raise ValueError(kind)
    xm, ym = rng.random(n) >= 0.05, rng.random(n) >= 0.03
    return np.ascontiguousarray(x), np.ascontiguousarray(y), orc.pack_validity(xm), orc.pack_validity(ym)


def metrics_of(result):
    """the three constraint metrics from one tgx_result, through the host's own verdict code"""
    d = {k: getattr(result, k) for k in ("total", "non_null", "sum_x", "sum_y", "sum_x2", "sum_y2", "sum_xy",
                                          "co_mean_x", "co_mean_y", "co_m2_x", "co_m2_y", "co_c_xy")}
    only = lambda b: b.build().spec["constraints"][0]  # noqa: E731
    wide = Assertion.Between(-1e300, 1e300)
    p = S.constraint_verdict(only(Check.builder("c").constraint(CorrelationConstraint.pearson("x", "y", wide))), [d])
    c = S.constraint_verdict(only(Check.builder("c").constraint(CorrelationConstraint.covariance("x", "y", wide))), [d])
    i = S.constraint_verdict(only(Check.builder("c").constraint(CorrelationConstraint.independence("x", "y", 1.0))), [d])
    assert p["status"] == c["status"] == i["status"] == "success", (p, c, i)
    return p["metric"], c["metric"], i["metric"]


def exact_corr(x, y, xv, yv):
    """two-pass CORR / COVAR_SAMP in extended precision: what the online accumulators approximate"""
    n = len(x)
    m = np.ones(n, bool)
    for v in (xv, yv):
        if v is not None:
            m &= np.unpackbits(v, bitorder="little")[:n].astype(bool)
    xl, yl = x[m].astype(np.longdouble), y[m].astype(np.longdouble)
    dx, dy = xl - xl.mean(), yl - yl.mean()
    cxy, m2x, m2y = (dx * dy).sum(), (dx * dx).sum(), (dy * dy).sum()
    return float(cxy / np.sqrt(m2x * m2y)), float(cxy / (len(xl) - 1))


def check_against_online(result, x, y, xv, yv):
    """Within 1e-6 relative of DataFusion's online accumulators (orc.corr_online) -- allowing for THEIR OWN rounding
    noise where the value is ill-conditioned: a correlation near 0 on offset columns comes out of the Welford
    recurrences with a relative error of 1e-3 (running means of 1.7e9-sized values carry 2e-7 of rounding each, the
    co-moment nearly cancels), measured here as |online - exact| against the two-pass value in extended precision.
    The kernels' own result must sit on the exact value (1e-9 of the metric's natural scale)."""
    want = orc.corr_online(x, y, xv, yv)
    ex_corr, ex_cov = exact_corr(x, y, xv, yv)
    assert result.non_null == want.n
    pearson, covar, indep = metrics_of(result)
    assert abs(pearson - want.corr) <= TOL * abs(want.corr) + 2 * abs(want.corr - ex_corr), (pearson, want.corr, ex_corr)
    assert abs(covar - want.covar_samp) <= TOL * abs(want.covar_samp) + 2 * abs(want.covar_samp - ex_cov), \
        (covar, want.covar_samp, ex_cov)
    assert indep == abs(pearson)
    assert abs(pearson - ex_corr) <= 1e-9 and abs(covar - ex_cov) <= 1e-9 * abs(ex_cov / ex_corr)
    # the analyzer's raw sums are still what TG/analyzers/advanced/correlation.rs:239-249 adds up
    raw = orc.comoments(x, y, xv, yv)
    for got, w in ((result.sum_x, raw.sum_x), (result.sum_y, raw.sum_y), (result.sum_x2, raw.sum_x2),
                   (result.sum_y2, raw.sum_y2), (result.sum_xy, raw.sum_xy)):
        assert rel_err(got, w) <= 1e-9


KINDS = ["epoch_seconds", "epoch_millis_int64", "mean_over_sigma_1e8", "independent_offset"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("n", [1000, 1_000_000, 3_000_017])  # small: stand-alone kernel; big: the pair rides on the scan
def test_offset_data_one_batch(kind, n):
    x, y, xv, yv = offset_pair(kind, n)
    T.init()
    plan = T.Plan([spec(T.COMOMENTS, 0, column2=1), spec(T.NUMERIC_STATS, 0), spec(T.NUMERIC_STATS, 1)])
    st = T.State(plan)
    st.update([numeric_column(x, xv, True), numeric_column(y, yv, True)])
    res = st.finalize()
    check_against_online(res[0], x, y, xv, yv)
    # a COMOMENTS spec alone takes the stand-alone kernel at every size
    alone = T.State(T.Plan([spec(T.COMOMENTS, 0, column2=1)]))
    alone.update([numeric_column(x, xv, True), numeric_column(y, yv, True)])
    check_against_online(alone.finalize()[0], x, y, xv, yv)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("device", [True, False])
def test_ragged_batches_and_merge(kind, device):
    """three ragged batches into one state; the same batches as three states merged; serialized and back"""
    n = 2_500_003
    x, y, xv, yv = offset_pair(kind, n, seed=11)
    cuts = [0, 1_000_001, 1_000_001 + 77, n]  # a big batch (rides on the scan), a tiny one, a ragged rest
    T.init()
    plan = T.Plan([spec(T.COMOMENTS, 0, column2=1), spec(T.NUMERIC_STATS, 0), spec(T.NUMERIC_STATS, 1)])
    batches = [[numeric_column(x, xv, device, offset=a, length=b - a), numeric_column(y, yv, device, offset=a, length=b - a)]
               for a, b in zip(cuts[:-1], cuts[1:])]
    st = T.State(plan)
    for b in batches:
        st.update(b)
    check_against_online(st.finalize()[0], x, y, xv, yv)
    parts = []
    for b in reversed(batches):  # every state picks its own pivots; the merge re-bases them
        s = T.State(plan)
        s.update(b)
        parts.append(s)
    merged = T.State(plan)
    merged.merge(parts)
    check_against_online(merged.finalize()[0], x, y, xv, yv)
    back = T.State.deserialize(plan, merged.serialize())
    r0, r1 = merged.finalize()[0], back.finalize()[0]
    assert (r0.co_c_xy, r0.co_m2_x, r0.co_m2_y, r0.sum_xy) == (r1.co_c_xy, r1.co_m2_x, r1.co_m2_y, r1.sum_xy)


@pytest.mark.parametrize("kind", ["epoch_seconds", "epoch_millis_int64"])
def test_eight_threaded_ranks(kind):
    """row shards over 8 ranks (threads on one GPU), tgx_allreduce: every rank ends with the table's metrics,
    bit-identical across ranks"""
    from test_gpu_distributed_sim import _run_ranks
    from term_amd.distributed import shard_rows

    n = 2_400_000 + 64 * 5
    x, y, xv, yv = offset_pair(kind, n, seed=23)
    T.init()
    plan = T.Plan([spec(T.COMOMENTS, 0, column2=1), spec(T.NUMERIC_STATS, 0), spec(T.NUMERIC_STATS, 1)])

    def shards_of(rank):
        lo, hi = shard_rows(n, 8, rank)
        return [numeric_column(x, xv, True, offset=lo, length=hi - lo), numeric_column(y, yv, True, offset=lo, length=hi - lo)]

    results = _run_ranks(8, plan, shards_of)
    first = results[0][0][0]
    for res, _st in results:
        r = res[0]
        assert (r.co_c_xy, r.co_m2_x, r.co_m2_y, r.co_mean_x, r.sum_xy) == \
            (first.co_c_xy, first.co_m2_x, first.co_m2_y, first.co_mean_x, first.sum_xy)
    check_against_online(first, x, y, xv, yv)


def test_suite_verdicts_on_offset_columns():
    """through ValidationSuite.run: Pairwise Pearson, Covariance, Range and Independence on epoch-like columns"""
    import pyarrow as pa

    n = 1_200_000
    x, y, _, _ = offset_pair("epoch_seconds", n, seed=5)
    want = orc.corr_online(x, y)
    tbl = pa.table({"x": pa.array(x), "y": pa.array(y)})
    lo, hi = want.corr * (1 - 1e-6), want.corr * (1 + 1e-6)
    chk = (Check.builder("chk").level(Level.ERROR)
           .constraint(CorrelationConstraint.pearson("x", "y", Assertion.Between(min(lo, hi), max(lo, hi))))
           .constraint(CorrelationConstraint.covariance("x", "y", Assertion.Between(want.covar_samp * (1 - 1e-6),
                                                                                    want.covar_samp * (1 + 1e-6))))
           .constraint(CorrelationConstraint.independence("x", "y", abs(want.corr) * (1 + 1e-6)))
           .build())
    r = ValidationSuite.builder("s").check(chk).build().run(tbl)
    assert r.is_success() and r.report.metrics.passed_checks == 3, r.to_json()
    m = r.report.metrics.custom_metrics
    assert rel_err(m["chk.correlation"], want.corr) <= TOL
    # the raw-moment product formula is three orders off on this column: the test would not pass with it
    raw = orc.comoments(x, y)
    nn = float(raw.n)
    product = (raw.sum_xy / nn - raw.sum_x / nn * raw.sum_y / nn) / np.sqrt(
        (raw.sum_x2 / nn - (raw.sum_x / nn) ** 2) * (raw.sum_y2 / nn - (raw.sum_y / nn) ** 2))
    assert rel_err(product, want.corr) > 10 * TOL


def test_pivot_edge_cases():
    """all-NULL leading rows (no pivot from the first look), a constant column (CORR = 0), one row, no rows"""
    T.init()
    plan = T.Plan([spec(T.COMOMENTS, 0, column2=1)])
    n = 300_000
    rng = np.random.default_rng(2)
    x = 1.7e9 + rng.random(n) * 10
    y = 2.0 * x + rng.standard_normal(n)
    m = np.ones(n, bool)
    m[: n // 2] = False  # the first batch is all NULL on x: pivots come from the second
    xv = orc.pack_validity(m)
    st = T.State(plan)
    half = n // 2 - (n // 2) % 64
    st.update([numeric_column(x, xv, True, length=half), numeric_column(y, None, True, length=half)])
    st.update([numeric_column(x, xv, True, offset=half, length=n - half), numeric_column(y, None, True, offset=half, length=n - half)])
    check_against_online(st.finalize()[0], x, y, xv, None)
    const = np.full(5000, 1.7e12)
    st = T.State(plan)
    st.update([numeric_column(const, None, True), numeric_column(x[:5000].copy(), None, True)])
    r = st.finalize()[0]
    assert r.co_m2_x == 0.0 and metrics_of(r)[0] == 0.0  # DataFusion: 0 when a deviation is 0
    st = T.State(plan)
    st.update([numeric_column(x[:1].copy(), None, True), numeric_column(y[:1].copy(), None, True)])
    r = st.finalize()[0]
    assert (r.non_null, r.co_m2_x, r.co_c_xy, r.sum_x) == (1, 0.0, 0.0, x[0])
    st = T.State(plan)
    st.update([numeric_column(x[:0].copy(), None, True), numeric_column(y[:0].copy(), None, True)])
    r = st.finalize()[0]
    assert (r.total, r.non_null, r.co_m2_x, r.sum_xy) == (0, 0, 0.0, 0.0)
