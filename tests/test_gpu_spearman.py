"""-m gpu: Spearman rank-sum state (SQL RANK(), min-rank ties, UInt64 wrapping) vs the oracle, bit-exact."""
import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import make_f64, make_i64, numeric_column, run_plan

pytestmark = pytest.mark.gpu


def state_of(r):
    return orc.Comoments(int(r.non_null), r.sum_x, r.sum_y, r.sum_x2, r.sum_y2, r.sum_xy)


def test_reference_vector(golden):
    """analyzers/advanced/correlation.rs:527-548: x = 0..99, y = 2x + 1 => Spearman 1.0"""
    c = golden["correlation"]
    x = np.arange(c["n"], dtype=np.float64)
    y = 2.0 * x + 1.0
    res, _, _ = run_plan([spec(T.SPEARMAN, 0, column2=1)], [[numeric_column(x, None, True), numeric_column(y, None, True)]])
    assert abs(orc.pearson(state_of(res[0])) - c["spearman"]["value"]) < c["spearman"]["tol"]
    want = orc.spearman_state(x, y)
    assert (res[0].non_null, res[0].sum_x, res[0].sum_x2, res[0].sum_xy) == (want.n, want.sum_x, want.sum_x2, want.sum_xy)


@pytest.mark.parametrize("n", [1, 2, 1000, 250_000])
def test_ties_nulls_mixed_types_bit_exact(n):
    rng = np.random.default_rng(n)
    xi, xv = make_i64(rng, n, -50, 50, null_frac=0.1)  # heavy ties
    yf, yv = make_f64(rng, n, "normal", null_frac=0.1)
    yf = np.round(yf, 1)
    yf[rng.random(n) < 0.05] = 0.0
    yf[rng.random(n) < 0.05] = -0.0
    res, _, _ = run_plan([spec(T.SPEARMAN, 0, column2=1), spec(T.COMOMENTS, 0, column2=1)],
                         [[numeric_column(xi, xv, True), numeric_column(yf, yv, True)]])
    want = orc.spearman_state(xi, yf, xv, yv)
    got = res[0]
    assert (got.total, got.non_null) == (n, want.n)
    assert (got.sum_x, got.sum_y, got.sum_x2, got.sum_y2, got.sum_xy) == \
        (want.sum_x, want.sum_y, want.sum_x2, want.sum_y2, want.sum_xy)
    assert res[1].non_null == want.n


def test_wrapping_past_the_u64_limit_and_exact_flag():
    """5 M rows: the reference's UInt64 sums of squares wrap; the default reproduces that bit for bit, the
    TGX_FLAG_EXACT_RANK_SUMS variant gives the true sums (deviation stated in DESIGN.md)"""
    n = 5_000_000
    rng = np.random.default_rng(4)
    x = rng.permutation(n).astype(np.float64)
    y = x + rng.standard_normal(n) * 1000
    cols = [numeric_column(x, None, True), numeric_column(y, None, True)]
    res, _, _ = run_plan([spec(T.SPEARMAN, 0, column2=1), spec(T.SPEARMAN, 0, column2=1, flags=T.FLAG_EXACT_RANK_SUMS)],
                         [cols])
    want = orc.spearman_state(x, y)
    assert (res[0].sum_x2, res[0].sum_xy) == (want.sum_x2, want.sum_xy)  # wrapped like the reference
    true_sum_sq = n * (n + 1) * (2 * n + 1) // 6
    assert true_sum_sq > 2**64 and res[1].sum_x2 == float(true_sum_sq) and res[1].sum_y2 == float(true_sum_sq)
    rho = orc.pearson(state_of(res[1]))
    import scipy.stats

    assert abs(rho - scipy.stats.spearmanr(x, y)[0]) < 1e-9  # no ties here, so min-rank == average-rank


def test_multi_batch_and_not_mergeable():
    rng = np.random.default_rng(8)
    n = 60_000
    x, _ = make_f64(rng, n, "uniform")
    y, yv = make_f64(rng, n, "normal", null_frac=0.2)
    T.init()
    plan = T.Plan([spec(T.SPEARMAN, 0, column2=1)])
    st = T.State(plan)
    for lo, hi in [(0, 10_000), (10_000, 10_001), (10_001, n)]:
        st.update([numeric_column(x, None, True, offset=lo, length=hi - lo),
                   numeric_column(y, yv, True, offset=lo, length=hi - lo)])
    res = st.finalize()
    want = orc.spearman_state(x, y, None, yv)
    assert (res[0].non_null, res[0].sum_x, res[0].sum_xy) == (want.n, want.sum_x, want.sum_xy)
    assert st.finalize()[0].sum_xy == want.sum_xy  # finalize does not consume the state
    other = T.State(plan)
    with pytest.raises(T.TgxError) as e:
        other.merge([st])
    assert e.value.status == "TGX_UNSUPPORTED"  # correlation.rs:103-109: rank-based states do not merge
    with pytest.raises(T.TgxError):
        st.serialize()


def test_state_reuse_across_sizes_and_pairs():
    """The ranking's work buffers stay with the state and are shared by its pairs (spearman_device.cpp): a small
    table, a larger one, a small one again through the SAME state (reset in between), two pairs per plan, and batch
    lengths that are no multiple of the compaction trip (2048 rows)."""
    T.init()
    plan = T.Plan([spec(T.SPEARMAN, 0, column2=1), spec(T.SPEARMAN, 1, column2=2)])
    st = T.State(plan)
    for n in (1500, 300_001, 777, 4097):
        rng = np.random.default_rng(n)
        a, av = make_f64(rng, n, "normal", null_frac=0.15)
        b, bv = make_i64(rng, n, -1000, 1000, null_frac=0.05)
        c, _ = make_f64(rng, n, "uniform")
        st.reset()
        cut = n // 3
        for lo, hi in ((0, cut), (cut, n)):
            st.update([numeric_column(a, av, True, offset=lo, length=hi - lo),
                       numeric_column(b, bv, True, offset=lo, length=hi - lo),
                       numeric_column(c, None, True, offset=lo, length=hi - lo)])
        res = st.finalize()
        for r, want in ((res[0], orc.spearman_state(a, b, av, bv)), (res[1], orc.spearman_state(b, c, bv, None))):
            assert (r.total, r.non_null) == (n, want.n)
            assert (r.sum_x, r.sum_y, r.sum_x2, r.sum_y2, r.sum_xy) == \
                (want.sum_x, want.sum_y, want.sum_x2, want.sum_y2, want.sum_xy), n
