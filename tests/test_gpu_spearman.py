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


# ---- the ranking machinery itself (kernels/sortrank.hip): distributions x shapes of the partition -------------------
def _sort_data(kind, n, rng):
    if kind == "uniform":
        return rng.random(n) * 1000.0
    if kind == "normal":
        return rng.standard_normal(n)
    if kind == "sorted_ids":
        return np.arange(n, dtype=np.int64) * 3 - 17
    if kind == "reversed":
        return np.arange(n, 0, -1, dtype=np.int64)
    if kind == "two_values":
        return rng.integers(0, 2, size=n, dtype=np.int64) * 1000 - 500
    if kind == "five_values":
        return rng.integers(0, 5, size=n, dtype=np.int64).astype(np.float64) * 0.25
    if kind == "all_equal":
        return np.full(n, 42.5)
    if kind == "one_heavy":  # half the rows one value, the rest spread
        v = rng.random(n)
        v[rng.random(n) < 0.5] = 0.5
        return v
    if kind == "clusters":  # cluster centres with jitter far below the gaps: bins of the last pass collide
        return rng.integers(0, 50, size=n).astype(np.float64) + rng.random(n) * 1e-9
    if kind == "exponents":  # spread over the whole exponent range, both signs
        return np.sign(rng.standard_normal(n)) * 10.0 ** rng.uniform(-300, 300, size=n)
    if kind == "specials":
        v = rng.standard_normal(n)
        pick = rng.random(n)
        v[pick < 0.05] = np.nan
        v[(pick >= 0.05) & (pick < 0.10)] = np.inf
        v[(pick >= 0.10) & (pick < 0.15)] = -np.inf
        v[(pick >= 0.15) & (pick < 0.20)] = 0.0
        v[(pick >= 0.20) & (pick < 0.25)] = -0.0
        return v
    if kind == "big_ints":  # beyond 2^53: CAST(.. AS DOUBLE) makes neighbours equal
        return (2**62 + rng.integers(0, 4096, size=n, dtype=np.int64)).astype(np.int64)
    if kind == "outlier":  # one far key stretches the first bucket's span
        v = rng.random(n) * 1e-6
        v[n // 2] = 1e300
        return v
    raise ValueError(kind)


SORT_KINDS = ["uniform", "normal", "sorted_ids", "reversed", "two_values", "five_values", "all_equal", "one_heavy",
              "clusters", "exponents", "specials", "big_ints", "outlier"]
# knobs of kernels/sortrank.hip (sr_tuning): None = the shipped shape.  20 000 rows each.
SORT_SHAPES = {
    "shipped": None,
    "three_passes": dict(TGX_SORT_TARGET="16", TGX_SORT_CAP="64", TGX_SORT_SPLIT="15"),
    "two_passes": dict(TGX_SORT_TARGET="32", TGX_SORT_CAP="128", TGX_SORT_SPLIT="31", TGX_SORT_SAMPLE="8"),
    "one_pass": dict(TGX_SORT_TARGET="512", TGX_SORT_CAP="1024"),
    "chunked_last_pass": dict(TGX_SORT_TARGET="300", TGX_SORT_CAP="64", TGX_SORT_SLOWCAP="128", TGX_SORT_SPLIT="31",
                              TGX_SORT_SAMPLE="2"),
    "coarse": dict(TGX_SORT_TARGET="64", TGX_SORT_SPLIT="3", TGX_SORT_SAMPLE="1"),
}


def _check_spearman(x, y, xv=None, yv=None):
    res, _, st = run_plan([spec(T.SPEARMAN, 0, column2=1)], [[numeric_column(x, xv, True), numeric_column(y, yv, True)]])
    want = orc.spearman_state(x, y, xv, yv)
    got = res[0]
    assert got.non_null == want.n
    assert (got.sum_x, got.sum_y, got.sum_x2, got.sum_y2, got.sum_xy) == \
        (want.sum_x, want.sum_y, want.sum_x2, want.sum_y2, want.sum_xy)
    again = st.finalize()[0]  # the pairs came back permuted, each x beside its y: the same answer again
    assert (again.sum_x, again.sum_x2, again.sum_xy) == (want.sum_x, want.sum_x2, want.sum_xy)


@pytest.mark.parametrize("shape", list(SORT_SHAPES))
@pytest.mark.parametrize("kind", SORT_KINDS)
def test_ranking_distributions_and_partition_shapes(kind, shape, monkeypatch):
    env = SORT_SHAPES[shape]
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    n = 20_000 if env else 150_000
    import zlib
    rng = np.random.default_rng(zlib.crc32(("%s/%s" % (kind, shape)).encode()))
    x = _sort_data(kind, n, rng)
    other = SORT_KINDS[(SORT_KINDS.index(kind) + 5) % len(SORT_KINDS)]
    y = _sort_data(other, n, rng)
    xv = orc.pack_validity(rng.random(n) >= 0.07)
    _check_spearman(x, y, xv, None)


@pytest.mark.parametrize("n", [2047, 2048, 2049, 4097, 262_144, 262_145, 3_000_001])
def test_ranking_sizes_around_the_pass_boundaries(n):
    """2048 keys are ranked by one workgroup, up to 256 * 1024 take one partition pass, up to 65 536 * 1024 two"""
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n)
    y = np.round(x * 3 + rng.standard_normal(n), 2)  # ties
    _check_spearman(x, y)


def test_ranking_three_passes_at_size(monkeypatch):
    """64 keys a bucket: 5 M rows are 78 K buckets, i.e. three passes of 43 ways"""
    monkeypatch.setenv("TGX_SORT_TARGET", "64")
    rng = np.random.default_rng(5)
    n = 5_000_000
    x = rng.standard_normal(n)
    y = rng.integers(0, 1000, size=n).astype(np.float64)
    _check_spearman(x, y)


def test_ranking_oversized_bucket_default_shape(monkeypatch):
    """one sample key per bucket: bucket sizes vary like an exponential distribution, many exceed the 2048 keys the
    small kernel ranks and go to the large one, some exceed its 4096 and are ranked chunk against chunk"""
    monkeypatch.setenv("TGX_SORT_SAMPLE", "1")
    rng = np.random.default_rng(99)
    n = 1_500_000
    x = rng.standard_normal(n)
    y = rng.random(n)
    _check_spearman(x, y)


# ---- room from the sample instead of a counting read (all passes but the last; kernels/sortrank.h) --------------------
ROOMY_SHAPES = {
    "three_passes": dict(TGX_SORT_TARGET="16", TGX_SORT_CAP="64", TGX_SORT_SPLIT="15"),
    "two_passes": dict(TGX_SORT_TARGET="32", TGX_SORT_CAP="128", TGX_SORT_SPLIT="31", TGX_SORT_SAMPLE="8"),
    "coarse": dict(TGX_SORT_TARGET="64", TGX_SORT_SPLIT="3", TGX_SORT_SAMPLE="1"),
    "many_stretches": dict(TGX_SORT_TARGET="16", TGX_SORT_CAP="64", TGX_SORT_SPLIT="15", TGX_SORT_PARTS="64"),
}


@pytest.mark.parametrize("shape", list(ROOMY_SHAPES))
@pytest.mark.parametrize("kind", SORT_KINDS)
def test_ranking_with_room_from_the_sample(kind, shape, monkeypatch, capfd):
    for k, v in ROOMY_SHAPES[shape].items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("TGX_SORT_OPTIMISTIC_MIN", "1")
    monkeypatch.setenv("TGX_SORT_DEBUG", "1")
    n = 60_000
    import zlib
    rng = np.random.default_rng(zlib.crc32(("roomy/%s/%s" % (kind, shape)).encode()))
    x = _sort_data(kind, n, rng)
    other = SORT_KINDS[(SORT_KINDS.index(kind) + 3) % len(SORT_KINDS)]
    y = _sort_data(other, n, rng)
    yv = orc.pack_validity(rng.random(n) >= 0.05)
    _check_spearman(x, y, None, yv)
    assert "room from the sample" in capfd.readouterr().err  # (whether or not a bucket then was full)


@pytest.mark.parametrize("sigmas_x2", ["0", "2"])
def test_ranking_full_bucket_is_counted_again(sigmas_x2, monkeypatch, capfd):
    """no slack on the sample's estimate: some bucket is full, the job says so, the ranking runs again with counted
    buckets -- from the untouched pairs -- and the sums are the oracle's"""
    monkeypatch.setenv("TGX_SORT_OPTIMISTIC_MIN", "1")
    monkeypatch.setenv("TGX_SORT_SIGMAS_X2", sigmas_x2)
    monkeypatch.setenv("TGX_SORT_DEBUG", "1")
    rng = np.random.default_rng(int(sigmas_x2) + 11)
    n = 2_000_000
    x = rng.standard_normal(n)
    y = np.round(x + rng.standard_normal(n), 3)
    _check_spearman(x, y)
    err = capfd.readouterr().err
    if sigmas_x2 == "0":
        assert "again with counted buckets" in err


def test_ranking_room_from_the_sample_at_size():
    """the shipped shape from 1 Mi pairs on: keys in order on one side (every stretch of pass 0 sees other buckets),
    heavy ties on the other"""
    rng = np.random.default_rng(21)
    n = 6_000_000
    x = np.arange(n, dtype=np.float64) * 0.5
    y = rng.integers(0, 300, size=n).astype(np.float64) + (np.arange(n) % 7 == 0) * rng.random(n)
    _check_spearman(x, y)
    _check_spearman(y, x)


# ---- a first DEVICE batch without NULLs is lent, not copied (spearman_device.cpp, LentBatch) ------------------------
@pytest.mark.parametrize("kind", SORT_KINDS)
def test_ranking_straight_from_the_callers_columns(kind, monkeypatch, capfd):
    for k, v in ROOMY_SHAPES["three_passes"].items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("TGX_SORT_OPTIMISTIC_MIN", "1")
    n = 40_000
    import zlib
    rng = np.random.default_rng(zlib.crc32(("lent/%s" % kind).encode()))
    x = _sort_data(kind, n, rng)
    y = _sort_data(SORT_KINDS[(SORT_KINDS.index(kind) + 4) % len(SORT_KINDS)], n, rng)
    _check_spearman(x, y)  # (its second finalize ranks the pairs the state has adopted from the first ranking)
    # the caller's columns are only read
    cols = [numeric_column(x, None, True), numeric_column(y, None, True)]
    res, _, _ = run_plan([spec(T.SPEARMAN, 0, column2=1)], [cols])
    for col, host in zip(cols, (x, y)):
        assert (col._keep[0].cpu().numpy().view(np.uint8) == host.view(np.uint8)).all()


@pytest.mark.parametrize("between", ["nothing", "sync", "finalize"])
def test_lent_batch_then_more_batches(between, monkeypatch):
    """the view of the first batch becomes pairs of the state's own when another batch arrives, at a synchronisation
    (the caller may free the batch then), or with the first result"""
    monkeypatch.setenv("TGX_SORT_OPTIMISTIC_MIN", "1000")
    rng = np.random.default_rng({"nothing": 1, "sync": 2, "finalize": 3}[between])
    n, cut = 300_000, 200_000
    x = rng.standard_normal(n)
    y = np.round(rng.standard_normal(n) + x, 2)
    yv = orc.pack_validity(np.concatenate([np.ones(cut, bool), rng.random(n - cut) >= 0.1]))
    T.init()
    plan = T.Plan([spec(T.SPEARMAN, 0, column2=1)])
    st = T.State(plan)
    first = [numeric_column(x, None, True, offset=0, length=cut), numeric_column(y, None, True, offset=0, length=cut)]
    st.update(first)
    if between == "sync":
        st.sync()
        first = None  # (the device copies may go)
    elif between == "finalize":
        want = orc.spearman_state(x[:cut], y[:cut], None, None)
        got = st.finalize()[0]
        assert (got.non_null, got.sum_x, got.sum_y2, got.sum_xy) == (want.n, want.sum_x, want.sum_y2, want.sum_xy)
        first = None
    st.update([numeric_column(x, None, True, offset=cut, length=n - cut),
               numeric_column(y, yv, True, offset=cut, length=n - cut)])
    want = orc.spearman_state(x, y, None, yv)
    got = st.finalize()[0]
    assert (got.total, got.non_null) == (n, want.n)
    assert (got.sum_x, got.sum_y, got.sum_x2, got.sum_y2, got.sum_xy) == \
        (want.sum_x, want.sum_y, want.sum_x2, want.sum_y2, want.sum_xy)
    st.reset()
    st.update([numeric_column(x, None, True, offset=0, length=cut), numeric_column(y, None, True, offset=0, length=cut)])
    st.reset()  # (a view dropped unread)
    assert st.finalize()[0].non_null == 0


def test_two_lent_pairs_share_the_work_arrays(monkeypatch):
    """two pairs of one plan, both lent: the second ranking runs in the arrays the first one's pairs were left in, so
    the first pair's become the state's own before that; a third batch afterwards finds both sets of pairs intact"""
    monkeypatch.setenv("TGX_SORT_OPTIMISTIC_MIN", "1000")
    rng = np.random.default_rng(404)
    n, more = 400_000, 50_000
    a = rng.standard_normal(n + more)
    b = np.round(a * 2 + rng.standard_normal(n + more), 1)
    c = rng.integers(-1000, 1000, size=n + more).astype(np.int64)
    T.init()
    plan = T.Plan([spec(T.SPEARMAN, 0, column2=1), spec(T.SPEARMAN, 2, column2=0)])
    st = T.State(plan)

    def feed(lo, hi):
        cols = [numeric_column(a, None, True, offset=lo, length=hi - lo), numeric_column(b, None, True, offset=lo, length=hi - lo),
                numeric_column(c, None, True, offset=lo, length=hi - lo)]
        st.update(cols)
        return cols

    def check(hi):
        res = st.finalize()
        for r, (x, y) in zip(res, ((a, b), (c, a))):
            want = orc.spearman_state(x[:hi], y[:hi], None, None)
            assert (r.non_null, r.sum_x, r.sum_y, r.sum_x2, r.sum_y2, r.sum_xy) == \
                (want.n, want.sum_x, want.sum_y, want.sum_x2, want.sum_y2, want.sum_xy)

    keep = feed(0, n)
    check(n)
    check(n)  # (the results kept)
    keep = feed(n, n + more)
    check(n + more)
    del keep
