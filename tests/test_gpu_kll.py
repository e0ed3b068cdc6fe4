"""-m gpu: KLL sketches built by kernels/kll.hip, queried through the C ABI.

Parity rule (DESIGN.md "KLL"): the reference's own sketch keeps every item (Compactor::compact hands one
half up and keeps the other), so it cannot be run at bench sizes and its answers drift far from the true
quantiles (its tests allow 60-85 % relative error).  The GPU sketch is a weight-preserving KLL; it must
(a) satisfy every assertion the reference's tests make, (b) stay within the stated rank error
eps = 1.65/sqrt(k) of the exact quantile, and (c) agree with the oracle within the reference's tolerance."""
import math

import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import numeric_column, run_plan

pytestmark = pytest.mark.gpu


def rank_error(sorted_vals, q, phi):
    """|rank(q) - phi*n| / n using the closest rank of q in the exact data"""
    n = len(sorted_vals)
    lo = np.searchsorted(sorted_vals, q, side="left")
    hi = np.searchsorted(sorted_vals, q, side="right")
    target = phi * n
    if lo <= target <= hi:
        return 0.0
    return min(abs(lo - target), abs(hi - target)) / n


def total_weight(st, idx):
    s = st.kll_summary(idx)
    return sum(len(st.kll_level_items(idx, l)) << l for l in range(s["num_levels"]))


def test_reference_known_answers(golden):
    for case in golden["kll"]:
        if "error_bound" in case:
            assert abs(T.lib().tgx_kll_relative_error_bound(case["k"]) - case["error_bound"]) < 0.001
            continue

        def series(s):
            if isinstance(s, str):
                lo, hi = s.split("..")
                return np.arange(int(lo), int(hi) + 1, dtype=np.float64)
            return np.array([float("nan") if v == "nan" else float(v) for v in s], dtype=np.float64)

        T.init()
        plan = T.Plan([spec(T.KLL, 0, kll_k=case["k"])])
        if "merge" in case:
            parts = []
            for sp in case["merge"]:
                st = T.State(plan)
                st.update([numeric_column(series(sp), None, True)])
                parts.append(st)
            st = parts[0]
            st.merge(parts[1:])
        else:
            st = T.State(plan)
            st.update([numeric_column(series(case["input"]), None, True)])
        res = st.finalize()
        assert res[0].kll_n == case["count"], case["ref"]
        for chk in case["checks"]:
            q = st.kll_quantile(0, chk["phi"])
            if "equals" in chk:
                assert q == chk["equals"]
            else:
                assert abs(q - chk["expected"]) / chk["expected"] < chk["rel_err_lt"]
                # and far tighter than the reference's own tolerance: these inputs fit level 0 unsampled
                assert abs(q - chk["expected"]) <= 2.0


def test_empty_and_bad_phi():
    T.init()
    plan = T.Plan([spec(T.KLL, 0, kll_k=100)])
    st = T.State(plan)
    with pytest.raises(T.TgxError) as e:
        st.kll_quantile(0, 0.5)
    assert "empty sketch" in str(e.value)  # kll_sketch.rs:247-251
    st.update([numeric_column(np.array([1.0, 2.0]), None, True)])
    with pytest.raises(T.TgxError) as e:
        st.kll_quantile(0, 1.5)
    assert "phi must be in [0, 1]" in str(e.value)
    with pytest.raises(T.TgxError):
        T.Plan([spec(T.KLL, 0, kll_k=1)])  # k must be at least 2 (kll_sketch.rs:167-169)


@pytest.mark.parametrize("dist", ["uniform", "normal", "lognormal", "sorted", "few_values"])
@pytest.mark.parametrize("n", [5_000, 300_000, 5_000_000])
def test_rank_error_within_stated_eps(dist, n):
    rng = np.random.default_rng(abs(hash((dist, n))) % 2**32)
    if dist == "uniform":
        vals = rng.random(n) * 1000
    elif dist == "normal":
        vals = rng.standard_normal(n)
    elif dist == "lognormal":
        vals = np.exp(rng.standard_normal(n) * 3)
    elif dist == "sorted":
        vals = np.arange(n, dtype=np.float64)
    else:
        vals = rng.integers(0, 7, size=n).astype(np.float64)
    mask = rng.random(n) >= 0.05
    vals[rng.random(n) < 0.01] = np.nan  # NaN is dropped like NULL
    validity = orc.pack_validity(mask)
    k = 200
    res, plan, st = run_plan([spec(T.KLL, 0, kll_k=k), spec(T.COUNT, 0)], [[numeric_column(vals, validity, True)]])
    kept = vals[mask & ~np.isnan(vals)]
    assert res[0].kll_n == len(kept)
    summ = st.kll_summary(0)
    assert summ["n"] == len(kept) and summ["min"] == kept.min() and summ["max"] == kept.max()
    assert total_weight(st, 0) == len(kept)  # weight-preserving compaction
    assert summ["num_retained"] <= 1024 * (summ["num_levels"] + 1)
    srt = np.sort(kept)
    eps = 1.65 / math.sqrt(k)
    worst = 0.0
    for phi in [0.01, 0.05, 0.25, 0.5, 0.75, 0.9, 0.95, 0.99]:
        q = st.kll_quantile(0, phi)
        worst = max(worst, rank_error(srt, q, phi))
    assert worst < eps
    assert worst < 0.02  # in practice the 512-item runs give ~0.1-1 %
    assert st.kll_quantile(0, 0.0) == kept.min() and st.kll_quantile(0, 1.0) == kept.max()
    # monotone quantiles within [min, max] (tests/advanced_analytics_integration.rs:100-116)
    qs = [st.kll_quantile(0, p) for p in (0.25, 0.5, 0.75, 0.95)]
    assert qs == sorted(qs) and kept.min() <= qs[0] and qs[-1] <= kept.max()


@pytest.mark.parametrize("dist", ["uniform", "sorted", "sawtooth", "reversed_blocks"])
@pytest.mark.parametrize("n,offset", [(20_000_000, 0), (33_554_432 + 12_345, 3)])
def test_sampled_batches_keep_exact_weight_and_tight_rank_error(dist, n, offset):
    """batches of 8 M rows and more go through the KLL sampler (one uniformly chosen value per 2^l consecutive
    values below the kept levels, kernels/kll.hip): total weight must still equal the number of values exactly
    and the rank error must stay an order of magnitude below the stated eps, whatever the arrangement --
    sorted input, a sawtooth whose period equals the sampling group, block-reversed input"""
    rng = np.random.default_rng(n % 1000 + len(dist))
    m = n + offset
    if dist == "uniform":
        vals = rng.random(m) * 1000
    elif dist == "sorted":
        vals = np.arange(m, dtype=np.float64)
    elif dist == "sawtooth":
        vals = (np.arange(m) % 4).astype(np.float64) * 1e6 + np.arange(m) * 1e-3  # period = group of 2^2 values
    else:
        idx = np.arange(m)
        vals = ((idx // 4096) * 4096 + (4095 - idx % 4096)).astype(np.float64)
    mask = rng.random(m) >= 0.03
    vals[rng.random(m) < 0.002] = np.nan
    validity = orc.pack_validity(mask)
    col = numeric_column(vals, validity, True, offset=offset, length=n)
    res, plan, st = run_plan([spec(T.KLL, 0, kll_k=200)], [[col]])
    v, mk = vals[offset:], mask[offset:]
    kept = v[mk & ~np.isnan(v)]
    assert res[0].kll_n == len(kept)
    assert total_weight(st, 0) == len(kept)
    summ = st.kll_summary(0)
    assert summ["min"] == kept.min() and summ["max"] == kept.max()
    srt = np.sort(kept)
    worst = max(rank_error(srt, st.kll_quantile(0, phi), phi) for phi in np.linspace(0.01, 0.99, 50))
    assert worst < 0.01, worst


def test_int64_column_and_batches_and_serialize():
    rng = np.random.default_rng(77)
    n = 1_200_000
    vals = rng.integers(-10**6, 10**6, size=n, dtype=np.int64)
    T.init()
    plan = T.Plan([spec(T.KLL, 0, kll_k=200), spec(T.NUMERIC_STATS, 0)])
    a, b = T.State(plan), T.State(plan)
    cut = [0, 100_000, 100_001, 650_000, n]
    for i, (lo, hi) in enumerate(zip(cut[:-1], cut[1:])):
        (a if i % 2 == 0 else b).update([numeric_column(vals, None, True, offset=lo, length=hi - lo)])
    other = T.State.deserialize(plan, b.serialize())
    a.merge([other])
    res = a.finalize()
    assert res[0].kll_n == n and res[1].non_null == n
    srt = np.sort(vals.astype(np.float64))
    assert total_weight(a, 0) == n
    for phi in (0.1, 0.5, 0.9, 0.99):
        assert rank_error(srt, a.kll_quantile(0, phi), phi) < 0.02


def test_against_oracle_within_reference_tolerance():
    """Same input through the oracle's restatement of the reference sketch: both answers must lie within the
    reference tests' own relative tolerance of the truth (kll_sketch.rs:459-468)."""
    vals = np.arange(1000, dtype=np.float64)
    sk = orc.Kll(100, 0)
    sk.update_many(vals)
    res, plan, st = run_plan([spec(T.KLL, 0, kll_k=100)], [[numeric_column(vals, None, True)]])
    for phi, truth in ((0.5, 500.0), (0.9, 900.0)):
        assert abs(sk.quantile(phi) - truth) / truth < 0.85
        assert abs(st.kll_quantile(0, phi) - truth) / truth < 0.85


# ---------------------------------------------------------------------------------------------------------------
# the fused path: a suite with range, quantile and correlation checks on the same columns reads them ONCE -- the KLL
# sampler and the co-moment lanes ride on the numeric scan (kernels/scan.hip: scan_kll_kernel, scan_pair_kernel)
@pytest.mark.parametrize("layout", ["aligned", "sliced", "misaligned"])
def test_sampler_and_comoments_riding_on_the_scan_match_the_oracle(layout):
    import torch

    rng = np.random.default_rng(31 + len(layout))
    n = 9_000_000 + 77  # big enough to be sampled (2^top = 2), ragged tail
    x = rng.standard_normal(n) * 50 + 10
    nan_at = rng.integers(0, n, 1000)
    x[nan_at] = np.nan                            # NaN: a value for MIN/MAX/COUNT, dropped by the sketch
    x[123] = -1e9
    x[n - 5] = 1e9                                # the extremes sit in the ragged tail / a tile
    y = 2 * x + rng.standard_normal(n)
    y[np.isnan(y)] = 7.0
    iv = rng.integers(-10**6, 10**6, size=n, dtype=np.int64)
    single = rng.random(n) * 100                  # a sampled column that is in no pair
    mx, my = rng.random(n) >= 0.05, rng.random(n) >= 0.2
    my[nan_at] = False                            # (no NaN among the rows both columns have: finite co-moments)
    off = {"aligned": 0, "sliced": 5, "misaligned": 1}[layout]
    pad = np.zeros(off)

    def dev(vals, mask):
        full = np.concatenate([pad.astype(vals.dtype), vals])
        t = torch.from_numpy(full).cuda()
        validity = None
        if mask is not None:
            validity = torch.from_numpy(np.concatenate([orc.pack_validity(np.concatenate([np.ones(off, bool), mask])),
                                                        np.zeros(64, np.uint8)])).cuda()
        if layout == "misaligned":  # the values buffer starts 8 bytes past a 16-byte boundary: no tiles
            ctor = T.Column.float64 if vals.dtype == np.float64 else T.Column.int64
            if validity is None:
                return ctor(t[1:], None, length=n)
            # bitmap bit 0 belongs to the dropped slot: keep the Arrow offset for the validity only via a shifted copy
            v2 = torch.from_numpy(np.concatenate([orc.pack_validity(mask), np.zeros(64, np.uint8)])).cuda()
            return ctor(t[1:], v2, length=n)
        ctor = T.Column.float64 if vals.dtype == np.float64 else T.Column.int64
        return ctor(t, validity, length=n, offset=off)

    cols = [dev(x, mx), dev(y, my), dev(iv, None), dev(single, mx)]
    specs = [spec(T.NUMERIC_STATS, 0), spec(T.COUNT, 1), spec(T.NUMERIC_STATS, 2), spec(T.NUMERIC_STATS, 3),
             spec(T.KLL, 0, kll_k=200), spec(T.KLL, 1, kll_k=200), spec(T.KLL, 2, kll_k=200), spec(T.KLL, 3, kll_k=200),
             spec(T.COMOMENTS, 0, column2=1), spec(T.COMOMENTS, 2, column2=3)]
    T.init()
    plan = T.Plan(specs)
    st = T.State(plan)
    st.profile_enable(True)
    st.update(cols)
    res = st.finalize()
    # nothing but the scan read the columns: no separate co-moment launch; the KLL work is the sketching of the picks
    assert st.profile_get("comoments")["launches"] == 0
    assert 0 < st.profile_get("kll")["bytes"] < 4 * n * 8
    vx, vy = orc.pack_validity(mx), orc.pack_validity(my)
    ox, oi, os_ = orc.stats(x, vx), orc.stats(iv, None), orc.stats(single, vx)
    assert (res[0].non_null, res[2].non_null, res[3].non_null) == (ox.non_null, n, os_.non_null)
    assert np.isnan(res[0].max_f) and res[0].min_f == ox.min_f   # totalOrder: NaN sorts last (arrow's total_cmp)
    assert (res[2].min_i, res[2].max_i, res[2].sum_i) == (oi.min_i, oi.max_i, oi.sum_i_wrapping)
    assert (res[3].min_f, res[3].max_f) == (os_.min_f, os_.max_f) and abs(res[3].sum_f - os_.sum_hi) <= 1e-9 * os_.sum_hi
    assert (res[1].total, res[1].non_null) == (n, int(my.sum()))
    for si, (vals, mask) in zip((4, 5, 6, 7), ((x, mx), (y, my), (iv.astype(np.float64), None), (single, mx))):
        kept = vals if mask is None else vals[mask]
        kept = np.sort(kept[~np.isnan(kept)])
        summ = st.kll_summary(si)
        assert (res[si].kll_n, summ["n"]) == (len(kept), len(kept)), si
        assert total_weight(st, si) == len(kept)                 # weight preserved exactly through the sampler
        assert (summ["min"], summ["max"]) == (kept[0], kept[-1])
        for phi in (0.01, 0.25, 0.5, 0.75, 0.95, 0.99):
            assert rank_error(kept, st.kll_quantile(si, phi), phi) < 1.65 / math.sqrt(200), (si, phi)
    for si, (a, b, va, vb) in zip((8, 9), ((x, y, vx, vy), (iv, single, None, vx))):
        o = orc.comoments(a, b, va, vb)
        r = res[si]
        assert (r.total, r.non_null) == (n, o.n)
        for got, want in ((r.sum_x, o.sum_x), (r.sum_y, o.sum_y), (r.sum_x2, o.sum_x2), (r.sum_y2, o.sum_y2),
                          (r.sum_xy, o.sum_xy)):
            if np.isnan(want):
                assert np.isnan(got)  # a NaN in a both-valid row poisons the sums, as in the oracle
            else:
                assert abs(got - want) <= 1e-9 * max(abs(want), 1.0)
    # a second batch on the same state (other size class: not sampled) and a merge keep the weights exact
    m = 3_000_000
    st.update([c.sliced(0, m) for c in cols])
    r2 = st.finalize()
    for si, (vals, mask) in zip((4, 7), ((x, mx), (single, mx))):
        kept = vals[:m][mask[:m]]
        assert r2[si].kll_n == res[si].kll_n + int((~np.isnan(kept)).sum())
        assert total_weight(st, si) == r2[si].kll_n


@pytest.mark.parametrize("case", ["zeros", "nan_payloads", "constant", "int_ties"])
def test_pair_scan_min_max_are_exact_in_total_order(case):
    """The pair kernel tests a value against the wave's running bounds and only runs the exact (IEEE totalOrder)
    MIN / MAX update for tiles that hold a value outside them: -0 / +0, NaNs of either sign and payload, ties with the
    current extreme and constant columns must come out bit for bit as the oracle's total_cmp says."""
    import struct
    import torch

    rng = np.random.default_rng({"zeros": 1, "nan_payloads": 2, "constant": 3, "int_ties": 4}[case])
    n = 1_500_000 + 13
    if case == "zeros":
        x = np.abs(rng.standard_normal(n)) + 1.0
        x[rng.integers(0, n, 50)] = 0.0          # +0 is the minimum ...
        x[700_001] = -0.0                        # ... until one -0 (equal to it for <, below it in totalOrder)
        y = -np.abs(rng.standard_normal(n)) - 1.0
        y[rng.integers(0, n, 50)] = -0.0         # -0 is the maximum until one +0
        y[900_003] = 0.0
    elif case == "nan_payloads":
        x = rng.standard_normal(n)
        bits = x.view(np.int64)
        bits[5] = 0x7FF8000000000001             # +NaN, small payload
        bits[1_000_000] = 0x7FFF00000000BEEF     # +NaN, bigger payload: the totalOrder maximum
        bits[77] = struct.unpack("<q", struct.pack("<Q", 0xFFF8000000000003))[0]  # -NaN: the totalOrder minimum
        y = rng.standard_normal(n) * 3
    elif case == "constant":
        x = np.full(n, 42.5)
        y = np.full(n, -7.0)
    else:
        x = rng.integers(-500, 500, size=n).astype(np.float64)
        y = rng.integers(0, 3, size=n).astype(np.float64)
    iv = rng.integers(-500, 500, size=n, dtype=np.int64) if case == "int_ties" else rng.integers(-10**12, 10**12, size=n, dtype=np.int64)
    mask = rng.random(n) >= 0.1
    validity = torch.from_numpy(np.concatenate([orc.pack_validity(mask), np.zeros(64, np.uint8)])).cuda()
    cols = [T.Column.float64(torch.from_numpy(x).cuda(), validity, length=n),
            T.Column.float64(torch.from_numpy(y).cuda(), None, length=n),
            T.Column.int64(torch.from_numpy(iv).cuda(), validity, length=n)]
    # the pair: the two Float64 columns; for "int_ties" the Int64 column with the first one
    pair = spec(T.COMOMENTS, 2, column2=0) if case == "int_ties" else spec(T.COMOMENTS, 0, column2=1)
    specs = [spec(T.NUMERIC_STATS, 0), spec(T.NUMERIC_STATS, 1), spec(T.NUMERIC_STATS, 2), pair]
    T.init()
    plan = T.Plan(specs)
    st = T.State(plan)
    st.profile_enable(True)
    st.update(cols)
    res = st.finalize()
    assert st.profile_get("comoments")["launches"] == 0  # the pair rode on the scan
    v = orc.pack_validity(mask)
    for r, (vals, vb) in zip(res[:3], ((x, v), (y, None), (iv, v))):
        o = orc.stats(vals, vb)
        assert r.non_null == o.non_null
        if vals.dtype == np.float64:
            got = (struct.pack("<d", r.min_f), struct.pack("<d", r.max_f))
            assert got == (struct.pack("<d", o.min_f), struct.pack("<d", o.max_f)), case
            if not np.isnan(o.sum_hi):
                assert abs(r.sum_f - o.sum_hi) <= 1e-9 * max(1.0, abs(o.sum_hi))
        else:
            assert (r.min_i, r.max_i, r.sum_i) == (o.min_i, o.max_i, o.sum_i_wrapping)
