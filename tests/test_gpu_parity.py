"""-m gpu parity tests: libtgx (HIP kernels behind the C ABI) vs the CPU oracle on the same seeded
inputs. Counts / min / max / distinct are bit-exact; float aggregates within 1e-6 relative (north star),
in practice ~1e-15 because the device sums are compensated."""
import math
import zlib

import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import make_f64, make_i64, numeric_column, pad_validity, rel_err, run_plan, to_device

pytestmark = pytest.mark.gpu

TOL = 1e-6  # north star: float aggregates within 1e-6 relative


def check_stats(res, st, variance=False):
    assert res.total == st.total
    assert res.non_null == st.non_null
    assert bool(res.has_value) == bool(st.has_value)
    if not st.has_value:
        return
    if st.is_float:
        assert orc.nan_equal(res.min_f, st.min_f) and orc.nan_equal(res.max_f, st.max_f)
        assert math.copysign(1, res.min_f) == math.copysign(1, st.min_f)
        assert rel_err(res.sum_f, st.sum_hi) < TOL
    else:
        assert (res.min_i, res.max_i) == (st.min_i, st.max_i)
        assert res.sum_i == st.sum_i_wrapping
        assert rel_err(res.sum_f, st.sum_hi) < 1e-15
    assert rel_err(res.mean, st.sum_hi / st.non_null) < TOL
    assert rel_err(res.mean, st.mean) < TOL
    if variance:
        assert bool(res.has_variance) == bool(st.has_variance)
        if st.has_variance:
            assert rel_err(res.var_samp, st.var_samp) < TOL
            assert rel_err(res.stddev_samp, st.stddev_samp) < TOL


@pytest.mark.parametrize("device", [True, False])
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 511, 512, 513, 4096, 100_003, 1_000_000])
def test_numeric_stats_sizes(n, device):
    rng = np.random.default_rng(1000 + n)
    iv, ivv = make_i64(rng, n, -(2**40), 2**40, null_frac=0.05)
    fv, fvv = make_f64(rng, n, "uniform", null_frac=0.05)
    gv, _ = make_f64(rng, n, "normal", null_frac=0.0)
    cols = [numeric_column(iv, ivv, device), numeric_column(fv, fvv, device), numeric_column(gv, None, device)]
    specs = [spec(T.NUMERIC_STATS, 0, flags=T.FLAG_VARIANCE), spec(T.NUMERIC_STATS, 1, flags=T.FLAG_VARIANCE),
             spec(T.NUMERIC_STATS, 2), spec(T.COUNT, 0), spec(T.COUNT, 2)]
    res, _, _ = run_plan(specs, [cols])
    check_stats(res[0], orc.stats(iv, ivv), variance=True)
    check_stats(res[1], orc.stats(fv, fvv), variance=True)
    check_stats(res[2], orc.stats(gv, None))
    c = orc.count(ivv, n)
    assert (res[3].total, res[3].non_null) == (c.total, c.non_null)
    assert (res[4].total, res[4].non_null) == (n, n)


@pytest.mark.parametrize("offset", [0, 1, 7, 8, 63, 64, 65, 129, 1000])
def test_sliced_arrays(offset):
    """Arrow `offset` applies to validity bits and value slots alike; not a multiple of 8 or 64."""
    rng = np.random.default_rng(7 + offset)
    total = 20_000 + offset
    iv, ivv = make_i64(rng, total, -1000, 1000, null_frac=0.1)
    fv, fvv = make_f64(rng, total, "normal", null_frac=0.1)
    n = total - offset - 13
    cols = [numeric_column(iv, ivv, True, offset=offset, length=n),
            numeric_column(fv, fvv, True, offset=offset, length=n)]
    specs = [spec(T.NUMERIC_STATS, 0), spec(T.NUMERIC_STATS, 1), spec(T.COUNT, 1)]
    res, _, _ = run_plan(specs, [cols])
    check_stats(res[0], orc.stats(iv, ivv, n=n, offset=offset))
    check_stats(res[1], orc.stats(fv, fvv, n=n, offset=offset))
    c = orc.count(fvv, n, offset=offset)
    assert (res[2].total, res[2].non_null) == (c.total, c.non_null)


def test_count_only_columns():
    """COUNT on columns whose values are never read: validity popcount with ragged bit offsets."""
    import torch

    rng = np.random.default_rng(3)
    for n, offset in [(1, 0), (5, 3), (64, 0), (1000, 61), (100_000, 5), (1_000_003, 77)]:
        mask = rng.random(n + offset) >= 0.3
        validity = pad_validity(orc.pack_validity(mask))
        dv = to_device(validity)
        # 1-byte misaligned validity base pointer as well
        shifted = torch.zeros(len(validity) + 8, dtype=torch.uint8, device="cuda")
        shifted[1:1 + len(validity)] = dv
        cols = [T.Column(T.INT64, n, values=None, validity=dv, offset=offset),
                T.Column(T.INT64, n, values=None, validity=shifted[1:], offset=offset)]
        res, _, _ = run_plan([spec(T.COUNT, 0), spec(T.COUNT, 1)], [cols])
        c = orc.count(validity, n, offset=offset)
        for r in res:
            assert (r.total, r.non_null) == (c.total, c.non_null), (n, offset)


def test_special_float_values():
    vals = np.array([0.0, -0.0, 1.5, -2.5, float("inf"), 5e-324, -5e-324, 1e308, 1e308], dtype=np.float64)
    res, _, _ = run_plan([spec(T.NUMERIC_STATS, 0)], [[numeric_column(vals, None, True)]])
    st = orc.stats(vals)
    assert res[0].min_f == st.min_f and res[0].max_f == st.max_f == float("inf")
    assert res[0].sum_f == float("inf")
    # NaN sorts above +inf in IEEE totalOrder (arrow-arith aggregate); -0.0 below +0.0
    vals = np.array([0.0, -0.0, float("nan"), 3.0], dtype=np.float64)
    res, _, _ = run_plan([spec(T.NUMERIC_STATS, 0)], [[numeric_column(vals, None, True)]])
    assert math.isnan(res[0].max_f) and res[0].min_f == 0.0 and math.copysign(1, res[0].min_f) == -1
    assert math.isnan(res[0].sum_f)
    # all NULL => SQL NULL aggregates
    vals = np.zeros(100, dtype=np.float64)
    validity = orc.pack_validity(np.zeros(100, dtype=bool))
    res, _, _ = run_plan([spec(T.NUMERIC_STATS, 0)], [[numeric_column(vals, validity, True)]])
    assert res[0].total == 100 and res[0].non_null == 0 and not res[0].has_value


def test_int64_extremes_and_wrapping_sum():
    vals = np.array([2**62, 2**62, 2**62, -(2**63), 2**63 - 1, -5], dtype=np.int64)
    res, _, _ = run_plan([spec(T.NUMERIC_STATS, 0)], [[numeric_column(vals, None, True)]])
    st = orc.stats(vals)
    assert res[0].sum_i == st.sum_i_wrapping
    assert (res[0].min_i, res[0].max_i) == (-(2**63), 2**63 - 1)
    exact = sum(int(v) for v in vals)
    assert res[0].sum_f == float(exact)


def test_ill_conditioned_sum_is_still_accurate():
    """Compensated device sums: large cancelling terms do not destroy a small true total."""
    rng = np.random.default_rng(11)
    big = rng.standard_normal(200_000) * 1e12
    vals = np.concatenate([big, -big, np.full(1000, 0.125)])
    rng.shuffle(vals)
    res, _, _ = run_plan([spec(T.NUMERIC_STATS, 0)], [[numeric_column(vals, None, True)]])
    assert rel_err(res[0].sum_f, 125.0) < 1e-9


@pytest.mark.parametrize("n_batches", [1, 3, 17])
def test_multi_batch_equals_single_pass(n_batches):
    rng = np.random.default_rng(99)
    n = 50_000
    iv, ivv = make_i64(rng, n, 0, 5000, null_frac=0.02)
    fv, fvv = make_f64(rng, n, "uniform", null_frac=0.02)
    specs = [spec(T.NUMERIC_STATS, 0, flags=T.FLAG_VARIANCE), spec(T.NUMERIC_STATS, 1, flags=T.FLAG_VARIANCE),
             spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.COMOMENTS, 0, column2=1)]
    bounds = np.linspace(0, n, n_batches + 1).astype(int)
    batches = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        batches.append([numeric_column(iv, ivv, True, offset=int(a), length=int(b - a)),
                        numeric_column(fv, fvv, True, offset=int(a), length=int(b - a))])
    res, _, _ = run_plan(specs, batches)
    check_stats(res[0], orc.stats(iv, ivv), variance=True)
    check_stats(res[1], orc.stats(fv, fvv), variance=True)
    d = orc.distinct_bits64(iv, ivv)
    assert (res[2].total, res[2].non_null, res[2].distinct, res[2].groups_once) == \
        (d.total, d.non_null, d.distinct, d.groups_once)
    cm = orc.comoments(iv, fv, ivv, fvv)
    assert res[3].non_null == cm.n
    for got, want in [(res[3].sum_x, cm.sum_x), (res[3].sum_y, cm.sum_y), (res[3].sum_x2, cm.sum_x2),
                      (res[3].sum_y2, cm.sum_y2), (res[3].sum_xy, cm.sum_xy)]:
        assert rel_err(got, want) < TOL


@pytest.mark.parametrize("case", ["bitmap_dense", "bitmap_dups", "hash_wide", "hash_float", "permutation",
                                  "all_ones_key", "all_null", "single_null"])
def test_distinct_modes(case):
    rng = np.random.default_rng(hash(case) % 2**32)
    n = 200_000
    validity = None
    if case == "bitmap_dense":
        vals, validity = make_i64(rng, n, 0, n // 2, null_frac=0.05)
    elif case == "bitmap_dups":
        vals, validity = make_i64(rng, n, -50, 50, null_frac=0.5)
    elif case == "hash_wide":
        vals, validity = make_i64(rng, n, -(2**62), 2**62, null_frac=0.05)
        vals[: n // 4] = vals[n // 4: n // 2]  # duplicates
    elif case == "hash_float":
        f = np.round(rng.standard_normal(n), 2)
        f[:10] = 0.0
        f[10:20] = -0.0
        vals = f.view(np.int64).copy()
    elif case == "permutation":
        vals = rng.permutation(n).astype(np.int64)
    elif case == "all_ones_key":
        vals = rng.integers(-3, 3, size=n, dtype=np.int64)  # includes -1 = 0xFFFF...FFFF
        vals = np.concatenate([vals, np.array([-(2**62), 2**62 - 1], dtype=np.int64)])  # force hash mode
        n = len(vals)
    elif case == "all_null":
        vals = np.zeros(n, dtype=np.int64)
        validity = orc.pack_validity(np.zeros(n, dtype=bool))
    elif case == "single_null":
        vals = np.arange(n, dtype=np.int64)
        mask = np.ones(n, dtype=bool)
        mask[12345] = False
        validity = orc.pack_validity(mask)
    if case == "hash_float":
        col = numeric_column(vals.view(np.float64), None, True)
    else:
        col = numeric_column(vals, validity, True)
    res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)], [[col]])
    d = orc.distinct_bits64(vals.view(np.uint64), validity, n=n)
    got = (res[0].total, res[0].non_null, res[0].distinct, res[0].groups_once)
    assert got == (d.total, d.non_null, d.distinct, d.groups_once), case


@pytest.mark.parametrize("mult", [False, True])
@pytest.mark.parametrize("shape", ["uniform", "skewed", "permutation", "two_batches"])
def test_distinct_partitioned_bitmap(shape, mult):
    """>= 2^20-row batches over a dense range take the bucket + LDS-slice path; skew spills to atomics."""
    rng = np.random.default_rng(abs(hash((shape, mult))) % 2**32)
    n = 3_000_017
    validity = None
    if shape == "uniform":
        vals, validity = make_i64(rng, n, -1_000_000, 1_500_000, null_frac=0.05)
    elif shape == "skewed":
        vals = rng.integers(0, 4_000_000, size=n, dtype=np.int64)
        hot = rng.random(n) < 0.9
        vals[hot] = rng.integers(2_000_000, 2_000_512, size=int(hot.sum()), dtype=np.int64)
    elif shape == "permutation":
        vals = rng.permutation(n).astype(np.int64) + 10**12
    else:
        vals = rng.integers(0, 3_000_000, size=n, dtype=np.int64)
    flags = T.FLAG_MULTIPLICITY if mult else 0
    if shape == "two_batches":
        half = n // 2
        batches = [[numeric_column(vals, None, True, offset=0, length=half)],
                   [numeric_column(vals, None, True, offset=half, length=n - half)]]
    else:
        batches = [[numeric_column(vals, validity, True)]]
    res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=flags), spec(T.NUMERIC_STATS, 0)], batches, hint=n)
    d = orc.distinct_bits64(vals.view(np.uint64), validity, n=n)
    assert (res[0].total, res[0].non_null, res[0].distinct) == (d.total, d.non_null, d.distinct)
    if mult:
        assert res[0].groups_once == d.groups_once
    check_stats(res[1], orc.stats(vals, validity))


def test_distinct_bitmap_escapes_to_hash_on_out_of_range_batch():
    rng = np.random.default_rng(5)
    a = rng.integers(0, 1000, size=50_000, dtype=np.int64)
    b = rng.integers(10**12, 10**12 + 1000, size=50_000, dtype=np.int64)
    c = rng.integers(-500, 1500, size=50_000, dtype=np.int64)
    res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.NUMERIC_STATS, 0)],
                         [[numeric_column(x, None, True)] for x in (a, b, c)])
    allv = np.concatenate([a, b, c])
    d = orc.distinct_bits64(allv.view(np.uint64))
    assert (res[0].distinct, res[0].groups_once) == (d.distinct, d.groups_once)
    check_stats(res[1], orc.stats(allv))


def test_hash_growth_across_batches():
    rng = np.random.default_rng(6)
    parts = [rng.integers(-(2**60), 2**60, size=30_000, dtype=np.int64) for _ in range(8)]
    parts[5][:1000] = parts[0][:1000]
    res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)],
                         [[numeric_column(p, None, True)] for p in parts])
    d = orc.distinct_bits64(np.concatenate(parts).view(np.uint64))
    assert (res[0].distinct, res[0].groups_once) == (d.distinct, d.groups_once)


def test_comoments_and_correlation_formulas(golden):
    c = golden["correlation"]
    x = np.arange(c["n"], dtype=np.float64)
    y = 2.0 * x + 1.0
    res, _, _ = run_plan([spec(T.COMOMENTS, 0, column2=1)],
                         [[numeric_column(x, None, True), numeric_column(y, None, True)]])
    st = orc.Comoments(int(res[0].non_null), res[0].sum_x, res[0].sum_y, res[0].sum_x2, res[0].sum_y2,
                       res[0].sum_xy)
    assert abs(orc.pearson(st) - 1.0) < c["pearson"]["tol"]
    assert c["covariance"]["lo"] < orc.covariance(st) < c["covariance"]["hi"]
    # mixed int64 / float64 with nulls on both sides
    rng = np.random.default_rng(8)
    n = 300_001
    iv, ivv = make_i64(rng, n, -10_000, 10_000, null_frac=0.1)
    fv, fvv = make_f64(rng, n, "normal", null_frac=0.1)
    res, _, _ = run_plan([spec(T.COMOMENTS, 0, column2=1)],
                         [[numeric_column(iv, ivv, True), numeric_column(fv, fvv, True)]])
    cm = orc.comoments(iv, fv, ivv, fvv)
    assert res[0].non_null == cm.n and res[0].total == n
    for got, want in [(res[0].sum_x, cm.sum_x), (res[0].sum_y, cm.sum_y), (res[0].sum_x2, cm.sum_x2),
                      (res[0].sum_y2, cm.sum_y2), (res[0].sum_xy, cm.sum_xy)]:
        assert rel_err(got, want) < TOL
    # sliced views: every combination of even / odd Arrow offsets (the wide path reads validity per row pair) and
    # 16-byte aligned / unaligned value pointers
    for xo, yo, m in ((0, 0, 1001), (2, 4, 70_000), (1, 1, 70_001), (3, 8, 99_999), (6, 5, 64)):
        res, _, _ = run_plan([spec(T.COMOMENTS, 0, column2=1)],
                             [[numeric_column(iv, ivv, True, offset=xo, length=m),
                               numeric_column(fv, fvv, True, offset=yo, length=m)]])
        cm = orc.comoments(iv[xo:xo + m].copy(), fv[yo:yo + m].copy(),
                           orc.pack_validity(orc.unpack_validity(ivv, n)[xo:xo + m]),
                           orc.pack_validity(orc.unpack_validity(fvv, n)[yo:yo + m]))
        assert res[0].non_null == cm.n and res[0].total == m, (xo, yo, m)
        for got, want in [(res[0].sum_x, cm.sum_x), (res[0].sum_y, cm.sum_y), (res[0].sum_xy, cm.sum_xy)]:
            assert rel_err(got, want) < TOL, (xo, yo, m)


def test_reference_known_answer_vectors_numeric(golden):
    """The reference's own unit-test vectors through the HIP path (host-resident Arrow buffers)."""
    for case in golden["statistics"]:
        vals, validity = orc.column_from_list(case["values"], np.float64)
        res, _, _ = run_plan([spec(T.NUMERIC_STATS, 0)], [[numeric_column(vals, validity, False)]])
        r = res[0]
        if case["status"] == "failure" and "message_contains" in case:
            assert not r.has_value
            continue
        got = {"mean": r.mean, "min": r.min_f, "max": r.max_f, "sum": r.sum_f}[case["stat"]]
        assert got == case["metric"], case["ref"]
    for case in golden["completeness"]:
        for cname in case["cols"]:
            vals, validity = orc.column_from_list(case["columns"][cname], np.int64)
            if len(vals) == 0:
                continue
            res, _, _ = run_plan([spec(T.COUNT, 0)], [[numeric_column(vals, validity, False)]])
            assert res[0].total == len(vals)
            assert res[0].non_null == sum(v is not None for v in case["columns"][cname])
    a = golden["analyzers"]
    ids, idv = orc.column_from_list(a["table"]["id"], np.int64)
    vals, vv = orc.column_from_list(a["table"]["value"], np.float64)
    res, _, _ = run_plan([spec(T.COUNT, 0), spec(T.NUMERIC_STATS, 1), spec(T.DISTINCT, 0)],
                         [[numeric_column(ids, idv, False), numeric_column(vals, vv, False)]])
    assert (res[0].total, res[0].non_null) == (5, 4)
    assert (res[1].sum_f, res[1].non_null, res[1].mean, res[1].min_f, res[1].max_f) == (100.0, 4, 25.0, 10.0, 40.0)
    assert res[2].distinct == 4


def test_merge_and_serialize_roundtrip():
    rng = np.random.default_rng(21)
    n = 120_000
    iv, ivv = make_i64(rng, n, 0, 40_000, null_frac=0.03)
    fv, fvv = make_f64(rng, n, "uniform", null_frac=0.03)
    specs = [spec(T.NUMERIC_STATS, 0, flags=T.FLAG_VARIANCE), spec(T.NUMERIC_STATS, 1), spec(T.COUNT, 1),
             spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.COMOMENTS, 0, column2=1)]
    T.init()
    plan = T.Plan(specs)
    half = n // 2 + 17
    states = []
    for a, b in [(0, half), (half, n)]:
        st = T.State(plan)
        st.update([numeric_column(iv, ivv, True, offset=a, length=b - a),
                   numeric_column(fv, fvv, True, offset=a, length=b - a)])
        states.append(st)
    # serialize -> deserialize -> merge (AnalyzerState::merge semantics, exact for DISTINCT)
    blob = states[1].serialize()
    other = T.State.deserialize(plan, blob)
    states[0].merge([other])
    res = states[0].finalize()
    check_stats(res[0], orc.stats(iv, ivv), variance=True)
    check_stats(res[1], orc.stats(fv, fvv))
    d = orc.distinct_bits64(iv, ivv)
    assert (res[3].total, res[3].non_null, res[3].distinct, res[3].groups_once) == \
        (d.total, d.non_null, d.distinct, d.groups_once)
    cm = orc.comoments(iv, fv, ivv, fvv)
    assert res[4].non_null == cm.n and rel_err(res[4].sum_xy, cm.sum_xy) < TOL


def test_distinct_owner_exchange_single_process():
    """The cross-rank exact-distinct protocol with both 'ranks' in one process: export by owner,
    swap the runs, import, then merge the owner-partitioned counts."""
    import torch

    rng = np.random.default_rng(31)
    n = 100_000
    vals = rng.integers(0, 60_000, size=n, dtype=np.int64)
    wide = rng.integers(-(2**61), 2**61, size=n, dtype=np.int64)
    for data in (vals, wide):
        T.init()
        plan = T.Plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)])
        world = 2
        shards = [data[: n // 3], data[n // 3:]]
        states, exported = [], []
        for sh in shards:
            st = T.State(plan)
            st.update([numeric_column(sh, None, True)])
            ptr, counts = st.distinct_export(0, world)
            # copy the runs out: the export buffer belongs to the state
            total = sum(counts)
            recs = torch.empty(total * 2, dtype=torch.int64, device="cuda")
            import ctypes

            torch.cuda.synchronize()
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipMemcpy(ctypes.c_void_p(recs.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(total * 16),
                          ctypes.c_int(3))
            states.append(st)
            exported.append((recs, counts))
        for r in range(world):
            parts = []
            for recs, counts in exported:
                start = sum(counts[:r])
                parts.append(recs[2 * start: 2 * (start + counts[r])])
            mine = torch.cat(parts).contiguous()
            torch.cuda.synchronize()
            states[r].distinct_import(0, mine.data_ptr(), mine.numel() // 2)
        states[0].merge([states[1]])
        res = states[0].finalize()
        d = orc.distinct_bits64(data.view(np.uint64))
        assert (res[0].total, res[0].non_null, res[0].distinct, res[0].groups_once) == \
            (d.total, d.non_null, d.distinct, d.groups_once)


def test_errors_are_reported_not_thrown():
    T.init()
    plan = T.Plan([spec(T.NUMERIC_STATS, 0)])
    st = T.State(plan)
    with pytest.raises(T.TgxError) as e:
        st.update([])
    assert e.value.status == "TGX_INVALID_ARGUMENT"
    a = np.arange(10, dtype=np.int64)
    plan2 = T.Plan([spec(T.NUMERIC_STATS, 0), spec(T.NUMERIC_STATS, 1)])
    st2 = T.State(plan2)
    with pytest.raises(T.TgxError):
        st2.update([numeric_column(a, None, True), numeric_column(a[:5].copy(), None, True)])  # ragged batch
    with pytest.raises(T.TgxError):
        T.Plan([spec(99, 0)])


@pytest.mark.parametrize("mult", [False, True])
def test_distinct_bitmap_slice_exchange_single_process(mult):
    """the range-bitmap form of the cross-rank exchange with three 'ranks' in one process: agree on [lo, hi],
    build congruent bitmaps, swap equal slices, OR + popcount the owned slice, merge the counts"""
    import torch

    rng = np.random.default_rng(77)
    n, world = 3_500_000, 3
    vals = rng.integers(-200_000, 2_300_000, size=n, dtype=np.int64)
    vals[: n // 5] = vals[n // 5: 2 * n // 5]  # cross-shard duplicates
    mask = rng.random(n) >= 0.03
    validity = orc.pack_validity(mask)
    flags = T.FLAG_MULTIPLICITY if mult else 0
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0, flags=flags)])
    bounds = [0, 1_000_000 // 64 * 64, 2_400_000 // 64 * 64, n]
    lo, hi = int(vals[mask].min()), int(vals[mask].max())
    states, views = [], []
    for r in range(world):
        st = T.State(plan)
        st.distinct_range_hint(0, lo, hi)
        a, b = bounds[r], bounds[r + 1]
        st.update([numeric_column(vals, validity, True, offset=a, length=b - a)])
        states.append(st)
        views.append(st.distinct_bitmap_view(0))
    base, n_words = views[0][0], views[0][1]
    assert all(v[0] == base and v[1] == n_words for v in views)  # congruent bitmaps
    slice_words = ((n_words + world - 1) // world + 3) // 4 * 4
    padded = slice_words * world

    class P:
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    def padded_copy(ptr):
        t = torch.zeros(padded, dtype=torch.int32, device="cuda")
        t[:n_words] = torch.as_tensor(P(ptr, n_words * 4), device="cuda").view(torch.int32)
        return t

    seen = [padded_copy(v[2]) for v in views]
    twice = [padded_copy(v[3]) for v in views] if mult else None
    for r in range(world):
        recv_seen = torch.cat([s[r * slice_words:(r + 1) * slice_words] for s in seen]).contiguous()
        recv_twice = torch.cat([s[r * slice_words:(r + 1) * slice_words] for s in twice]).contiguous() if mult else None
        torch.cuda.synchronize()
        if r % 2 == 0:
            states[r].distinct_adopt_slices(0, base + r * slice_words * 32, recv_seen.data_ptr(),
                                            recv_twice.data_ptr() if mult else None, world, slice_words)
        else:
            # strided form: seen and twice slices interleaved per peer, as an all-to-all of several parts leaves them
            both = torch.stack([recv_seen.view(world, slice_words),
                                (recv_twice if mult else recv_seen).view(world, slice_words)], dim=1).contiguous()
            torch.cuda.synchronize()
            states[r].distinct_adopt_slices(0, base + r * slice_words * 32, both.data_ptr(),
                                            both.data_ptr() + 4 * slice_words if mult else None, world, slice_words,
                                            2 * slice_words)
    blobs = [s.serialize() for s in states]
    assert all(len(b) < 220 for b in blobs)  # owner-partitioned states travel as counts only
    merged = T.State.deserialize(plan, blobs[0])
    merged.merge([T.State.deserialize(plan, b) for b in blobs[1:]])
    res = merged.finalize()
    d = orc.distinct_bits64(vals.view(np.uint64), validity)
    assert (res[0].total, res[0].non_null, res[0].distinct) == (d.total, d.non_null, d.distinct)
    if mult:
        assert res[0].groups_once == d.groups_once


def test_range_hint_violation_is_reported_not_miscounted():
    vals = np.arange(2_000_000, dtype=np.int64)
    vals[123_456] = 5_000_000_000  # outside the declared range
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0)])
    st = T.State(plan)
    st.distinct_range_hint(0, 0, 1_999_999)
    st.update([numeric_column(vals, None, True)])
    with pytest.raises(T.TgxError) as e:
        st.finalize()
    assert "outside the range bitmap" in str(e.value)
    with pytest.raises(T.TgxError):
        st.distinct_range_hint(0, 0, 10)  # only before the first batch


def test_several_distinct_columns_share_one_range_readback():
    """Int64 columns decide bitmap vs hash from the scan's running MIN / MAX; with two or more of them the
    accumulators are read back once per update (tgx_api.cpp: scan_snapshot).  Four columns with different fates over
    three batches: dense range (bitmap), a range that widens in the second batch (bitmap regrown / escapes), sparse
    (hash), and one that is all-NULL in the first batch (decides later)."""
    rng = np.random.default_rng(77)
    n = 3 * 120_000
    dense = rng.integers(-40_000, 90_000, size=n, dtype=np.int64)
    widening = rng.integers(0, 50_000, size=n, dtype=np.int64)
    widening[120_000:240_000] += 10**9
    sparse = rng.integers(-(2**61), 2**61, size=n, dtype=np.int64)
    sparse[200_000:210_000] = sparse[:10_000]
    late = rng.integers(5, 5000, size=n, dtype=np.int64)
    late_valid = np.ones(n, dtype=bool)
    late_valid[:120_000] = False
    late_valid[rng.random(n) < 0.1] = False
    cols = [(dense, None), (widening, None), (sparse, None), (late, orc.pack_validity(late_valid))]
    specs = []
    for ci in range(4):
        specs += [spec(T.DISTINCT, ci, flags=T.FLAG_MULTIPLICITY if ci % 2 else 0), spec(T.NUMERIC_STATS, ci)]
    batches = [[numeric_column(v, b, True, offset=lo, length=120_000) for v, b in cols] for lo in (0, 120_000, 240_000)]
    res, _, _ = run_plan(specs, batches)
    for ci, (v, b) in enumerate(cols):
        d = orc.distinct_bits64(v.view(np.uint64), b, n=n)
        r = res[2 * ci]
        assert (r.total, r.non_null, r.distinct) == (d.total, d.non_null, d.distinct), ci
        if ci % 2:
            assert r.groups_once == d.groups_once, ci
        check_stats(res[2 * ci + 1], orc.stats(v, b))


# ---------------------------------------------------------------------------------------------------------------
# The range bitmap of a dense Int64 key column is laid out from a SAMPLE of the first batch (so that no scan has to
# finish first and the DISTINCT pass can take the column's MIN / MAX / SUM / COUNT along); keys the sample did not
# announce are counted by the kernels and brought in when the host next looks at the state.
def _dense_with_outliers(rng, n, n_out, dup_outliers):
    vals = rng.permutation(n).astype(np.int64) + 1000
    # rows the 2^16-row sample does not visit: it reads row k * step, every 16th one half a step later
    step = n // 65536
    rows = [r for r in rng.integers(0, n, 4 * n_out + 64).tolist() if r % step not in (0, step // 2)][:n_out]
    far = np.array([10**15, -10**15, -1, 2**62, -2**62, 10**15 + 1], dtype=np.int64)  # -1: the all-ones pattern
    for i, r in enumerate(rows):
        vals[r] = far[i % len(far)] if dup_outliers else far[i % len(far)] + i // len(far) * 7
    return vals


@pytest.mark.parametrize("device", [True, False])
@pytest.mark.parametrize("mult", [False, True])
def test_keys_outside_the_sampled_range_are_repaired(device, mult):
    rng = np.random.default_rng(41 + mult)
    n = 3_000_000 + 17
    vals = _dense_with_outliers(rng, n, 40, dup_outliers=mult)
    mask = rng.random(n) >= 0.02
    validity = orc.pack_validity(mask)
    flags = T.FLAG_MULTIPLICITY if mult else 0
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0, flags=flags), spec(T.NUMERIC_STATS, 0), spec(T.COUNT, 0)])
    st = T.State(plan)
    st.profile_enable(True)
    st.update([numeric_column(vals, validity, device)])
    res = st.finalize()
    d = orc.distinct_bits64(vals.view(np.uint64), validity)
    o = orc.stats(vals, validity)
    assert (res[0].total, res[0].non_null, res[0].distinct) == (d.total, d.non_null, d.distinct)
    if mult:
        assert res[0].groups_once == d.groups_once
    # the aggregates came out of the DISTINCT pass (the scan did not read the column) and are exact all the same
    assert st.profile_get("scan")["launches"] == 0
    assert (res[1].non_null, res[1].min_i, res[1].max_i, res[1].sum_i) == (o.non_null, o.min_i, o.max_i, o.sum_i_wrapping)
    assert (res[2].total, res[2].non_null) == (n, o.non_null)
    # finalize is repeatable, and a later batch (outliers of its own) joins the repaired set
    again = st.finalize()
    assert (again[0].distinct, again[0].groups_once) == (res[0].distinct, res[0].groups_once)
    more = _dense_with_outliers(rng, n, 25, dup_outliers=False) + 500_000
    st.update([numeric_column(more, None, device)])
    both = np.concatenate([vals, more])
    both_valid = orc.pack_validity(np.concatenate([mask, np.ones(n, bool)]))
    d2 = orc.distinct_bits64(both.view(np.uint64), both_valid)
    r2 = st.finalize()
    assert (r2[0].total, r2[0].non_null, r2[0].distinct) == (d2.total, d2.non_null, d2.distinct)
    if mult:
        assert r2[0].groups_once == d2.groups_once


def test_outliers_in_a_later_batch_and_through_merge_and_serialize():
    rng = np.random.default_rng(7)
    n = 2_500_000
    clean = rng.permutation(n).astype(np.int64)
    dirty = _dense_with_outliers(rng, n, 30, False)
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)])
    a, b = T.State(plan), T.State(plan)
    a.update([numeric_column(clean, None, True)])      # the bitmap is laid out from this batch ...
    a.update([numeric_column(dirty, None, True)])      # ... and this one brings keys far outside it
    b.update([numeric_column(dirty[::-1].copy(), None, True)])
    blob = a.serialize()                               # a resolve point: the blob carries the repaired set
    c = T.State.deserialize(plan, blob)
    c.merge([b])
    want = orc.distinct_bits64(np.concatenate([clean, dirty, dirty]).view(np.uint64), None)
    got = c.finalize()[0]
    assert (got.total, got.distinct, got.groups_once) == (want.total, want.distinct, want.groups_once)
    one = orc.distinct_bits64(np.concatenate([clean, dirty]).view(np.uint64), None)
    ra = a.finalize()[0]
    assert (ra.distinct, ra.groups_once) == (one.distinct, one.groups_once)


@pytest.mark.parametrize("with_stats", [True, False])
@pytest.mark.parametrize("shape", ["shuffled", "two_batches_nulls", "skewed"])
def test_distinct_partitioned_bitmap_20_bit_entries(shape, with_stats, monkeypatch):
    """A dense range of more than 2048 x 2^16 values (no 2-byte entries) without multiplicity: the bucket lists hold 20-bit
    entries, three to an 8-byte word, runs padded to 24 entries by repeating their last key (kernels/distinct.hip, PACK20;
    the keys-in-order form of the same lists: tests/test_gpu_ordered_keys.py `ascending_step64`).  Bit-exact against the
    oracle, and equal to the 4-byte lists (TGX_PACK20=0)."""
    rng = np.random.default_rng(zlib.crc32(shape.encode()) + int(with_stats))
    n = 3_200_011
    validity = None
    vals = rng.permutation(n).astype(np.int64) * 61 - 7_000_000_000   # 195 M values wide, all distinct
    if shape == "two_batches_nulls":
        vals[rng.random(n) < 0.02] = vals[5]                            # repeats
        validity = orc.pack_validity(rng.random(n) >= 0.07)
    elif shape == "skewed":                                             # 70 % of the rows in two buckets: lists overflow, runs spill
        hot = rng.random(n) < 0.7
        vals[hot] = -7_000_000_000 + rng.integers(50_000_000, 50_000_000 + (1 << 21), size=int(hot.sum()), dtype=np.int64)
    specs = [spec(T.DISTINCT, 0)] + ([spec(T.NUMERIC_STATS, 0)] if with_stats else [spec(T.COUNT, 0)])
    if shape == "two_batches_nulls":
        cut = 1_700_003
        batches = [[numeric_column(vals, validity, True, offset=0, length=cut)],
                   [numeric_column(vals, validity, True, offset=cut, length=n - cut)]]
    else:
        batches = [[numeric_column(vals, validity, True)]]
    res, _, st = run_plan(specs, batches, hint=n)
    assert st.profile_get("distinct")["launches"] >= 0
    d = orc.distinct_bits64(vals.view(np.uint64), validity, n=n)
    assert (res[0].total, res[0].non_null, res[0].distinct) == (d.total, d.non_null, d.distinct)
    if with_stats:
        check_stats(res[1], orc.stats(vals, validity))
    monkeypatch.setenv("TGX_PACK20", "0")   # the same batches through 4-byte entries
    again, _, _ = run_plan(specs, batches, hint=n)
    assert (again[0].distinct, again[0].non_null) == (res[0].distinct, res[0].non_null)
