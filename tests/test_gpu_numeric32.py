"""-m gpu: Int32 / Date32 / Float32 columns (include/tgx.h: TGX_INT32, TGX_FLOAT32) are widened on the device; every
check must give exactly what it gives on the widened Int64 / Float64 column (the oracle runs on the widened arrays)."""
import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import pad_validity, rel_err, run_plan, to_device
from test_gpu_parity import check_stats

pytestmark = pytest.mark.gpu


def col32(vals, validity, device, offset=0, length=None):
    v = pad_validity(validity)
    if device:
        vals_d, v_d = to_device(vals), to_device(v)
    else:
        vals_d, v_d = vals, v
    n = (len(vals) - offset) if length is None else length
    ctor = T.Column.int32 if vals.dtype == np.int32 else T.Column.float32
    return ctor(vals_d, v_d, length=n, offset=offset)


@pytest.mark.parametrize("device", [True, False])
@pytest.mark.parametrize("n", [1, 63, 4099, 300_001])
def test_checks_equal_the_widened_column(n, device):
    rng = np.random.default_rng(n + int(device))
    i32 = rng.integers(-2**31, 2**31, size=n, dtype=np.int64).astype(np.int32)
    i32[rng.random(n) < 0.3] = rng.integers(-50, 50)          # duplicates
    dates = rng.integers(0, 20_000, size=n, dtype=np.int64).astype(np.int32)  # Date32: days since the epoch
    f32 = (rng.standard_normal(n) * 1e3).astype(np.float32)
    f32[rng.random(n) < 0.05] = np.float32(-0.0)
    mask = rng.random(n) >= 0.1
    vb = orc.pack_validity(mask)
    cols = [(i32, vb), (dates, None), (f32, vb)]
    specs = []
    for ci in range(3):
        specs += [spec(T.COUNT, ci), spec(T.NUMERIC_STATS, ci, flags=T.FLAG_VARIANCE),
                  spec(T.DISTINCT, ci, flags=T.FLAG_MULTIPLICITY)]
    specs += [spec(T.COMOMENTS, 0, column2=2), spec(T.KLL, 2, kll_k=200)]
    # two batches, the second a slice with a non-zero Arrow offset that is no multiple of 64
    cut = n // 3
    batches = [[col32(v, b, device, offset=0, length=cut) for v, b in cols],
               [col32(v, b, device, offset=cut, length=n - cut) for v, b in cols]]
    res, _, st = run_plan(specs, batches)
    for ci, (v, b) in enumerate(cols):
        wide = v.astype(np.int64) if v.dtype == np.int32 else v.astype(np.float64)
        c = orc.count(b, n)
        assert (res[3 * ci].total, res[3 * ci].non_null) == (c.total, c.non_null)
        check_stats(res[3 * ci + 1], orc.stats(wide, b), variance=True)
        d = orc.distinct_bits64(wide.view(np.uint64), b, n=n)  # Float64 keys are compared by their bits
        r = res[3 * ci + 2]
        assert (r.total, r.non_null, r.distinct, r.groups_once) == (d.total, d.non_null, d.distinct, d.groups_once), ci
    want = orc.comoments(i32.astype(np.int64), f32.astype(np.float64), vb, vb)
    assert res[9].non_null == want.n and rel_err(res[9].sum_xy, want.sum_xy) < 1e-9
    assert res[10].kll_n == int(mask.sum())


@pytest.mark.parametrize("device", [True, False])
@pytest.mark.parametrize("n,lead", [(5, 0), (700, 3), (70_000, 64), (1_000_003, 129)])
def test_scan_reads_four_byte_values_in_place(n, lead, device):
    """A column only the scan needs (COUNT, min / max / sum / mean / variance) is not widened: the tile path loads its
    4-byte pairs directly (kernels/scan.hip: tile_load32).  Slices with odd and even offsets take the tile path or the
    per-lane path depending on the alignment of the first pair."""
    rng = np.random.default_rng(n + lead)
    total = n + lead + 5
    i32 = rng.integers(-2**31, 2**31, size=total, dtype=np.int64).astype(np.int32)
    f32 = (rng.standard_normal(total) * 1e4).astype(np.float32)
    f32[rng.random(total) < 0.01] = np.float32(np.inf)
    mask = rng.random(total) >= 0.07
    vb = orc.pack_validity(mask)
    cols = [(i32, vb), (f32, None), (f32, vb)]
    specs = []
    for ci in range(3):
        specs += [spec(T.COUNT, ci), spec(T.NUMERIC_STATS, ci, flags=T.FLAG_VARIANCE if ci != 1 else 0)]
    res, _, _ = run_plan(specs, [[col32(v, b, device, offset=lead, length=n) for v, b in cols]])
    for ci, (v, b) in enumerate(cols):
        wide = v.astype(np.int64) if v.dtype == np.int32 else v.astype(np.float64)
        c = orc.count(b, n, offset=lead)
        assert (res[2 * ci].total, res[2 * ci].non_null) == (c.total, c.non_null)
        want = orc.stats(wide, b, n=n, offset=lead)
        r = res[2 * ci + 1]
        assert (r.total, r.non_null) == (want.total, want.non_null)
        if wide.dtype == np.int64:
            assert (r.min_i, r.max_i, r.sum_i) == (want.min_i, want.max_i, want.sum_i_wrapping)
        else:
            assert (r.min_f, r.max_f) == (want.min_f, want.max_f)
            assert (not np.isfinite(want.sum_hi) and not np.isfinite(r.sum_f)) or rel_err(r.sum_f, want.sum_hi) < 1e-9
        if ci == 0 and want.has_variance:
            assert rel_err(r.var_samp, want.var_samp) < 1e-9


def test_from_arrow_32_bit_and_int64_shaped_types():
    import datetime

    import pyarrow as pa

    n = 5000
    rng = np.random.default_rng(3)
    ints = [None if rng.random() < 0.1 else int(rng.integers(-1000, 1000)) for _ in range(n)]
    days = [int(rng.integers(10_000, 20_000)) for _ in range(n)]
    micros = [None if rng.random() < 0.2 else int(rng.integers(0, 10**15)) for _ in range(n)]
    floats = [float(np.float32(rng.standard_normal())) for _ in range(n)]
    tbl = [pa.array(ints, type=pa.int32()), pa.array(days, type=pa.int32()).cast(pa.date32()),
           pa.array(micros, type=pa.int64()).cast(pa.timestamp("us")), pa.array(floats, type=pa.float32())]
    tbl = [a.slice(7, n - 11) for a in tbl]  # sliced arrays: Arrow offsets
    cols = [T.Column.from_arrow(a) for a in tbl]
    specs = []
    for ci in range(4):
        specs += [spec(T.NUMERIC_STATS, ci), spec(T.DISTINCT, ci)]
    res, _, _ = run_plan(specs, [cols])
    for ci, a in enumerate(tbl):
        py = a.cast(pa.int64() if ci < 3 else pa.float64()) if ci != 1 else a.cast(pa.int32()).cast(pa.int64())
        vals = [x for x in py.to_pylist() if x is not None]
        st, d = res[2 * ci], res[2 * ci + 1]
        assert (st.total, st.non_null) == (len(a), len(vals))
        assert d.distinct == len(set(vals))
        if ci < 3:
            assert (st.min_i, st.max_i, st.sum_i) == (min(vals), max(vals), sum(vals))
        else:
            assert (st.min_f, st.max_f) == (min(vals), max(vals)) and rel_err(st.sum_f, float(np.sum(np.array(vals)))) < 1e-9
    del datetime


def test_suite_over_32_bit_columns():
    """the suite front end (term_amd/suite.py -> host layer -> tgx) on a table of Int32 / Float32 / Date32 /
    Timestamp columns.  With `strict_reference_types(False)`: verdicts and metrics as on the Int64 / Float64 twin of the
    table (the widening deviation).  By default: what the reference does -- MIN / MAX of such a column come back in the
    column's own type, which StatisticalConstraint cannot read: "Failed to extract statistic value"
    (constraints/statistics.rs:277-308), a failed check; SUM (Int64) and AVG (Float64) of them are read."""
    import pyarrow as pa

    from term_amd.suite import Assertion, Check, Level, ValidationSuite

    n = 20_000
    rng = np.random.default_rng(21)
    qty = rng.integers(1, 51, size=n).astype(np.int32)
    price = (rng.random(n) * 100).astype(np.float32)
    day = rng.integers(8000, 12000, size=n).astype(np.int32)
    ts = rng.integers(10**15, 2 * 10**15, size=n).astype(np.int64)
    key = rng.permutation(n).astype(np.int32)

    def table(narrow):
        return pa.table({
            "qty": pa.array(qty, type=pa.int32() if narrow else pa.int64()),
            "price": pa.array(price if narrow else price.astype(np.float64), type=pa.float32() if narrow else pa.float64()),
            "day": pa.array(day, type=pa.int32()).cast(pa.date32()) if narrow else pa.array(day.astype(np.int64)),
            "ts": pa.array(ts, type=pa.int64()).cast(pa.timestamp("us")) if narrow else pa.array(ts),
            "key": pa.array(key, type=pa.int32() if narrow else pa.int64()),
        })

    check = (Check.builder("chk").level(Level.ERROR)
             .has_min("qty", Assertion.Equals(1.0)).has_max("qty", Assertion.Equals(50.0))
             .has_mean("price", Assertion.Between(45.0, 55.0)).has_sum("qty", Assertion.GreaterThan(0.0))
             .has_min("day", Assertion.GreaterThanOrEqual(8000.0)).has_max("ts", Assertion.LessThan(2.0e15))
             .validates_uniqueness(["key"], 1.0).validates_uniqueness(["qty"], 0.5).build())
    strict = ValidationSuite.builder("s").check(check).build()
    suite = ValidationSuite.builder("s").check(check).strict_reference_types(False).build()
    as_reference, twin = strict.run(table(True)), strict.run(table(False))
    errors = [i for i in as_reference.report.issues if i.message.startswith("Error evaluating constraint")]
    assert [(i.constraint_name, i.message) for i in errors] == [
        (name, "Error evaluating constraint: Internal error: Failed to extract statistic value")
        for name in ("min", "max", "min", "max")]   # qty Int32, qty Int32, day Date32, ts Timestamp
    assert as_reference.is_failure() and as_reference.report.metrics.failed_checks == 5
    kept = {k: v for k, v in twin.report.metrics.custom_metrics.items() if k not in ("chk.min", "chk.max")}
    assert as_reference.report.metrics.custom_metrics == kept   # mean(price Float32), sum(qty Int32), the uniqueness ratios
    narrow, wide = suite.run(table(True)), suite.run(table(False))
    assert narrow.report.metrics.custom_metrics == wide.report.metrics.custom_metrics
    assert [i.message for i in narrow.report.issues] == [i.message for i in wide.report.issues]
    assert narrow.report.metrics.passed_checks == wide.report.metrics.passed_checks
    assert len(narrow.report.issues) == 1 and "Uniqueness ratio" in narrow.report.issues[0].message  # qty repeats


@pytest.mark.parametrize("kind", ["growing_ids", "sparse_keys", "tuple"])
def test_retained_views_of_widened_device_columns_survive_later_batches(kind, monkeypatch):
    """A DEVICE Int32 key column is widened into per-update scratch; what a key set keeps for a later repair (keys
    outside the sampled range, overflowed lists) must not point there: the next update reuses the scratch
    (tgx_api.cpp: distinct_run_numeric `keep`, retained_numeric_view).  Three batches of growing ids: every batch
    after the first lies outside the range the first one's sample laid the bitmap out for."""
    if kind != "growing_ids":
        monkeypatch.setenv("TGX_FP_LISTS_MIN_ROWS", "1000")
    rng = np.random.default_rng(5)
    n = 100_000
    batches, arrays = [], []
    for b in range(3):
        if kind == "growing_ids":
            ids = (np.arange(n, dtype=np.int64) + b * 3 * n)
            ids[rng.random(n) < 0.01] = b * 3 * n + 7     # some repeats inside every batch
        else:
            ids = rng.integers(0, 2**31 - 1, size=n, dtype=np.int64)
            ids[: n // 2] = ids[0]                         # one heavy key: its list overflows
        arrays.append(ids.astype(np.int32))
    if kind == "tuple":
        specs = [spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY, columns=[0, 1])]
        second = [(a // 3).astype(np.int32) for a in arrays]
        batches = [[col32(a, None, True), col32(s2, None, True)] for a, s2 in zip(arrays, second)]
    else:
        specs = [spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)]
        batches = [[col32(a, None, True)] for a in arrays]
    res, _, _ = run_plan(specs, batches)
    if kind == "tuple":
        import collections
        keys = collections.Counter(zip(np.concatenate(arrays).tolist(), np.concatenate(second).tolist()))
        want = (3 * n, len(keys), sum(1 for v in keys.values() if v == 1))
        assert (res[0].total, res[0].distinct, res[0].groups_once) == want
    else:
        wide = np.concatenate(arrays).astype(np.int64)
        d = orc.distinct_bits64(wide.view(np.uint64), None, n=3 * n)
        assert (res[0].total, res[0].distinct, res[0].groups_once) == (d.total, d.distinct, d.groups_once)
