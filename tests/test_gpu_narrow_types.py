"""-m gpu: Arrow types beyond the eight of rounds 1-4 (round-4 verdict, "missing" 4): Int8 / Int16 / UInt8 / UInt16 /
UInt32 columns are widened to Int64 on the device, UInt64 columns are read in place as keys, Boolean columns are
bit-packed -- the reference's completeness / uniqueness SQL takes any column type (constraints/completeness.rs:158-163,
uniqueness.rs:612-617).  Oracle leg: the same logical values widened to int64 by numpy, through the oracle's Int64
functions; every count, extreme, sum and distinct count bit-exact."""
import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import pad_validity, rel_err, run_plan, to_device

pytestmark = pytest.mark.gpu

NARROW = [(T.INT8, np.int8), (T.INT16, np.int16), (T.UINT8, np.uint8), (T.UINT16, np.uint16), (T.UINT32, np.uint32)]


def narrow_column(type_id, vals, validity, device, offset=0, length=None):
    validity = pad_validity(validity)
    pad = np.zeros(64, dtype=vals.dtype)
    vals = np.concatenate([vals, pad])
    n = (len(vals) - 64 - offset) if length is None else length
    if device:
        import torch

        v = torch.from_numpy(vals.view(np.uint8)).cuda()
        b = to_device(validity)
    else:
        v, b = vals, validity
    return T.Column(type_id, n, values=v, validity=b, offset=offset)


def draw(rng, dtype, n, kind):
    info = np.iinfo(dtype)
    if kind == "full":
        return rng.integers(info.min, int(info.max) + 1, size=n, dtype=np.int64).astype(dtype)
    if kind == "few":
        return rng.integers(max(info.min, -3), min(int(info.max), 17), size=n, dtype=np.int64).astype(dtype)
    return np.full(n, info.max, dtype=dtype)  # "edge": every value the type's maximum (UInt32: above 2^31)


@pytest.mark.parametrize("type_id,dtype", NARROW)
@pytest.mark.parametrize("device", [True, False])
def test_narrow_integers_are_the_int64_columns_they_stand_for(type_id, dtype, device):
    rng = np.random.default_rng(type_id * 2 + int(device))
    specs = [spec(T.COUNT, 0), spec(T.NUMERIC_STATS, 0, flags=T.FLAG_VARIANCE), spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY),
             spec(T.KLL, 0, kll_k=200), spec(T.COMOMENTS, 0, column2=1), spec(T.DISTINCT, 0, columns=[0, 1])]
    for n, kind, null_frac in ((0, "full", 0.0), (1, "edge", 0.0), (777, "few", 0.3), (200_003, "full", 0.07), (1_500_000, "full", 0.0)):
        vals = draw(rng, dtype, n, kind)
        other = draw(rng, dtype, n, "few")
        validity = orc.pack_validity(rng.random(n) >= null_frac) if null_frac else None
        wide, wide2 = vals.astype(np.int64), other.astype(np.int64)
        # three batches with Arrow offsets: the window arithmetic of 1- and 2-byte elements
        cuts = [0, n // 3, n // 3 + min(n, 129), n] if n > 400 else [0, n]
        batches = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            batches.append([narrow_column(type_id, vals, validity, device, offset=lo, length=hi - lo),
                            narrow_column(type_id, other, None, device, offset=lo, length=hi - lo)])
        res, _, st = run_plan(specs, batches, hint=n)
        if not device and n > 20_000:
            # DataFusion-sized HOST batches are only noted: their windows are widened on the way into the pinned arena
            # (coalesce.cpp, widen_copy) and the flush sees an Int64 column -- the same answers
            stream = [[narrow_column(type_id, vals, validity, False, offset=lo, length=min(8192, n - lo)),
                       narrow_column(type_id, other, None, False, offset=lo, length=min(8192, n - lo))]
                      for lo in range(0, n, 8192)]
            res2, _, st2 = run_plan(specs, stream, hint=n)
            assert st2.profile_get("coalesce")["bytes"] == len(stream)   # every batch was only noted
            key = lambda r: (r.total, r.non_null, r.min_i, r.max_i, r.sum_i, r.distinct, r.groups_once, r.kll_n)
            assert [key(r) for r in res2] == [key(r) for r in res]
        oc, os_ = orc.count(validity, n), orc.stats(wide, validity)
        assert (res[0].total, res[0].non_null) == (oc.total, oc.non_null)
        assert (res[1].total, res[1].non_null, res[1].is_float) == (n, os_.non_null, 0)
        if os_.non_null:
            assert (res[1].min_i, res[1].max_i, res[1].sum_i) == (os_.min_i, os_.max_i, os_.sum_i_wrapping)
            assert rel_err(res[1].mean, os_.mean) < 1e-12
        if os_.has_variance:
            assert rel_err(res[1].var_samp, os_.var_samp) < 1e-9
        od = orc.distinct_bits64(wide, validity)
        assert (res[2].distinct, res[2].groups_once, res[2].non_null) == (od.distinct, od.groups_once, od.non_null)
        assert res[3].kll_n == os_.non_null
        ocm = orc.comoments(wide, wide2, validity, None)
        assert res[4].non_null == ocm.n
        if ocm.n:
            assert rel_err(res[4].sum_xy, ocm.sum_xy) < 1e-9
        # the tuple (a, b): a NULL component is a value of its own
        pairs = set()
        mask = orc.unpack_validity(validity, n) if validity is not None else np.ones(n, bool)
        for a, ok, b in zip(wide.tolist(), mask.tolist(), wide2.tolist()):
            pairs.add((a if ok else None, b))
        assert res[5].distinct == len(pairs) or n == 0


@pytest.mark.parametrize("device", [True, False])
def test_uint64_and_boolean_columns_count_and_distinct(device):
    import torch

    rng = np.random.default_rng(99 + int(device))
    n = 300_007
    u = rng.integers(0, 1 << 63, size=n, dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n).astype(np.uint64)
    u[rng.random(n) < 0.5] = np.uint64((1 << 64) - 1) - np.uint64(rng.integers(0, 5))  # keys above 2^63, heavy repeats
    uval = orc.pack_validity(rng.random(n) >= 0.1)
    bools = rng.random(n) < 0.3
    bbits = np.concatenate([orc.pack_validity(bools), np.zeros(64, np.uint8)])
    bval = orc.pack_validity(rng.random(n) >= 0.2)

    def cols(lo, hi):
        uu = np.concatenate([u, np.zeros(8, np.uint64)])
        if device:
            uc = T.Column(T.UINT64, hi - lo, values=torch.from_numpy(uu.view(np.int64)).cuda(), validity=to_device(pad_validity(uval)), offset=lo)
            bc = T.Column(T.BOOL, hi - lo, values=to_device(bbits), validity=to_device(pad_validity(bval)), offset=lo)
        else:
            uc = T.Column(T.UINT64, hi - lo, values=uu, validity=pad_validity(uval), offset=lo)
            bc = T.Column(T.BOOL, hi - lo, values=bbits, validity=pad_validity(bval), offset=lo)
        return [uc, bc]

    specs = [spec(T.COUNT, 0), spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.COUNT, 1),
             spec(T.DISTINCT, 1, flags=T.FLAG_MULTIPLICITY), spec(T.DISTINCT, 0, columns=[0, 1])]
    cuts = [0, 100_001, 100_130, n]
    res, plan, st = run_plan(specs, [cols(lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:])], hint=n)
    ou = orc.distinct_bits64(u.view(np.int64), uval)
    assert (res[0].total, res[0].non_null) == (n, ou.non_null)
    assert (res[1].distinct, res[1].groups_once) == (ou.distinct, ou.groups_once)
    ob = orc.distinct_bits64(bools.astype(np.int64), bval)
    assert (res[2].total, res[2].non_null) == (n, ob.non_null)
    assert (res[3].distinct, res[3].groups_once) == (ob.distinct, ob.groups_once) and res[3].distinct == 2
    um, bm = orc.unpack_validity(uval, n), orc.unpack_validity(bval, n)
    pairs = {(a if x else None, b if y else None) for a, x, b, y in zip(u.tolist(), um.tolist(), bools.tolist(), bm.tolist())}
    assert res[4].distinct == len(pairs)
    if not device:  # the same columns as a stream of 8192-row HOST batches (Boolean bits at any offset, UInt64 as it is)
        stream = [cols(lo, min(n, lo + 8192)) for lo in range(0, n, 8192)]
        res_s, _, st_s = run_plan(specs, stream, hint=n)
        assert st_s.profile_get("coalesce")["bytes"] == len(stream)
        assert [(r.total, r.non_null, r.distinct, r.groups_once) for r in res_s] == [(r.total, r.non_null, r.distinct, r.groups_once) for r in res]
    # a blob round trip and a merge keep the key sets
    other = T.State.deserialize(plan, st.serialize())
    other.merge([st])
    again = other.finalize()
    assert (again[1].distinct, again[3].distinct) == (ou.distinct, 2)
    # statistics of such columns are refused, not invented (the reference cannot read them either: statistics.rs:277-308)
    for bad in (spec(T.NUMERIC_STATS, 0), spec(T.KLL, 1, kll_k=200), spec(T.COMOMENTS, 0, column2=1)):
        with pytest.raises(T.TgxError) as e:
            run_plan([bad], [cols(0, 1000)])
        assert e.value.status == "TGX_UNSUPPORTED" and "COUNT and DISTINCT checks only" in str(e.value)
    # completeness alone reads no values at all
    res, _, _ = run_plan([spec(T.COUNT, 0)], [[T.Column(T.BOOL, n, values=None, validity=(to_device(pad_validity(bval)) if device else pad_validity(bval)))]])
    assert (res[0].total, res[0].non_null) == (n, ob.non_null)


def test_arrow_arrays_of_the_new_types():
    import pyarrow as pa

    arrays = {"i8": pa.array([1, -2, None, 127, -128], pa.int8()), "u16": pa.array([0, 65535, 7, None, 7], pa.uint16()),
              "u32": pa.array([4_000_000_000, 1, 1, None, 0], pa.uint32()), "u64": pa.array([2**64 - 1, 2**63, 5, 5, None], pa.uint64()),
              "b": pa.array([True, False, None, True, True], pa.bool_())}
    table = pa.table(arrays).slice(1, 4)  # (offsets into every buffer, the Boolean bits included)
    cols = [T.Column.from_arrow(table.column(i).chunk(0)) for i in range(table.num_columns)]
    specs = [spec(T.COUNT, i) for i in range(5)] + [spec(T.DISTINCT, i) for i in range(5)] + [spec(T.NUMERIC_STATS, 2)]
    res, _, _ = run_plan(specs, [cols])
    assert [r.non_null for r in res[:5]] == [3, 3, 3, 3, 3]
    assert [r.distinct for r in res[5:10]] == [3, 2, 2, 2, 2]
    assert (res[10].min_i, res[10].max_i, res[10].sum_i) == (0, 1, 2)


def test_binary_decimal_and_fixed_width_columns_as_byte_strings():
    """Binary / LargeBinary / BinaryView have the string layouts; FixedSizeBinary(w), Decimal128 and Decimal256 are w-byte
    values under synthetic offsets (no copy of the values): COUNT and COUNT(DISTINCT) on them against Python sets, on
    SLICED arrays with NULLs (the offsets, the validity bits and the value slots all start mid-buffer)."""
    import decimal

    import pyarrow as pa

    rng = np.random.default_rng(77)
    n, cut = 50_000, 1_237
    pool = [bytes(rng.integers(0, 256, size=int(k), dtype=np.uint8)) for k in rng.integers(0, 40, size=300)]
    blobs = [None if rng.random() < 0.1 else pool[int(i)] for i in rng.integers(0, len(pool), size=n)]
    fixed = [None if b is None else (b + b"\0" * 12)[:12] for b in blobs]
    cents = [None if rng.random() < 0.2 else decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-400, 400, size=n)]
    big = [None if c is None else c * 10**30 for c in cents]
    arrays = [pa.array(blobs, pa.binary()), pa.array(blobs, pa.large_binary()), pa.array(blobs, pa.binary_view()),
              pa.array(fixed, pa.binary(12)), pa.array(cents, pa.decimal128(12, 2)), pa.array(big, pa.decimal256(50, 2))]
    pylists = [blobs, blobs, blobs, fixed, cents, big]
    cols = [T.Column.from_arrow(a.slice(cut, n - 2 * cut)) for a in arrays]
    specs = [spec(T.COUNT, i) for i in range(6)] + [spec(T.DISTINCT, i, flags=T.FLAG_MULTIPLICITY) for i in range(6)]
    res, _, _ = run_plan(specs, [cols])
    for i, vals in enumerate(pylists):
        part = vals[cut:n - cut]
        live = [v for v in part if v is not None]
        assert (res[i].total, res[i].non_null) == (len(part), len(live)), arrays[i].type
        counts = {}
        for v in live:
            counts[v] = counts.get(v, 0) + 1
        assert res[6 + i].distinct == len(counts), arrays[i].type
        assert res[6 + i].groups_once == sum(1 for c in counts.values() if c == 1), arrays[i].type
    # the same column in several batches cut at odd rows (each cut a slice with its own offset)
    bounds = [0, 1, 8191, 8192, 30_001, n]
    batches = [[T.Column.from_arrow(a.slice(lo, hi - lo)) for a in arrays] for lo, hi in zip(bounds, bounds[1:])]
    res2, _, _ = run_plan(specs, batches)
    for i, vals in enumerate(pylists):
        live = [v for v in vals if v is not None]
        assert (res2[i].total, res2[i].non_null, res2[6 + i].distinct) == (n, len(live), len(set(live))), arrays[i].type


def test_completeness_of_any_arrow_type():
    """COUNT reads the validity bitmap and the length: lists, structs, maps, intervals, NullArrays .. go through
    Column.validity_only (what ValidationSuite.run falls back to for a column only completeness / size look at)"""
    import pyarrow as pa

    from term_amd.suite import Assertion, Check, CompletenessOptions, Level, ValidationSuite

    n = 20_000
    rng = np.random.default_rng(3)
    keep = rng.random(n) > 0.25
    lists = pa.array([[1, 2] if k else None for k in keep], pa.list_(pa.int32()))
    structs = pa.array([{"a": 1, "b": "x"} if k else None for k in keep[::-1]], pa.struct([("a", pa.int8()), ("b", pa.string())]))
    nulls = pa.nulls(n)
    months = pa.array([pa.MonthDayNano([3, 0, 0]) if k else None for k in keep], pa.month_day_nano_interval())
    arrays = [lists, structs, nulls, months]
    cols = [T.Column.validity_only(a.slice(77, n - 100)) for a in arrays]
    res, _, _ = run_plan([spec(T.COUNT, i) for i in range(4)], [cols])
    want = [int(keep[77:n - 23].sum()), int(keep[::-1][77:n - 23].sum()), 0, int(keep[77:n - 23].sum())]
    assert [(r.total, r.non_null) for r in res] == [(n - 100, w) for w in want]
    # through the suite: a table with a list column beside the checked ones no longer needs the list column dropped
    ids = pa.array(np.arange(n, dtype=np.int64))
    table = pa.table({"id": ids, "tags": lists, "rec": structs})
    check = (Check.builder("c").level(Level.ERROR).validates_uniqueness(["id"], 1.0)
             .completeness("tags", CompletenessOptions.threshold(0.9)).completeness("rec", CompletenessOptions.threshold(0.7))
             .has_size(Assertion.Equals(float(n))).build())
    out = ValidationSuite.builder("s").check(check).build().run(table)
    assert (out.report.metrics.total_checks, out.report.metrics.failed_checks) == (4, 1)
    assert [i.constraint_name for i in out.report.issues] == ["completeness"] and "tags" in out.report.issues[0].message
