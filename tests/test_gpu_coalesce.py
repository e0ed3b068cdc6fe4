"""-m gpu: library-side coalescing of small batches (tgx_api.cpp "coalescing", kernels/gather.hip).  DataFusion hands a
drop-in 8192-row RecordBatches (TG/core/context.rs:28-38); tgx_update only notes them and a flush runs the pending
ones as ONE batch.  Whatever the batching, the results must be those of the table: integers / counts / key sets /
pattern hits bit-exact against the oracle and against the same table fed as one batch with coalescing off, float
aggregates to 1e-12 (the association of a compensated sum moves its last digits, nothing else)."""
import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import numeric_column, pad_validity, rel_err, to_device
from test_gpu_parity import check_stats

pytestmark = pytest.mark.gpu


def ragged_cuts(n, rng, sizes=(8192, 8192, 1000, 7, 65536, 8192, 64, 1, 4099)):
    cuts, at, k = [0], 0, 0
    while at < n:
        at = min(n, at + sizes[k % len(sizes)])
        cuts.append(at)
        k += 1
    return cuts


def feed(plan, cols_np, cuts, device, env=None, monkeypatch=None):
    if monkeypatch is not None:
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
    st = T.State(plan)
    if monkeypatch is not None:
        for k in (env or {}):
            monkeypatch.delenv(k)
    for a, b in zip(cuts[:-1], cuts[1:]):
        st.update([numeric_column(v, m, device, offset=a, length=b - a) for v, m in cols_np])
    return st


def exact_key(r):
    return (r.kind, r.total, r.non_null, r.has_value, r.min_i, r.max_i, r.sum_i, r.distinct, r.groups_once, r.matches,
            r.kll_n, r.min_f, r.max_f)


def compare(got, want):
    assert [exact_key(r) for r in got] == [exact_key(r) for r in want]
    for g, w in zip(got, want):
        for f in ("sum_f", "mean", "var_samp", "sum_x", "sum_y", "sum_x2", "sum_y2", "sum_xy", "co_c_xy", "co_m2_x"):
            a, b = getattr(g, f), getattr(w, f)
            assert (np.isnan(a) and np.isnan(b)) or rel_err(a, b) <= 1e-12, (f, a, b)


@pytest.mark.parametrize("device", [True, False])
@pytest.mark.parametrize("flush_rows", [None, "20000"])  # one flush at finalize / a flush every few batches
def test_numeric_suite_equals_one_batch(device, flush_rows, monkeypatch):
    rng = np.random.default_rng(7)
    n = 300_000 + 37
    ids = rng.permutation(n).astype(np.int64)
    keys = rng.integers(0, n // 10, size=n, dtype=np.int64)
    sparse = rng.integers(-2**62, 2**62, size=n, dtype=np.int64)
    sparse[: n // 3] = sparse[n // 3: 2 * (n // 3)]
    f1 = rng.standard_normal(n) * 1e3
    f2 = 0.25 * f1 + rng.standard_normal(n)
    f2[rng.random(n) < 0.001] = -0.0
    masks = [None, rng.random(n) >= 0.05, rng.random(n) >= 0.5, rng.random(n) >= 0.05, None]
    cols_np = [(v, None if m is None else orc.pack_validity(m)) for v, m in zip([ids, keys, sparse, f1, f2], masks)]
    specs = []
    for ci in range(5):
        specs += [spec(T.COUNT, ci), spec(T.NUMERIC_STATS, ci, flags=T.FLAG_VARIANCE if ci == 3 else 0)]
    specs += [spec(T.DISTINCT, 0), spec(T.DISTINCT, 1, flags=T.FLAG_MULTIPLICITY), spec(T.DISTINCT, 2, flags=T.FLAG_MULTIPLICITY),
              spec(T.DISTINCT, 4), spec(T.COMOMENTS, 3, column2=4), spec(T.KLL, 3, kll_k=200)]
    T.init()
    plan = T.Plan(specs)
    whole = feed(plan, cols_np, [0, n], True, {"TGX_COALESCE": "0"}, monkeypatch)
    want = whole.finalize()
    assert whole.profile_get("coalesce")["launches"] == 0
    env = {} if flush_rows is None else {"TGX_COALESCE_FLUSH_ROWS": flush_rows}
    st = feed(plan, cols_np, ragged_cuts(n, rng), device, env, monkeypatch)
    got = st.finalize()
    stats = st.profile_get("coalesce")
    assert stats["launches"] >= (1 if flush_rows is None else 5) and stats["bytes"] >= 20  # flushes, batches coalesced
    compare(got, want)
    # and against the oracle
    for ci, (v, b) in enumerate(cols_np):
        c = orc.count(b, n)
        assert (got[2 * ci].total, got[2 * ci].non_null) == (c.total, c.non_null)
        check_stats(got[2 * ci + 1], orc.stats(v, b), variance=ci == 3)
    for k, ci in ((10, 0), (11, 1), (12, 2), (13, 4)):
        v, b = cols_np[ci]
        d = orc.distinct_bits64(v.view(np.uint64), b, n=n)
        assert (got[k].distinct, got[k].non_null) == (d.distinct, d.non_null)
        if specs[k].flags & T.FLAG_MULTIPLICITY:
            assert got[k].groups_once == d.groups_once
    assert got[15].kll_n == int(masks[3].sum())
    # quantiles of a sketch built from flushes stay inside the stated rank error
    vals = np.sort(f1[masks[3]])
    for phi in (0.05, 0.5, 0.95):
        q = st.kll_quantile(15, phi)
        assert abs(np.searchsorted(vals, q) / len(vals) - phi) < 1.65 / np.sqrt(200)


@pytest.mark.parametrize("device", [True, False])
def test_growing_ids_are_repaired_across_flushes(device, monkeypatch):
    """a sampled-range key set keeps views of its batches -- here views into the coalescing regions, which take turns:
    ids that grow past the first flush's range arrive many flushes later and must still be counted exactly"""
    n = 400_000
    ids = np.arange(n, dtype=np.int64) * 3
    ids[n // 2:] += 10_000_000            # far outside the range the first flush's sample laid out
    rng = np.random.default_rng(1)
    ids[rng.random(n) < 0.01] = 5          # repeats everywhere
    cols_np = [(ids, None)]
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.NUMERIC_STATS, 0)])
    st = feed(plan, cols_np, list(range(0, n, 8192)) + [n], device, {"TGX_COALESCE_FLUSH_ROWS": "70000"}, monkeypatch)
    res = st.finalize()
    assert st.profile_get("coalesce")["launches"] >= 5
    d = orc.distinct_bits64(ids.view(np.uint64), None, n=n)
    assert (res[0].total, res[0].distinct, res[0].groups_once) == (d.total, d.distinct, d.groups_once)
    assert (res[1].min_i, res[1].max_i, res[1].sum_i) == (int(ids.min()), int(ids.max()), int(ids.sum()))


@pytest.mark.parametrize("large", [False, True])
def test_host_string_batches(large, monkeypatch):
    """Utf8 / LargeUtf8 HOST batches (what DataFusion streams): offsets re-based, bytes concatenated; the pattern,
    length and DISTINCT checks of the coalesced column equal those of the table"""
    import pyarrow as pa

    rng = np.random.default_rng(3)
    n = 120_000
    vals = []
    for i in range(n):
        r = rng.random()
        if r < 0.02:
            vals.append(None)
        elif r < 0.8:
            vals.append("user%d@example%d.com" % (i % 50_000, i % 1000))
        elif r < 0.9:
            vals.append("")
        else:
            vals.append("not an e-mail é中 %d" % (i % 97))
    arr = pa.array(vals, type=pa.large_string() if large else pa.string())
    specs = [spec(T.REGEX_MATCH, 0, pattern=r"^[^@]+@[^@]+\.[^@]+$", flags=T.FLAG_NULL_IS_VALID),
             spec(T.REGEX_MATCH, 0, pattern="@"), spec(T.LENGTH, 0, length_min=1, length_max=24),
             spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.COUNT, 0)]
    T.init()
    plan = T.Plan(specs)

    def column(a, lo, hi):
        sl = a.slice(lo, hi - lo)  # a sliced array: Arrow offset != 0, shared buffers
        bufs = sl.buffers()
        validity = None if bufs[0] is None else np.frombuffer(bufs[0], dtype=np.uint8)
        offsets = np.frombuffer(bufs[1], dtype=np.int64 if large else np.int32)
        data = np.frombuffer(bufs[2], dtype=np.uint8) if bufs[2] is not None else np.zeros(1, np.uint8)
        ctor = T.Column.large_utf8 if large else T.Column.utf8
        return ctor(offsets, data, pad_validity(validity) if validity is not None else None, length=len(sl), offset=sl.offset)

    monkeypatch.setenv("TGX_COALESCE", "0")
    whole = T.State(plan)
    monkeypatch.delenv("TGX_COALESCE")
    whole.update([column(arr, 0, n)])
    want = whole.finalize()
    monkeypatch.setenv("TGX_COALESCE_FLUSH_ROWS", "30000")
    st = T.State(plan)
    monkeypatch.delenv("TGX_COALESCE_FLUSH_ROWS")
    cuts = ragged_cuts(n, rng, sizes=(8192, 1000, 3, 8192, 1, 5000))
    for a, b in zip(cuts[:-1], cuts[1:]):
        st.update([column(arr, a, b)])
    got = st.finalize()
    assert st.profile_get("coalesce")["launches"] >= 3
    compare(got, want)
    import re

    pat = re.compile(r"^[^@]+@[^@]+\.[^@]+\Z")
    assert got[0].matches == sum(1 for v in vals if v is None or pat.search(v))
    assert got[1].matches == sum(1 for v in vals if v is not None and "@" in v)
    assert got[2].matches == sum(1 for v in vals if v is None or 1 <= len(v) <= 24)
    assert got[3].distinct == len({v for v in vals if v is not None})
    assert (got[4].total, got[4].non_null) == (n, sum(v is not None for v in vals))


def test_int32_float32_and_count_only_columns(monkeypatch):
    from test_gpu_numeric32 import col32

    rng = np.random.default_rng(9)
    n = 150_000
    i32 = rng.integers(0, 40_000, size=n, dtype=np.int64).astype(np.int32)
    f32 = (rng.standard_normal(n) * 100).astype(np.float32)
    mask = rng.random(n) >= 0.1
    vb = orc.pack_validity(mask)
    specs = [spec(T.COUNT, 0), spec(T.NUMERIC_STATS, 0), spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY),
             spec(T.NUMERIC_STATS, 1, flags=T.FLAG_VARIANCE), spec(T.COUNT, 2)]
    T.init()
    plan = T.Plan(specs)
    for device in (True, False):
        monkeypatch.setenv("TGX_COALESCE_FLUSH_ROWS", "50000")
        st = T.State(plan)
        monkeypatch.delenv("TGX_COALESCE_FLUSH_ROWS")
        cuts = ragged_cuts(n, rng)
        vpad = pad_validity(vb)
        vdev = to_device(vpad) if device else vpad
        for a, b in zip(cuts[:-1], cuts[1:]):
            # column 2 is validity-only (COUNT): no values buffer at all
            count_only = T.Column.int64(None, vdev, length=b - a, offset=a)
            st.update([col32(i32, vb, device, offset=a, length=b - a), col32(f32, None, device, offset=a, length=b - a),
                       count_only])
        res = st.finalize()
        assert st.profile_get("coalesce")["launches"] >= 2
        wide = i32.astype(np.int64)
        c = orc.count(vb, n)
        assert (res[0].total, res[0].non_null) == (c.total, c.non_null) == (res[4].total, res[4].non_null)
        check_stats(res[1], orc.stats(wide, vb))
        d = orc.distinct_bits64(wide.view(np.uint64), vb, n=n)
        assert (res[2].distinct, res[2].groups_once) == (d.distinct, d.groups_once)
        check_stats(res[3], orc.stats(f32.astype(np.float64), None), variance=True)


def test_pending_batches_are_seen_by_every_reader():
    """merge, serialize, sync and reset with batches still pending"""
    rng = np.random.default_rng(4)
    n = 50_000
    v = rng.integers(0, 1000, size=n, dtype=np.int64)
    specs = [spec(T.NUMERIC_STATS, 0), spec(T.DISTINCT, 0)]
    T.init()
    plan = T.Plan(specs)
    a, b = T.State(plan), T.State(plan)
    a.update([numeric_column(v, None, True, length=8192)])
    b.update([numeric_column(v, None, False, offset=8192, length=n - 8192)])
    blob = b.serialize()                       # flushes b
    a.merge([T.State.deserialize(plan, blob)])  # flushes a first
    r = a.finalize()
    assert (r[0].total, r[0].sum_i, r[1].distinct) == (n, int(v.sum()), len(np.unique(v)))
    c = T.State(plan)
    c.update([numeric_column(v, None, True, length=100)])
    c.reset()                                  # pending batches are dropped with the rest
    c.update([numeric_column(v, None, True, offset=100, length=50)])
    c.sync()
    r = c.finalize()
    assert (r[0].total, r[0].sum_i) == (50, int(v[100:150].sum()))
    e = T.State(plan)
    assert e.finalize()[0].total == 0


@pytest.mark.parametrize("shape", ["up", "down", "both", "jump_too_far"])
def test_host_streams_of_growing_ids_grow_the_bitmap(shape, monkeypatch):
    """HOST batches: the host sees every key on its way into the arena, so a flush knows its value range -- the first
    flush lays the bitmap out without sampling, later flushes whose ids lie outside it GROW it (whole slices, old
    words moved) instead of leaving their keys to the repair; a jump that would make the bitmap too sparse stays with
    the repair.  Exact either way."""
    n = 600_000
    base = np.arange(n, dtype=np.int64)
    if shape == "up":
        ids = base * 2 + 7_000_000
    elif shape == "down":
        ids = 50_000_000 - base * 3
    elif shape == "both":
        ids = np.where(base % 2 == 0, 10_000_000 + base, 10_000_000 - base)
    else:
        ids = base.copy()
        ids[n // 2:] += 1 << 40  # the second half far away: too sparse for one bitmap
    rng = np.random.default_rng(8)
    ids[rng.random(n) < 0.02] = ids[3]
    mask = rng.random(n) >= 0.03
    vb = orc.pack_validity(mask)
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.NUMERIC_STATS, 0), spec(T.COUNT, 0)])
    st = feed(plan, [(ids, vb)], list(range(0, n, 8192)) + [n], False, {"TGX_COALESCE_FLUSH_ROWS": "70000"}, monkeypatch)
    res = st.finalize()
    assert st.profile_get("coalesce")["launches"] >= 5
    d = orc.distinct_bits64(ids.view(np.uint64), vb, n=n)
    assert (res[0].total, res[0].non_null, res[0].distinct, res[0].groups_once) == (d.total, d.non_null, d.distinct, d.groups_once)
    o = orc.stats(ids, vb)
    assert (res[1].min_i, res[1].max_i, res[1].sum_i) == (o.min_i, o.max_i, o.sum_i_wrapping)


def test_states_fed_from_several_threads_at_once():
    """A state per partition stream, every stream on its own thread (HOST 8192-row batches: the pinned-arena copies
    share ONE helper thread process-wide).  Two callers used to overwrite each other's job list there -- copies
    skipped or a caller waiting for ever; the differential tester's threaded ranks found it."""
    import threading

    n, n_threads, n_cols = 8192 * 40, 4, 8
    plan_specs = []
    for c in range(n_cols):
        plan_specs += [spec(T.COUNT, c), spec(T.NUMERIC_STATS, c)]
    plan_specs.append(spec(T.DISTINCT, 0))
    T.init()
    plan = T.Plan(plan_specs)
    tables, results, errors = [], [None] * n_threads, []
    for t in range(n_threads):
        rng = np.random.default_rng(100 + t)
        cols = [(rng.permutation(n).astype(np.int64) + t * n, None)]
        cols += [(rng.integers(-2**40, 2**40, size=n, dtype=np.int64), orc.pack_validity(rng.random(n) >= 0.1))
                 for _ in range(n_cols - 1)]
        tables.append(cols)
    start = threading.Barrier(n_threads)

    def worker(t):
        try:
            import torch

            torch.cuda.set_device(0)
            st = T.State(plan)
            start.wait()
            for rep in range(3):
                st.reset()
                for a in range(0, n, 8192):
                    st.update([numeric_column(v, m, False, offset=a, length=8192) for v, m in tables[t]])
                results[t] = st.finalize()
        except Exception:  # noqa: BLE001
            import traceback

            errors.append(traceback.format_exc())

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not errors, errors
    assert not any(th.is_alive() for th in threads), "a feeding thread is stuck"
    for t in range(n_threads):
        res = results[t]
        for c, (v, m) in enumerate(tables[t]):
            cnt = orc.count(m, n)
            assert (res[2 * c].total, res[2 * c].non_null) == (cnt.total, cnt.non_null)
            check_stats(res[2 * c + 1], orc.stats(v, m))
        assert res[2 * n_cols].distinct == n


@pytest.mark.parametrize("order", ["ascending", "descending"])
@pytest.mark.parametrize("pattern", ["HDHDH", "DHDHD", "DDHHH", "HHDDD"])
def test_growing_ids_from_host_and_device_batches(order, pattern, monkeypatch):
    """A stream of ids that keep growing (or falling), some batches HOST, some DEVICE.  A HOST flush tells the library
    its value range and the bitmap grows to cover it; a DEVICE flush does not, and its keys wait outside the range for
    the repair -- the bitmap must not grow over them meanwhile (the repair would take them for keys it already has:
    the differential tester lost a whole batch of keys that way)."""
    n = 293_798
    vals = (1000 + np.arange(n, dtype=np.int64)) if order == "ascending" else (4_705_409_122 - np.arange(n, dtype=np.int64))
    dv = to_device(vals)
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.NUMERIC_STATS, 0)])
    for flush_rows in ("20000", None):
        if flush_rows:
            monkeypatch.setenv("TGX_COALESCE_FLUSH_ROWS", flush_rows)
        st = T.State(plan)
        monkeypatch.delenv("TGX_COALESCE_FLUSH_ROWS", raising=False)
        for b, lo in enumerate(range(0, n, 65536)):
            ln = min(65536, n - lo)
            st.update([T.Column.int64(vals if pattern[b] == "H" else dv, None, length=ln, offset=lo)])
        r = st.finalize()
        assert (r[0].total, r[0].distinct, r[0].groups_once) == (n, n, n), (flush_rows, r[0].distinct)
        check_stats(r[1], orc.stats(vals))


def test_a_refused_batch_is_noted_for_no_column(monkeypatch):
    """a batch whose SECOND column is refused (offsets that decrease) had already left a segment in the
    first column's pending list -- the next flush sized the coalesced buffers for `rows` and gathered rows + nrows.  A
    batch is noted for all columns or for none: the state goes on as if the call had not been made."""
    rng = np.random.default_rng(3)
    n = 40_000
    ids = rng.permutation(n).astype(np.int64)
    words = ["w%d" % (i % 1000) for i in range(n)]
    offsets = np.zeros(n + 1, dtype=np.int32)
    offsets[1:] = np.cumsum([len(w) for w in words])
    data = np.frombuffer("".join(words).encode(), dtype=np.uint8)
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0), spec(T.NUMERIC_STATS, 0), spec(T.DISTINCT, 1), spec(T.COUNT, 1)])
    st = T.State(plan)

    def batch(a, b, offs=offsets):
        return [numeric_column(ids, None, False, offset=a, length=b - a), T.Column.utf8(offs, data, None, length=b - a, offset=a)]

    st.update(batch(0, 8192))
    bad = offsets.copy()
    bad[8192 + 100:] -= 50_000  # the window's last offset lies before its first
    with pytest.raises(T.TgxError) as e:
        st.update(batch(8192, 16384, bad))
    assert e.value.status == "TGX_INVALID_ARGUMENT"
    for a in range(8192, n, 8192):
        st.update(batch(a, min(n, a + 8192)))
    res = st.finalize()
    assert (res[0].distinct, res[1].total, res[1].min_i, res[1].max_i) == (n, n, 0, n - 1)
    assert (res[2].distinct, res[3].total, res[3].non_null) == (1000, n, n)


def test_empty_batches_do_not_flush():
    """an empty RecordBatch between the others (streams interleave them) is nothing to flush for"""
    rng = np.random.default_rng(4)
    n = 64_000
    v = rng.integers(0, 1000, size=n, dtype=np.int64)
    T.init()
    plan = T.Plan([spec(T.NUMERIC_STATS, 0), spec(T.DISTINCT, 0)])
    st = T.State(plan)
    for a in range(0, n, 8000):
        st.update([numeric_column(v, None, False, offset=a, length=8000)])
        st.update([numeric_column(v, None, False, offset=a, length=0)])
    assert st.profile_get("coalesce")["launches"] == 0  # eight noted batches, no flush yet
    res = st.finalize()
    assert (res[0].total, res[0].sum_i, res[1].distinct) == (n, int(v.sum()), len(np.unique(v)))


# ---- Utf8View and Dictionary<Int32, Utf8> batches (what DataFusion reads Parquet strings as), HOST buffers -----------
def _string_specs():
    return [spec(T.REGEX_MATCH, 0, pattern=r"^[^@]+@[^@]+\.[^@]+$", flags=T.FLAG_NULL_IS_VALID),
            spec(T.REGEX_MATCH, 0, pattern="@"), spec(T.LENGTH, 0, length_min=1, length_max=24),
            spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.COUNT, 0), spec(T.NUMERIC_STATS, 1)]


def _string_values(rng, n):
    vals = []
    for i in range(n):
        r = rng.random()
        if r < 0.05:
            vals.append(None)
        elif r < 0.55:
            vals.append("user%d@example%d.com" % (i % 5000, i % 7))  # long values: 13+ bytes, in the data buffers
        elif r < 0.75:
            vals.append("s%d" % (i % 300))                            # short values: inline in the view
        elif r < 0.8:
            vals.append("")
        else:
            vals.append("not an e-mail é中 %d" % (i % 97))
    return vals


@pytest.mark.parametrize("flush_rows", [None, "30000"])
def test_host_utf8view_batches(flush_rows, monkeypatch):
    """views copied as they are, the stretches of the data buffers a window's long views point into laid one behind the
    other, views re-pointed on the device (kernels/gather.hip, kind 3): results of the same values as one batch"""
    from test_gpu_utf8view import encode_views

    rng = np.random.default_rng(21)
    n = 120_000 + 11
    vals = _string_values(rng, n)
    nums = rng.integers(-1000, 1000, size=n, dtype=np.int64)
    views, bufs, validity = encode_views(vals, rng, n_buffers=3)
    v = pad_validity(validity) if validity is not None else None
    T.init()
    plan = T.Plan(_string_specs())

    def batch(lo, hi):
        return [T.Column.utf8_view(views, bufs, validity=v, length=hi - lo, offset=lo), numeric_column(nums, None, False, offset=lo, length=hi - lo)]

    monkeypatch.setenv("TGX_COALESCE", "0")
    whole = T.State(plan)
    monkeypatch.delenv("TGX_COALESCE")
    whole.update(batch(0, n))
    want = whole.finalize()
    if flush_rows:
        monkeypatch.setenv("TGX_COALESCE_FLUSH_ROWS", flush_rows)
    st = T.State(plan)
    if flush_rows:
        monkeypatch.delenv("TGX_COALESCE_FLUSH_ROWS")
    cuts = ragged_cuts(n, rng, sizes=(8192, 1000, 3, 8192, 1, 5000, 8192))
    for a, b in zip(cuts[:-1], cuts[1:]):
        st.update(batch(a, b))
    assert st.profile_get("coalesce")["bytes"] == len(cuts) - 1  # ("bytes": batches that were only noted -- all of them)
    got = st.finalize()
    assert st.profile_get("coalesce")["launches"] >= (3 if flush_rows else 1)
    compare(got, want)
    assert got[3].distinct == len({x for x in vals if x is not None})
    assert (got[4].total, got[4].non_null) == (n, sum(x is not None for x in vals))


def test_host_dictionary_batches(monkeypatch):
    """index windows shifted to where their dictionary starts in the coalesced one; a dictionary shared by the batches
    of a file is taken once per flush, a different one (the next file) appended behind it -- unused and repeated entries
    allowed, NULL dictionary values are NULL rows"""
    import pyarrow as pa

    rng = np.random.default_rng(22)
    n = 90_000
    words_a = ["user%d@example%d.com" % (i, i % 7) for i in range(4000)] + [None, "", "plain"]
    words_b = ["other%d@x.org" % i for i in range(1500)] + ["plain", "é中"]
    nums = rng.integers(0, 50, size=n, dtype=np.int64)
    T.init()
    plan = T.Plan(_string_specs())

    def dict_column(words, idx, mask):
        d = pa.array(words, type=pa.string())
        arr = pa.DictionaryArray.from_arrays(pa.array(idx, type=pa.int32(), mask=~mask), d)
        return arr

    half = n // 2
    idx_a = rng.integers(0, len(words_a), size=half)
    idx_b = rng.integers(0, len(words_b), size=n - half)
    mask = rng.random(n) >= 0.04
    arr_a, arr_b = dict_column(words_a, idx_a, mask[:half]), dict_column(words_b, idx_b, mask[half:])

    def batches(step):
        out = []
        for arr, base in ((arr_a, 0), (arr_b, half)):
            for lo in range(0, len(arr), step):
                hi = min(len(arr), lo + step)
                out.append([T.Column.from_arrow(arr.slice(lo, hi - lo)), numeric_column(nums, None, False, offset=base + lo, length=hi - lo)])
        return out

    monkeypatch.setenv("TGX_COALESCE", "0")
    whole = T.State(plan)
    monkeypatch.delenv("TGX_COALESCE")
    for cols in batches(1 << 20):
        whole.update(cols)
    want = whole.finalize()
    monkeypatch.setenv("TGX_COALESCE_FLUSH_ROWS", "40000")
    st = T.State(plan)
    monkeypatch.delenv("TGX_COALESCE_FLUSH_ROWS")
    for cols in batches(8192):
        st.update(cols)
    got = st.finalize()
    assert st.profile_get("coalesce")["launches"] >= 2
    compare(got, want)
    truth = [words_a[i] if m else None for i, m in zip(idx_a, mask[:half])] + [words_b[i] if m else None for i, m in zip(idx_b, mask[half:])]
    assert got[3].distinct == len({x for x in truth if x is not None})
    assert (got[4].total, got[4].non_null) == (n, sum(x is not None for x in truth))


def _retained(col):
    """the same view, promised to stay as it is until the next flushing call (TGX_MEM_HOST_RETAINED)"""
    col.c.mem = T.MEM_HOST_RETAINED
    if col.c.dictionary:
        col.c.dictionary.contents.mem = T.MEM_HOST_RETAINED
    return col


@pytest.mark.parametrize("flush_rows", [None, "30000"])
def test_host_buffers_kept_until_the_flush(flush_rows, monkeypatch):
    """TGX_MEM_HOST_RETAINED (include/tgx.h): the windows of a noted batch are copied when the flush runs -- all of them
    together, on the copy threads, the key column's MIN / MAX taken by whoever copies the piece -- instead of inside
    tgx_update.  Same results as the plain HOST stream and as one batch: numeric suite with a growing key column, a Utf8
    column, a Utf8View column and a dictionary column; a batch that mixes plain and kept HOST columns is a plain one."""
    import pyarrow as pa
    from test_gpu_utf8view import encode_views

    rng = np.random.default_rng(23)
    n = 150_000 + 5
    ids = np.arange(n, dtype=np.int64) * 3 + 1_000_000   # a growing key: the flush's range comes from the copiers
    ids[rng.random(n) < 0.01] = 1_000_000
    kval = orc.pack_validity(rng.random(n) >= 0.05)
    f = rng.standard_normal(n)
    vals = _string_values(rng, n)
    offs, data, sval = orc.utf8_from_list(vals)
    views, bufs, vval = encode_views(vals, rng, n_buffers=2)
    words = ["w%d@x.org" % i for i in range(3000)] + [None, ""]
    idx = rng.integers(0, len(words), size=n)
    darr = pa.DictionaryArray.from_arrays(pa.array(idx, type=pa.int32()), pa.array(words, type=pa.string()))
    T.init()
    from term_amd.csrc_patterns import EMAIL
    plan = T.Plan([spec(T.COUNT, 0), spec(T.NUMERIC_STATS, 0), spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY),
                   spec(T.NUMERIC_STATS, 1, flags=T.FLAG_VARIANCE), spec(T.REGEX_MATCH, 2, pattern=EMAIL), spec(T.DISTINCT, 2),
                   spec(T.REGEX_MATCH, 3, pattern=r"@"), spec(T.DISTINCT, 3), spec(T.DISTINCT, 4), spec(T.LENGTH, 4, length_min=2, length_max=9)])
    sv = np.concatenate([data, np.zeros(16, np.uint8)])

    def batch(lo, hi, kept, mixed=False):
        cols = [numeric_column(ids, kval, False, offset=lo, length=hi - lo), numeric_column(f, None, False, offset=lo, length=hi - lo),
                T.Column(T.UTF8, hi - lo, offsets=offs, data=sv, validity=pad_validity(sval), offset=lo),
                T.Column.utf8_view(views, bufs, validity=pad_validity(vval) if vval is not None else None, length=hi - lo, offset=lo),
                T.Column.from_arrow(darr.slice(lo, hi - lo))]
        if kept:
            cols = [_retained(c) if not (mixed and k == 1) else c for k, c in enumerate(cols)]
        return cols

    monkeypatch.setenv("TGX_COALESCE", "0")
    whole = T.State(plan)
    monkeypatch.delenv("TGX_COALESCE")
    whole.update(batch(0, n, False))
    want = whole.finalize()
    cuts = ragged_cuts(n, rng, sizes=(8192, 8192, 1000, 3, 8192, 1, 5000, 8192, 65536))
    results = {}
    for mode in ("plain", "kept", "mixed"):
        if flush_rows:
            monkeypatch.setenv("TGX_COALESCE_FLUSH_ROWS", flush_rows)
        st = T.State(plan)
        if flush_rows:
            monkeypatch.delenv("TGX_COALESCE_FLUSH_ROWS")
        for a, b in zip(cuts[:-1], cuts[1:]):
            st.update(batch(a, b, mode != "plain", mixed=(mode == "mixed")))
        assert st.profile_get("coalesce")["bytes"] == len(cuts) - 1   # every batch was only noted
        results[mode] = st.finalize()
        compare(results[mode], want)
    od = orc.distinct_bits64(ids, kval)
    assert (results["kept"][2].distinct, results["kept"][2].groups_once) == (od.distinct, od.groups_once)
    # a big batch of kept buffers is read before tgx_update returns, like a plain HOST one
    st = T.State(plan)
    st.update(batch(0, n, True))
    compare(st.finalize(), want)
    # and a reset forgets copies that were still waiting
    st = T.State(plan)
    st.update(batch(0, 8192, True))
    st.reset()
    st.update(batch(0, n, True))
    compare(st.finalize(), want)


def test_kept_host_streams_of_several_threads_share_the_copy_threads():
    """Four threads feed four states with kept 8192-row HOST batches at the same time: the copy threads are one pool, a
    state takes whichever of them are idle when it has 4 MB of windows noted (coalesce_start_copies) and waits for its
    own at the flush.  Every stream's answers equal the oracle's on its own data."""
    import threading

    T.init()
    plan = T.Plan([spec(T.COUNT, 0), spec(T.NUMERIC_STATS, 0), spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY),
                   spec(T.NUMERIC_STATS, 1, flags=T.FLAG_VARIANCE), spec(T.COUNT, 1)])
    n = 1_200_000 + 11
    data, want, errors = [], [], []
    for t in range(4):
        rng = np.random.default_rng(100 + t)
        ids = (np.arange(n, dtype=np.int64) * (t + 2)) + 7_000_000 * t   # a growing key, a range per stream
        ids[rng.random(n) < 0.02] = 7_000_000 * t
        kval = orc.pack_validity(rng.random(n) >= 0.03)
        f = rng.standard_normal(n) * (t + 1)
        fval = orc.pack_validity(rng.random(n) >= 0.10)
        data.append((ids, kval, f, fval))
        want.append((orc.stats(ids, kval, n), orc.distinct_bits64(ids.view(np.uint64), kval, n), orc.stats(f, fval, n)))
    got = [None] * 4

    def stream(t):
        try:
            ids, kval, f, fval = data[t]
            st = T.State(plan)
            for rep in range(2):  # (the second pass over a reset state: arenas and workers change hands again)
                st.reset()
                for lo in range(0, n, 8192):
                    hi = min(n, lo + 8192)
                    st.update([_retained(numeric_column(ids, kval, False, offset=lo, length=hi - lo)),
                               _retained(numeric_column(f, fval, False, offset=lo, length=hi - lo))])
                got[t] = st.finalize()
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=stream, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors and all(not th.is_alive() for th in threads), errors
    for t in range(4):
        si, di, sf = want[t]
        r = got[t]
        assert (r[0].total, r[0].non_null) == (n, si.non_null)
        assert (r[1].min_i, r[1].max_i, r[1].sum_i) == (si.min_i, si.max_i, si.sum_i_wrapping)
        assert (r[2].distinct, r[2].groups_once) == (di.distinct, di.groups_once)
        assert (r[3].non_null, r[3].min_f, r[3].max_f) == (sf.non_null, sf.min_f, sf.max_f)
        assert rel_err(r[3].sum_f, sf.sum_f) <= 1e-9 and rel_err(r[3].var_samp, sf.var_samp) <= 1e-9
        assert r[4].non_null == sf.non_null


@pytest.mark.parametrize("flush_rows", [None, "20000"])
def test_retained_window_with_empty_batches_interleaved(flush_rows, monkeypatch):
    """The shim's protocol (shim/src/planner.rs): every non-empty batch is handed over as
    TGX_MEM_HOST_RETAINED and kept in a window of `tgx_state_pending` batches; what falls out of the window is FREED
    (here: overwritten with garbage).  Empty RecordBatches between them are never noted by the library, are not counted
    by tgx_state_pending, and so must stay out of the window -- counting them would release the oldest batch the
    library still has to copy.  Results equal the one-batch run."""
    rng = np.random.default_rng(31)
    n = 120_000
    ids = rng.permutation(n).astype(np.int64)
    f = rng.standard_normal(n)
    mask = rng.random(n) >= 0.05
    T.init()
    plan = T.Plan([spec(T.COUNT, 0), spec(T.NUMERIC_STATS, 0), spec(T.DISTINCT, 0), spec(T.NUMERIC_STATS, 1, flags=T.FLAG_VARIANCE)])
    monkeypatch.setenv("TGX_COALESCE", "0")
    whole = T.State(plan)
    monkeypatch.delenv("TGX_COALESCE")
    whole.update([numeric_column(ids, orc.pack_validity(mask), False), numeric_column(f, None, False)])
    want = whole.finalize()
    if flush_rows:
        monkeypatch.setenv("TGX_COALESCE_FLUSH_ROWS", flush_rows)
    st = T.State(plan)
    if flush_rows:
        monkeypatch.delenv("TGX_COALESCE_FLUSH_ROWS")
    held = []  # (arrays of the batch, its columns): what the shim keeps alive
    cuts = ragged_cuts(n, rng, sizes=(8192, 8192, 1000, 3, 8192, 1, 5000))
    noted = 0
    for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        for _ in range(k % 3):  # zero, one or two empty batches before each real one
            before = st.pending()
            st.update([numeric_column(ids, None, False, offset=a, length=0), numeric_column(f, None, False, offset=a, length=0)])
            assert st.pending() == before  # an empty batch is not noted and not counted
        own_ids, own_f = ids[a:b].copy(), f[a:b].copy()
        own_val = pad_validity(orc.pack_validity(mask[a:b]))
        cols = [_retained(T.Column.int64(own_ids, own_val, length=b - a)), _retained(T.Column.float64(own_f, None, length=b - a))]
        st.update(cols)
        held.append((own_ids, own_f, own_val, cols))
        noted += 1
        pending, _rows = st.pending()
        assert pending <= len(held)
        while len(held) > pending:  # released = freed: the library must not look at these again
            o_ids, o_f, o_val, _ = held.pop(0)
            o_ids[:] = -7
            o_f[:] = np.nan
            o_val[:] = 0
    compare(st.finalize(), want)
    assert st.pending() == (0, 0)
