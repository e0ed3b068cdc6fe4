"""-m gpu: APPROX_DISTINCT as a HyperLogLog lane of the numeric scan (kernels/scan.hip, scan_hll_kernel;
TG/constraints/approx_count_distinct.rs:56-66).  The registers are a pure function of the SET of values, so the device's
must equal the oracle's byte for byte (oracle/tgx_oracle.c, orc_hll_registers) whatever the batching, the sharding or
the merge order, and the estimate with them; where the lane does not apply the exact key set answers."""
import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import numeric_column, pad_validity, to_device

pytestmark = pytest.mark.gpu


def registers_of(state):
    """the last task's registers: the tail of the state blob (term_amd/wire.py)"""
    blob = state.serialize()
    return np.frombuffer(blob[-16384:], dtype=np.uint8)


def columns(rng, n):
    ids = rng.permutation(n).astype(np.int64) * 3 - n
    dup = rng.integers(0, max(2, n // 10), size=n, dtype=np.int64)
    f = np.round(rng.standard_normal(n), 3)
    f[rng.random(n) < 0.01] = -0.0
    masks = [None, rng.random(n) >= 0.07, rng.random(n) >= 0.3]
    return [(v, None if m is None else orc.pack_validity(m)) for v, m in zip((ids, dup, f), masks)]


@pytest.mark.parametrize("n", [0, 1, 63, 5000, 700_001, 3_000_000])
@pytest.mark.parametrize("device", [True, False])
def test_registers_equal_the_oracle(n, device):
    rng = np.random.default_rng(n + 1)
    cols = columns(rng, n)
    # column 0: the lane alone (its MIN / MAX / SUM are skipped); column 1: next to NUMERIC_STATS; column 2: Float64
    specs = [spec(T.APPROX_DISTINCT, 0), spec(T.COUNT, 0), spec(T.APPROX_DISTINCT, 1), spec(T.NUMERIC_STATS, 1),
             spec(T.APPROX_DISTINCT, 2)]
    T.init()
    plan = T.Plan(specs)
    st = T.State(plan)
    cut = (n // 3) // 64 * 64 + (5 if n > 100 else 0)  # a second batch with a ragged Arrow offset
    for lo, hi in ((0, cut), (cut, n)):
        st.update([numeric_column(v, b, device, offset=lo, length=hi - lo) for v, b in cols])
    res = st.finalize()
    for k, ci in ((0, 0), (2, 1), (4, 2)):
        v, b = cols[ci]
        regs = orc.hll_registers(v.view(np.int64), b, n=n)
        assert res[k].distinct == orc.hll_estimate(regs), (ci, res[k].distinct)
        c = orc.count(b, n)
        assert (res[k].total, res[k].non_null) == (c.total, c.non_null)
        if n >= 5000:
            true = orc.distinct_bits64(v.view(np.uint64), b, n=n).distinct
            assert abs(res[k].distinct / true - 1) < 0.03  # "within 3 %" (approx_count_distinct.rs:346)
    if n:
        assert np.array_equal(registers_of(st), orc.hll_registers(cols[2][0].view(np.int64), cols[2][1], n=n))
    o = orc.stats(cols[1][0], cols[1][1])
    if n:
        assert (res[3].min_i, res[3].max_i, res[3].sum_i, res[3].non_null) == (o.min_i, o.max_i, o.sum_i_wrapping, o.non_null)
    assert (res[1].total, res[1].non_null) == (n, n)


def test_int32_float32_streams_merge_and_ranks(monkeypatch):
    from test_gpu_distributed_sim import _run_ranks
    from test_gpu_numeric32 import col32
    from term_amd.distributed import shard_rows

    rng = np.random.default_rng(2)
    n = 640_000
    i32 = rng.integers(-2**31, 2**31, size=n, dtype=np.int64).astype(np.int32)
    i32[: n // 4] = i32[n // 4: n // 2]
    f32 = (rng.standard_normal(n) * 50).astype(np.float32)
    vb = orc.pack_validity(rng.random(n) >= 0.1)
    want = [orc.hll_registers(i32.astype(np.int64), vb), orc.hll_registers(f32.astype(np.float64).view(np.int64))]
    specs = [spec(T.APPROX_DISTINCT, 0), spec(T.APPROX_DISTINCT, 1)]
    T.init()
    plan = T.Plan(specs)
    # a stream of 8192-row batches (coalesced), three states merged in reverse, a blob round trip
    parts = []
    for lo, hi in ((0, 200_000), (200_000, 200_064), (200_064, n)):
        s = T.State(plan)
        for a in range(lo, hi, 8192):
            b = min(hi, a + 8192)
            s.update([col32(i32, vb, a % 2 == 0, offset=a, length=b - a), col32(f32, None, a % 2 == 0, offset=a, length=b - a)])
        parts.append(s)
    merged = T.State(plan)
    merged.merge(list(reversed(parts)))
    res = merged.finalize()
    assert [r.distinct for r in res] == [orc.hll_estimate(w) for w in want]
    assert np.array_equal(registers_of(merged), want[1])
    back = T.State.deserialize(plan, merged.serialize())
    assert [r.distinct for r in back.finalize()] == [r.distinct for r in res]
    assert (res[0].total, res[0].non_null) == (n, orc.count(vb, n).non_null)

    # row shards over 4 ranks: every rank ends with the table's registers
    def shards_of(rank):
        lo, hi = shard_rows(n, 4, rank)
        return [col32(i32, vb, True, offset=lo, length=hi - lo), col32(f32, None, True, offset=lo, length=hi - lo)]

    for r, _st in _run_ranks(4, plan, shards_of):
        assert [x.distinct for x in r] == [orc.hll_estimate(w) for w in want]


def test_exact_key_set_answers_where_the_lane_does_not_apply():
    import pyarrow as pa

    rng = np.random.default_rng(3)
    n = 100_000
    v = rng.integers(0, 5000, size=n, dtype=np.int64)
    true = len(np.unique(v))
    T.init()
    # (a) an exact DISTINCT check of the same column: one key set answers both; (b) variance lanes on the column's scan
    for specs in ([spec(T.APPROX_DISTINCT, 0), spec(T.DISTINCT, 0)],
                  [spec(T.APPROX_DISTINCT, 0), spec(T.NUMERIC_STATS, 0, flags=T.FLAG_VARIANCE)]):
        st = T.State(T.Plan(specs))
        st.update([numeric_column(v, None, True)])
        res = st.finalize()
        assert res[0].distinct == true and (res[0].total, res[0].non_null) == (n, n)
    # (c) a string column
    words = ["w%d" % (i % 777) for i in range(20_000)] + [None] * 5
    arr = pa.array(words, type=pa.string())
    st = T.State(T.Plan([spec(T.APPROX_DISTINCT, 0), spec(T.COUNT, 0)]))
    st.update([T.Column.from_arrow(arr)])
    res = st.finalize()
    assert (res[0].distinct, res[0].total, res[0].non_null) == (777, 20_005, 20_000)
    assert (res[1].total, res[1].non_null) == (20_005, 20_000)


def test_merge_refuses_mixed_forms_before_it_takes_anything():
    """tgx_merge found out that one source holds HyperLogLog registers and another a key set AFTER it had
    folded the first sources' accumulators into dst.  All sources are checked first: a refused merge leaves dst as it
    was."""
    import pyarrow as pa

    rng = np.random.default_rng(6)
    n = 50_000
    nums = rng.integers(0, 5000, size=n, dtype=np.int64)
    T.init()
    plan = T.Plan([spec(T.APPROX_DISTINCT, 0), spec(T.COUNT, 0)])
    a, b, c, dst = T.State(plan), T.State(plan), T.State(plan), T.State(plan)
    a.update([numeric_column(nums, None, True)])
    b.update([numeric_column(nums, None, True)])
    c.update([T.Column.from_arrow(pa.array(["s%d" % (i % 77) for i in range(1000)]))])  # a string column: the key set
    dst.update([numeric_column(nums[:1000], None, True)])
    before = dst.finalize()
    with pytest.raises(T.TgxError) as e:
        dst.merge([a, b, c])
    assert e.value.status == "TGX_INVALID_ARGUMENT" and "nothing was merged" in str(e.value)
    after = dst.finalize()
    assert (after[1].total, after[0].distinct) == (before[1].total, before[0].distinct) and after[1].total == 1000
    dst.merge([a, b])
    assert dst.finalize()[1].total == 1000 + 2 * n
