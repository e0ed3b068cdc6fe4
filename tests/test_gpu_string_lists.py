"""-m gpu: COUNT(DISTINCT) of big Utf8 batches -- the fingerprints partitioned into lists and deduplicated list by list
in LDS (kernels/distinct128.hip, fp_*) -- vs the oracle.  The path normally starts at 2 Mi rows; TGX_FP_LISTS_MIN_ROWS
lowers that so the oracle still finishes in seconds, and the "distinct_lists" profile entry proves which path ran.
One test runs at the real threshold on a generated column whose answer is known in closed form."""
import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from test_gpu_regex import utf8_column
from test_gpu_strings import check, make_strings

pytestmark = pytest.mark.gpu


@pytest.fixture
def low_threshold(monkeypatch):
    monkeypatch.setenv("TGX_FP_LISTS_MIN_ROWS", "1000")


def run(specs, col):
    T.init()
    plan = T.Plan(specs)
    st = T.State(plan)
    st.profile_enable()
    st.update(col if isinstance(col, list) else [col])
    res = st.finalize()
    return res, st, plan


def took_lists(st):
    return st.profile_get("distinct_lists")["launches"]


@pytest.mark.parametrize("n,card", [(300_000, 10**9), (200_000, 30_000), (150_001, 75_000)])
@pytest.mark.parametrize("large", [False, True])
def test_lists_vs_oracle(low_threshold, n, card, large):
    rng = np.random.default_rng(n % 977 + card % 1000)
    vals = make_strings(rng, n, card)
    offs, data, validity = orc.utf8_from_list(vals)
    want = orc.distinct_utf8(offs, data, validity)
    res, st, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.COUNT, 0)],
                     utf8_column(offs, data, validity, True, large=large))
    assert took_lists(st) == 1
    check(res[0], want)
    assert res[1].non_null == want.non_null
    # finalize does not consume the lists: a second look gives the same answer
    check(st.finalize()[0], want)


def test_without_multiplicity_and_without_validity(low_threshold):
    rng = np.random.default_rng(5)
    vals = [v if v is not None else "was-null" for v in make_strings(rng, 100_000, 60_000)]
    offs, data, _ = orc.utf8_from_list(vals)
    want = orc.distinct_utf8(offs, data, None)
    res, st, _ = run([spec(T.DISTINCT, 0)], utf8_column(offs, data, None, True))
    assert took_lists(st) == 1
    assert (res[0].total, res[0].non_null, res[0].distinct) == (want.total, want.non_null, want.distinct)


def test_long_values_that_do_not_fit_the_stage(low_threshold):
    """128 consecutive values of more than 4 KiB in all are fingerprinted from global memory, shorter stretches from
    LDS: the two must agree (the same value appears in both kinds of stretch)"""
    rng = np.random.default_rng(11)
    vals = []
    for block in range(300):
        long_block = block % 3 == 0
        for i in range(128):
            k = int(rng.integers(0, 5000))
            vals.append(("L%d-" % k) + "y" * ((k % 200) if long_block else (k % 9)) if k % 17 else None)
        vals.append("q" * 5000 if block % 50 == 0 else "")  # (shifts the 128-row steps against the blocks)
    offs, data, validity = orc.utf8_from_list(vals)
    want = orc.distinct_utf8(offs, data, validity)
    for large in (False, True):
        res, st, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)],
                         utf8_column(offs, data, validity, True, large=large))
        assert took_lists(st) == 1
        check(res[0], want)


@pytest.mark.parametrize("device", [True, False])
@pytest.mark.parametrize("n,card,n_buffers", [(120_000, 10**9, 3), (120_000, 10**9, 1), (100_000, 20_000, 1),
                                              (100_000, 20_000, 2), (50_000, 4, 3)])
def test_utf8view_batches(low_threshold, n, card, n_buffers, device):
    """Utf8View: inline and out-of-line values, junk views under NULLs; with ONE data buffer the long values of
    consecutive rows lie one after the other (what Arrow's builders produce: the kernel stages their span in LDS),
    with several they are scattered (read from global memory, as are the rare 5000-byte values); card 4 overflows the
    lists (the batch is redone through the table, for HOST batches before tgx_update returns)"""
    from test_gpu_utf8view import view_column

    rng = np.random.default_rng(n + card % 1000 + device + 7 * n_buffers)
    vals = make_strings(rng, n, card)
    vals[1000] = "z" * 5000  # (longer than the stage: that step falls back to global memory)
    offs, data, validity = orc.utf8_from_list(vals)
    want = orc.distinct_utf8(offs, data, validity)
    res, st, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)], view_column(vals, rng, device, n_buffers=n_buffers))
    assert took_lists(st) == 1
    check(res[0], want)
    # a second batch (the table takes over from the lists), sliced with a lead of junk views
    lead = 37
    st.update([view_column([None] * lead + vals[: n // 3], rng, device, offset=lead, length=n // 3)])
    check(st.finalize()[0], want.__class__(total=want.total + n // 3, non_null=want.non_null + orc.distinct_utf8(
        offs, data, validity, n=n // 3).non_null, distinct=want.distinct, groups_once=orc.distinct_utf8(
            *orc.utf8_from_list(vals + vals[: n // 3])).groups_once))


def test_sliced_column(low_threshold):
    rng = np.random.default_rng(6)
    vals = make_strings(rng, 90_000, 10**9)
    offs, data, validity = orc.utf8_from_list(vals)
    lo, n = 12_345, 70_001
    want = orc.distinct_utf8(offs, data, validity, n=n, offset=lo)
    res, st, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)],
                     utf8_column(offs, data, validity, True, offset=lo, length=n))
    assert took_lists(st) == 1
    check(res[0], want)


@pytest.mark.parametrize("card", [1, 50])
def test_repeated_values_overflow_the_lists_and_the_batch_is_redone(low_threshold, card):
    """a few values repeated thousands of times: their lists overflow, the result must still be exact"""
    rng = np.random.default_rng(7 + card)
    vals = make_strings(rng, 60_000, card)
    offs, data, validity = orc.utf8_from_list(vals)
    want = orc.distinct_utf8(offs, data, validity)
    res, st, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)], utf8_column(offs, data, validity, True))
    assert took_lists(st) == 1
    check(res[0], want)
    check(st.finalize()[0], want)


def test_second_batch_merge_and_serialize_after_the_lists(low_threshold):
    rng = np.random.default_rng(8)
    n = 160_000
    vals = make_strings(rng, n, 50_000)
    offs, data, validity = orc.utf8_from_list(vals)
    want = orc.distinct_utf8(offs, data, validity)
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)])
    cut = 100_000
    first = utf8_column(offs, data, validity, True, offset=0, length=cut)
    second = utf8_column(offs, data, validity, True, offset=cut, length=n - cut)
    # (1) lists, then a second batch into the same state
    a = T.State(plan)
    a.profile_enable()
    a.update([first])
    a.update([second])
    assert took_lists(a) == 1
    check(a.finalize()[0], want)
    # (2) two states, each on the lists, united through serialize -> deserialize -> merge
    b, c = T.State(plan), T.State(plan)
    b.update([first])
    c.update([second])
    check(b.finalize()[0], orc.distinct_utf8(offs, data, validity, n=cut, offset=0))
    u = T.State.deserialize(plan, b.serialize())
    u.merge([T.State.deserialize(plan, c.serialize())])
    check(u.finalize()[0], want)
    # (3) direct merge of a state that still holds lists into one that does too
    d, e = T.State(plan), T.State(plan)
    d.update([first])
    e.update([second])
    d.merge([e])
    check(d.finalize()[0], want)
    # (4) reset and reuse
    d.reset()
    d.update([second])
    check(d.finalize()[0], orc.distinct_utf8(offs, data, validity, n=n - cut, offset=cut))


def test_host_batches(low_threshold):
    rng = np.random.default_rng(9)
    vals = make_strings(rng, 50_000, 10**9)
    offs, data, validity = orc.utf8_from_list(vals)
    want = orc.distinct_utf8(offs, data, validity)
    res, st, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)], utf8_column(offs, data, validity, False))
    assert took_lists(st) == 1
    check(res[0], want)
    dup = make_strings(rng, 50_000, 3)  # overflows: redone before tgx_update returns, from the staged copy
    offs, data, validity = orc.utf8_from_list(dup)
    res, st, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)], utf8_column(offs, data, validity, False))
    check(res[0], orc.distinct_utf8(offs, data, validity))


def test_sharded_exchange_after_the_lists(low_threshold):
    """tgx_allreduce over 3 simulated ranks whose key sets are still lists when the exchange starts"""
    from test_gpu_distributed_sim import _run_ranks

    rng = np.random.default_rng(10)
    n = 90_000
    vals = make_strings(rng, n, 20_000)
    offs, data, validity = orc.utf8_from_list(vals)
    want = orc.distinct_utf8(offs, data, validity)
    specs = [spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)]
    cuts = [0, 30_000, 30_000 + 25_000, n]

    def shards_of(rank):
        return [utf8_column(offs, data, validity, True, offset=cuts[rank], length=cuts[rank + 1] - cuts[rank])]

    T.init()
    for res, _ in _run_ranks(3, T.Plan(specs), shards_of):
        check(res[0], want)


def test_real_threshold_closed_form():
    """20 M generated e-mail rows at the default threshold: every non-NULL row is a distinct string"""
    import torch
    from test_gpu_configs import _email_column

    n = 20_000_000
    offsets, data, validity, L, expect = _email_column(torch, n)
    col = T.Column(T.LARGE_UTF8, n, offsets=offsets, data=data, validity=validity)
    res, st, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)], col)
    assert took_lists(st) == 1
    r = res[0]
    assert (r.total, r.non_null, r.distinct, r.groups_once) == (n, n - expect["nulls"], n - expect["nulls"],
                                                               n - expect["nulls"])
    # the same rows as two batches (lists, then the table takes both): same answer
    half = n // 2
    st.reset()
    st.update([col.sliced(0, half)])
    st.update([col.sliced(half, n - half)])
    r2 = st.finalize()[0]
    assert (r2.total, r2.non_null, r2.distinct, r2.groups_once) == (r.total, r.non_null, r.distinct, r.groups_once)


def test_batch_beyond_the_small_table_closed_form():
    """170 M generated e-mail rows: a final list holds more than the 3072 records the 16 KiB table takes, so the count
    pass runs with its 64 KiB table (the 128 KiB one, past ~0.6 G rows, is exercised by tools/bench_strings.py
    --rows 900000000, which asserts the same closed form)"""
    import torch
    from test_gpu_configs import _email_column

    n = 170_000_000
    offsets, data, validity, L, expect = _email_column(torch, n)
    col = T.Column(T.LARGE_UTF8, n, offsets=offsets, data=data, validity=validity)
    res, st, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)], col)
    assert took_lists(st) == 1
    r = res[0]
    assert (r.total, r.non_null, r.distinct, r.groups_once) == (n, n - expect["nulls"], n - expect["nulls"],
                                                               n - expect["nulls"])


def test_tuples_through_the_lists(low_threshold):
    """COUNT(DISTINCT (a, b, ...)): the first big batch of a tuple task takes the lists too (every row is a record;
    rows whose components are all non-NULL are counted on the side).  The tuple test of test_gpu_strings.py, run with
    the low threshold: (string, int) tuples fit the lists, (int, float) tuples repeat hundreds of times (overflow:
    redone through the table), batches / serialize / merge after the lists."""
    from collections import Counter

    from gpu_util import numeric_column
    import test_gpu_strings

    test_gpu_strings.test_distinct_over_column_tuples()
    rng = np.random.default_rng(31)
    n = 80_000
    a = rng.integers(-2**40, 2**40, size=n, dtype=np.int64)
    b = rng.integers(0, 3, size=n, dtype=np.int64)
    amask = rng.random(n) >= 0.1
    av = orc.pack_validity(amask)
    cnt = Counter((int(a[i]) if amask[i] else None, int(b[i])) for i in range(n))
    res, st, _ = run([spec(T.DISTINCT, 0, columns=[0, 1], flags=T.FLAG_MULTIPLICITY)],
                     [numeric_column(a, av, True), numeric_column(b, None, True)])
    assert took_lists(st) == 1
    r = res[0]
    assert (r.total, r.non_null, r.distinct, r.groups_once) == \
        (n, int(amask.sum()), len(cnt), sum(1 for v in cnt.values() if v == 1))
