"""-m gpu: COUNT(DISTINCT) of big batches of keys WITHOUT a dense range (sparse Int64 ids, Float64 bit patterns) -- the
mixed keys partitioned into lists and deduplicated list by list in LDS (kernels/distinct.hip, key_*; kernels/lists.h)
-- vs the oracle, bit-exact.  The path normally starts at 2 Mi rows; TGX_FP_LISTS_MIN_ROWS lowers that, and the
"distinct_lists" profile entry proves which path ran.  One test runs at the real threshold."""
import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import numeric_column

pytestmark = pytest.mark.gpu


@pytest.fixture
def low_threshold(monkeypatch):
    monkeypatch.setenv("TGX_FP_LISTS_MIN_ROWS", "1000")


def check(res, d):
    assert (res.total, res.non_null, res.distinct, res.groups_once) == (d.total, d.non_null, d.distinct, d.groups_once)


def run(specs, cols):
    T.init()
    plan = T.Plan(specs)
    st = T.State(plan)
    st.profile_enable()
    st.update(cols)
    return st.finalize(), st, plan


def took_lists(st):
    return st.profile_get("distinct_lists")["launches"]


def sparse_keys(rng, n, card):
    pool = rng.integers(-2**63, 2**63 - 1, size=card, dtype=np.int64)
    pool[:3] = [-1, np.iinfo(np.int64).min, np.iinfo(np.int64).max]  # -1 is the table's free-slot marker
    return pool[rng.integers(0, card, size=n)]


@pytest.mark.parametrize("device", [True, False])
@pytest.mark.parametrize("n,card", [(300_000, 10**7), (200_000, 40_000), (100_000, 3)])
def test_sparse_int64_keys(low_threshold, n, card, device):
    """all distinct / every key a few times / three keys thousands of times each (their lists overflow: the batch is
    redone through the table); NULLs; the all-ones key counted on the side"""
    rng = np.random.default_rng(n + card % 997 + device)
    keys = sparse_keys(rng, n, card)
    mask = rng.random(n) >= 0.07
    valid = orc.pack_validity(mask)
    want = orc.distinct_bits64(keys.view(np.uint64), valid)
    res, st, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.NUMERIC_STATS, 0)],
                     [numeric_column(keys, valid, device)])
    assert took_lists(st) == 1
    check(res[0], want)
    assert res[1].non_null == want.non_null
    check(st.finalize()[0], want)  # finalize does not consume the lists


def test_float64_bit_patterns(low_threshold):
    rng = np.random.default_rng(5)
    n = 250_000
    v = rng.standard_normal(n) * 10.0 ** rng.integers(-200, 200, size=n)
    pool = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 1.7976931348623157e308])
    sel = rng.random(n) < 0.3
    v[sel] = rng.choice(pool, size=int(sel.sum()))
    v[::7] = v[3]  # and a value that repeats
    want = orc.distinct_bits64(v.view(np.uint64), None)
    res, st, _ = run([spec(T.DISTINCT, 0)], [numeric_column(v, None, True)])
    assert took_lists(st) == 1
    assert (res[0].total, res[0].non_null, res[0].distinct) == (want.total, want.non_null, want.distinct)


def test_second_batch_merge_serialize_and_exchange(low_threshold):
    from test_gpu_distributed_sim import _run_ranks

    rng = np.random.default_rng(8)
    n = 240_000
    keys = sparse_keys(rng, n, 90_000)
    mask = rng.random(n) >= 0.05
    valid = orc.pack_validity(mask)
    want = orc.distinct_bits64(keys.view(np.uint64), valid)
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)])
    cut = 131_072  # (validity bytes line up with the slice)
    first = [numeric_column(keys, valid, True, offset=0, length=cut)]
    second = [numeric_column(keys, valid, True, offset=cut, length=n - cut)]
    a = T.State(plan)
    a.profile_enable()
    a.update(first)
    a.update(second)   # the table takes over from the lists
    assert took_lists(a) == 1
    check(a.finalize()[0], want)
    b, c = T.State(plan), T.State(plan)
    b.update(first)
    c.update(second)
    u = T.State.deserialize(plan, b.serialize())
    u.merge([T.State.deserialize(plan, c.serialize())])
    check(u.finalize()[0], want)
    b.merge([c])
    check(b.finalize()[0], want)
    bounds = [0, 65_536, cut, n]

    def shards_of(rank):
        return [numeric_column(keys, valid, True, offset=bounds[rank], length=bounds[rank + 1] - bounds[rank])]

    for res, _ in _run_ranks(3, plan, shards_of):
        check(res[0], want)


def test_real_threshold_sparse_ids():
    """5 M sparse ids (every id twice) at the default threshold, on the device: generated with torch, checked by the
    closed form; then the same rows as two batches"""
    import torch

    n = 5_000_000
    g = torch.Generator(device="cuda").manual_seed(11)
    half = torch.randint(-2**62, 2**62, (n // 2,), dtype=torch.int64, device="cuda", generator=g)
    distinct_half = int(torch.unique(half).numel())
    keys = torch.cat([half, half])[torch.randperm(n, device="cuda", generator=g)]
    col = T.Column.int64(keys, None, length=n)
    res, st, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)], [col])
    assert took_lists(st) == 1
    r = res[0]
    assert (r.total, r.non_null, r.distinct) == (n, n, distinct_half)
    assert r.groups_once == 0  # every id is there twice (or four times, should torch.randint repeat itself)
    st.reset()
    st.update([col.sliced(0, n // 2)])
    st.update([col.sliced(n // 2, n - n // 2)])
    r2 = st.finalize()[0]
    assert (r2.total, r2.non_null, r2.distinct, r2.groups_once) == (r.total, r.non_null, r.distinct, r.groups_once)


_M64 = (1 << 64) - 1


def _unmix64(x):
    """inverse of the splitmix64 finaliser the key lists partition by (kernels/distinct.hip, unmix64)"""
    x = np.asarray(x, dtype=np.uint64).copy()
    x ^= (x >> np.uint64(31)) ^ (x >> np.uint64(62))
    x *= np.uint64(0x319642B2D24D8EC3)
    x ^= (x >> np.uint64(27)) ^ (x >> np.uint64(54))
    x *= np.uint64(0x96DE1B173F119089)
    x ^= (x >> np.uint64(30)) ^ (x >> np.uint64(60))
    return x


@pytest.mark.parametrize("cap2", [17_000, 24_000])
@pytest.mark.parametrize("shape", ["distinct", "repeats", "crowded"])
def test_long_lists_counted_in_lds(monkeypatch, cap2, shape):
    """The count of LONG lists (more than 12 288 records: what 800 M+ keys produce) on a few hundred thousand keys
    chosen -- through the inverse of the mix -- to fall into a dozen of the 65 536 lists, with room made by
    TGX_FP_LIST_CAPS (the 32 Ki-slot table at one workgroup a CU: before round 5 only runs of that size reached it).
    distinct: every key once; repeats: every key 1-6 times, some a thousand times; crowded: a third of a list's records
    start at the SAME slot, another third share a tag with different keys (probe chains thousands of slots long)."""
    monkeypatch.setenv("TGX_FP_LISTS_MIN_ROWS", "1000")
    monkeypatch.setenv("TGX_FP_LIST_CAPS", "%d:%d" % (cap2 + 2_000, cap2))
    rng = np.random.default_rng([cap2, len(shape)])
    lists = rng.choice(65536, size=12, replace=False).astype(np.uint64)
    per = cap2 - 900
    mixed = []
    for li in lists:
        low = rng.integers(0, 1 << 48, size=per, dtype=np.uint64)
        if shape == "repeats":
            pool = low[: per // 3]
            low = pool[rng.integers(0, len(pool), size=per)]
            low[: 1000] = pool[0]
        elif shape == "crowded":
            third = per // 3
            low[:third] = (low[:third] & np.uint64(0xFFFF_8000_FFFF_FFFF)) | (np.uint64(12345) << np.uint64(32))  # one slot
            low[third:2 * third] = (low[third:2 * third] & np.uint64(0xFFFF_FFFF_FFFF_0000)) | np.uint64(0xBEEF)  # one tag
            low[-50:] = low[:50]
        mixed.append((li << np.uint64(48)) | low)
    mixed = np.concatenate(mixed)
    rng.shuffle(mixed)
    keys = _unmix64(mixed).view(np.int64)
    n = len(keys)
    mask = rng.random(n) >= 0.03
    valid = orc.pack_validity(mask)
    want = orc.distinct_bits64(keys.view(np.uint64), valid)
    res, st, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)], [numeric_column(keys, valid, True)])
    assert took_lists(st) == 1
    check(res[0], want)
    res, st, _ = run([spec(T.DISTINCT, 0)], [numeric_column(keys, None, True)])
    want = orc.distinct_bits64(keys.view(np.uint64), None)
    assert (res[0].non_null, res[0].distinct) == (want.non_null, want.distinct)
