"""-m gpu: COUNT(DISTINCT) over byte-string keys when two DISTINCT VALUES SHARE A FINGERPRINT.

The reference counts by value: DataFusion's hash aggregation groups with hash + equality
(TG/constraints/uniqueness.rs:612-617, 671-681, 709-715).  The library reduces string / tuple keys to 128-bit keyed
fingerprints (kernels/distinct128.hip: Chaskey-8 under the plan's key); whoever holds the key can write down distinct
values with one fingerprint (tests/fp_reference.py does), whoever does not cannot.  Two things are pinned here:

* an EXACT key set (TGX_FLAG_EXACT_KEYS) counts such values as the oracle does on every route that feeds a state --
  the table (small batches), Utf8View, dictionaries, tuples, several batches, multiplicities -- because equal
  fingerprints are confirmed byte by byte; a plain fingerprint set counts them once (the deviation, asserted so that
  the test shows the collision is real);
* the pair the round-5 review derived against the old seedless function ("user0001@example.com/abc" /
  "user0102@example^:Z P/,W") and every other input count right on every route, exact or not, under a key drawn from
  the operating system -- including the routes on which only fingerprints travel (merge, blobs, ranks)."""
import numpy as np
import pytest

import fp_reference as F
import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import pad_validity, to_device
from test_gpu_regex import utf8_column
from test_gpu_strings import check

pytestmark = pytest.mark.gpu

KEY = bytes(range(0x10, 0x20))
REVIEW_PAIR = (b"user0001@example.com/abc", b"user0102@example^:Z P/,W")


@pytest.fixture(scope="module")
def pairs():
    rng = np.random.default_rng(61)
    out = [F.colliding_pair(KEY, rng) for _ in range(200)]
    out += [F.colliding_pair(KEY, rng, tail_len=15) for _ in range(3)]  # a padded last block (K2)
    assert len({v for p in out for v in p}) == 2 * len(out)
    return out


def column_of(values, device=True, large=False, offset=0, length=None):
    offs, data, validity = orc.utf8_from_list(values)
    return utf8_column(offs, data, validity, device, large=large, offset=offset, length=length), (offs, data, validity)


def background(rng, n, card):
    """ordinary values around the planted ones: e-mail-like, repeated, some NULL"""
    ks = rng.integers(0, card, size=n)
    return [None if k % 19 == 0 else b"user%07d@example%d.com" % (k, k % 1000) for k in ks]


def plant(values, pairs, rng, times=(1, 1)):
    """puts each value of each pair at `times` random places"""
    values = list(values)
    free = [i for i in rng.permutation(len(values))]
    for a, b in pairs:
        for v, t in ((a, times[0]), (b, times[1])):
            for _ in range(t):
                values[free.pop()] = v
    return values


def run(specs, batches, key=KEY, profile=False):
    T.init()
    plan = T.Plan(specs, fingerprint_key=key)
    st = T.State(plan)
    if profile:
        st.profile_enable()
    for b in batches:
        st.update(b)
    return st.finalize(), st, plan


def test_python_reference_matches_the_device_function(pairs):
    """the fingerprint the kernels compute IS fp_reference.fingerprint: a pair that collides there counts once in a
    fingerprint set -- for values of every length class (empty, < 16, = 16, 17..31, = 32, longer; padded and full last
    blocks) a planted partner with an equal fingerprint could not be told apart otherwise.  Checked through exported
    key records: tgx_distinct_export hands out (fa, fb) per key."""
    rng = np.random.default_rng(3)
    vals = [b"", b"a", b"0123456789abcde", b"0123456789abcdef", b"0123456789abcdefg", bytes(31), bytes(32), bytes(33),
            bytes(rng.integers(0, 256, size=100, dtype=np.uint8)), REVIEW_PAIR[0], REVIEW_PAIR[1]]
    col, _ = column_of(vals)
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0)], fingerprint_key=KEY)
    st = T.State(plan)
    st.update([col])
    recs = st.distinct_export_records(0)
    got = {(int(r[0]), int(r[1])) for r in recs}
    assert got == {F.fingerprint(KEY, v) for v in vals}
    # the same keys from an exact set (its slots hold references: the second word comes from the key store)
    plan_x = T.Plan([spec(T.DISTINCT, 0, flags=T.FLAG_EXACT_KEYS)], fingerprint_key=KEY)
    sx = T.State(plan_x)
    sx.update([col])
    assert {(int(r[0]), int(r[1])) for r in sx.distinct_export_records(0)} == got
    assert plan.fingerprint_key() == KEY


def test_device_function_reproduces_the_published_chaskey8_vectors():
    """the kernels' fingerprint of the published messages under the published key IS the published tag (the known
    answers of the Chaskey authors' reference implementation: tests/test_fingerprint_kat.py has them and holds the
    Python statement to them on the CPU): the records tgx_distinct_export hands out are (v0 | v1 << 32, v2 | v3 << 32)"""
    import struct

    from test_fingerprint_kat import KEY as KAT_KEY, KNOWN

    vals = [bytes(range(n)) for n, _ in KNOWN]
    col, _ = column_of(vals)
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0)], fingerprint_key=KAT_KEY)
    st = T.State(plan)
    st.update([col])
    got = {(int(r[0]), int(r[1])) for r in st.distinct_export_records(0)}
    assert got == {(w[0] | (w[1] << 32), w[2] | (w[3] << 32)) for _, w in KNOWN}
    assert struct.unpack("<4I", plan.fingerprint_key()) == struct.unpack("<4I", KAT_KEY)


@pytest.mark.parametrize("device", [True, False])
@pytest.mark.parametrize("large", [False, True])
def test_small_batches_table_path(pairs, device, large):
    rng = np.random.default_rng(7)
    vals = plant(background(rng, 20_000, 3_000), pairs, rng, times=(1, 2))
    col, (offs, data, validity) = column_of(vals, device=device, large=large)
    want = orc.distinct_utf8(offs, data, validity)
    res, _, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY | T.FLAG_EXACT_KEYS)], [[col]])
    check(res[0], want)
    # a fingerprint set sees one key per pair: three rows of it, never "once"
    res, _, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)], [[col]])
    assert res[0].distinct == want.distinct - len(pairs)
    assert res[0].groups_once == want.groups_once - len(pairs)
    assert (res[0].total, res[0].non_null) == (want.total, want.non_null)


def lists_ran(st):
    return st.profile_get("distinct_lists")["launches"]


@pytest.mark.parametrize("large", [False, True])
def test_big_batch_on_the_lists(pairs, monkeypatch, large):
    """a batch big enough for the fingerprint lists (threshold lowered: the oracle has to finish).  An exact set over the
    caller's DEVICE buffers takes the lists too: its records carry rows, and fp_count_kernel settles equal fingerprints
    on the rows' bytes (ExactUtf8Eq).  A HOST batch of an exact set goes to the table (its staged copy does not outlive
    the update)."""
    monkeypatch.setenv("TGX_FP_LISTS_MIN_ROWS", "1000")
    rng = np.random.default_rng(8)
    vals = plant(background(rng, 300_000, 10**9), pairs, rng, times=(2, 1))
    col, (offs, data, validity) = column_of(vals, large=large)
    want = orc.distinct_utf8(offs, data, validity)
    sp = [spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY | T.FLAG_EXACT_KEYS)]
    res, st, _ = run(sp, [[col]], profile=True)
    assert lists_ran(st) == 1
    check(res[0], want)
    check(st.finalize()[0], want)  # (a second look)
    res, st, _ = run(sp, [[column_of(vals, device=False, large=large)[0]]], profile=True)
    assert lists_ran(st) == 0
    check(res[0], want)
    res, st, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)], [[col]], profile=True)
    assert lists_ran(st) == 1
    assert res[0].distinct == want.distinct - len(pairs)


def test_lists_then_more(pairs, monkeypatch):
    """what follows an exact set's lists: a second batch, tgx_state_sync, a blob or a merge BEFORE tgx_finalize find the
    first batch still there and move its keys into the table with their bytes (partners that arrive later are told
    apart); AFTER tgx_finalize the caller may have released the batch, and its keys go on as their 128-bit fingerprints
    (include/tgx.h, TGX_FLAG_EXACT_KEYS)."""
    monkeypatch.setenv("TGX_FP_LISTS_MIN_ROWS", "1000")
    rng = np.random.default_rng(18)
    n1, n2 = 120_000, 30_000
    first = background(rng, n1, 10**9)
    second = background(rng, n2, 10**9)
    for j, (a, b) in enumerate(pairs):
        first[100 + 13 * j] = a
        (first if j % 2 else second)[50 + 11 * j] = b  # every other pair is split between the batches
    split = sum(1 for j in range(len(pairs)) if j % 2 == 0)
    col1, col2 = column_of(first)[0], column_of(second)[0]
    offs, data, validity = orc.utf8_from_list(first + second)
    want = orc.distinct_utf8(offs, data, validity)
    sp = [spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY | T.FLAG_EXACT_KEYS)]
    T.init()
    plan = T.Plan(sp, fingerprint_key=KEY)
    # (1) a second batch straight away
    st = T.State(plan)
    st.profile_enable()
    st.update([col1])
    st.update([col2])
    assert lists_ran(st) == 1
    check(st.finalize()[0], want)
    # (2) tgx_state_sync in between (the caller may release the first batch after it)
    st = T.State(plan)
    st.update([col1])
    st.sync()
    st.update([col2])
    check(st.finalize()[0], want)
    # (3) merge of two states that are both on the lists: each converts with its bytes, the union is by fingerprint
    a, b = T.State(plan), T.State(plan)
    a.update([col1])
    b.update([col2])
    a.merge([b])
    got = a.finalize()[0]
    assert (got.total, got.non_null, got.distinct) == (want.total, want.non_null, want.distinct - split)
    # (4) finalize first: the batch is the caller's again, the keys of the lists are fingerprints from here on
    st = T.State(plan)
    st.update([col1])
    o1 = orc.utf8_from_list(first)
    check(st.finalize()[0], orc.distinct_utf8(*o1))
    st.update([col2])
    got = st.finalize()[0]
    stands_for = {b: a for a, b in pairs}  # one fingerprint per pair
    by_fingerprint = len({stands_for.get(v, v) for v in first + second if v is not None})
    assert (got.total, got.non_null, got.distinct) == (want.total, want.non_null, by_fingerprint)


def test_views_and_tuples_on_the_exact_lists(pairs, monkeypatch):
    from test_gpu_utf8view import view_column

    monkeypatch.setenv("TGX_FP_LISTS_MIN_ROWS", "1000")
    rng = np.random.default_rng(19)
    vals = plant(background(rng, 100_000, 10**9) + [b"s%d" % (i % 300) for i in range(20_000)], pairs, rng)
    offs, data, validity = orc.utf8_from_list(vals)
    want = orc.distinct_utf8(offs, data, validity)
    res, st, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY | T.FLAG_EXACT_KEYS)], [[view_column(vals, rng, True)]],
                     profile=True)
    assert lists_ran(st) == 1
    check(res[0], want)
    # tuples (Int64, Utf8) through the lists
    n = len(vals)
    ints = rng.integers(0, 3, size=n).astype(np.int64)
    where = {v: i for i, v in enumerate(vals)}
    for a, b in pairs:
        ints[where[b]] = ints[where[a]]
    col_i = T.Column.int64(to_device(ints), None)
    col_s, _ = column_of(vals)
    want_t = len({(int(ints[i]), vals[i]) for i in range(n)})
    res, st, _ = run([spec(T.DISTINCT, 0, columns=[0, 1], flags=T.FLAG_EXACT_KEYS)], [[col_i, col_s]], profile=True)
    assert lists_ran(st) == 1 and res[0].distinct == want_t
    res, st, _ = run([spec(T.DISTINCT, 0, columns=[0, 1])], [[col_i, col_s]], profile=True)
    assert lists_ran(st) == 1 and res[0].distinct == want_t - len(pairs)


def test_partners_in_different_batches_and_store_growth(pairs):
    """the partner of a key arrives batches later: the first value's bytes are in the key store by then.  Many small
    batches of mostly-new keys also walk the store through several re-allocations."""
    rng = np.random.default_rng(9)
    n_b, per = 12, 40_000
    firsts = [p[0] for p in pairs]
    seconds = [p[1] for p in pairs]
    batches, everything = [], []
    for k in range(n_b):
        vals = [b"batch%02d-row%06d-%s" % (k, i, b"x" * (i % 23)) for i in range(per)]
        for j, v in enumerate(firsts if k == 1 else seconds if k == n_b - 2 else []):
            vals[17 * j + 5] = v
        everything += vals
        batches.append([column_of(vals)[0]])
    offs, data, validity = orc.utf8_from_list(everything)
    want = orc.distinct_utf8(offs, data, validity)
    res, _, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_EXACT_KEYS)], batches)
    assert (res[0].total, res[0].non_null, res[0].distinct) == (want.total, want.non_null, want.distinct)
    res, _, _ = run([spec(T.DISTINCT, 0)], batches)
    assert res[0].distinct == want.distinct - len(pairs)


@pytest.mark.parametrize("device", [True, False])
def test_utf8view(pairs, device):
    from test_gpu_utf8view import view_column

    rng = np.random.default_rng(10)
    vals = plant(background(rng, 30_000, 4_000) + [b"s%d" % (i % 300) for i in range(5_000)], pairs, rng)
    offs, data, validity = orc.utf8_from_list(vals)
    want = orc.distinct_utf8(offs, data, validity)
    col = view_column(vals, rng, device, n_buffers=3)
    res, _, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY | T.FLAG_EXACT_KEYS)], [[col]])
    check(res[0], want)
    res, _, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)], [[col]])
    assert res[0].distinct == want.distinct - len(pairs)


@pytest.mark.parametrize("device", [True, False])
def test_dictionary(pairs, device):
    from test_gpu_dictionary import encode

    rng = np.random.default_rng(11)
    vals = plant([v if v is not None else None for v in background(rng, 60_000, 2_000)], pairs, rng, times=(2, 1))
    offs, data, validity = orc.utf8_from_list(vals)
    want = orc.distinct_utf8(offs, data, validity)
    col = encode(vals, rng, extra_entries=[b"never-referenced"], repeat_entries=True, device=device)
    res, _, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY | T.FLAG_EXACT_KEYS)], [[col]])
    check(res[0], want)
    res, _, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)], [[col]])
    assert res[0].distinct == want.distinct - len(pairs)


def test_tuples_with_a_string_component(pairs):
    """COUNT(DISTINCT (a, b)) (uniqueness.rs:557-562): tuples whose string components collide have one tuple
    fingerprint; an exact set compares the components"""
    rng = np.random.default_rng(12)
    n = 30_000
    strs = plant([b"v%d" % (i % 900) for i in range(n)], pairs, rng)
    ints = rng.integers(0, 5, size=n).astype(np.int64)
    where = {}
    for i, s in enumerate(strs):
        where[s] = i
    mask = rng.random(n) >= 0.03
    for a, b in pairs:  # the two members of a pair sit next to the SAME integer (or NULL): only the strings tell them apart
        ints[where[b]] = ints[where[a]]
        mask[where[b]] = mask[where[a]]
    ival = orc.pack_validity(mask)
    col_s, _ = column_of(strs)
    col_i = T.Column.int64(to_device(ints), to_device(pad_validity(ival)))
    want = len({(int(ints[i]) if mask[i] else None, strs[i]) for i in range(n)})
    specs = lambda flags: [spec(T.DISTINCT, 0, columns=[0, 1], flags=flags)]
    res, _, _ = run(specs(T.FLAG_EXACT_KEYS), [[col_i, col_s]])
    assert res[0].distinct == want
    res, _, _ = run(specs(0), [[col_i, col_s]])
    assert res[0].distinct == want - len(pairs)


def test_fixed_width_keys_decimal256():
    """Decimal256 / FixedSizeBinary(32) keys are 32-byte strings to COUNT(DISTINCT) (term_amd/_lib.py from_arrow): a
    colliding pair of 32-byte values is a pair of such keys"""
    pa = pytest.importorskip("pyarrow")
    rng = np.random.default_rng(13)
    planted = [F.colliding_pair(KEY, rng) for _ in range(20)]
    vals = [bytes(rng.integers(0, 256, size=32, dtype=np.uint8)) for _ in range(5000)]
    vals += [v for p in planted for v in p] + vals[:100]
    arr = pa.array(vals, type=pa.binary(32))
    col = T.Column.from_arrow(arr)
    want = len(set(vals))
    res, _, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_EXACT_KEYS)], [[col]])
    assert res[0].distinct == want
    res, _, _ = run([spec(T.DISTINCT, 0)], [[col]])
    assert res[0].distinct == want - len(planted)


def test_reset_and_reuse_of_an_exact_state(pairs):
    rng = np.random.default_rng(14)
    vals = plant(background(rng, 10_000, 500), pairs[:20], rng)
    col, (offs, data, validity) = column_of(vals)
    want = orc.distinct_utf8(offs, data, validity)
    res, st, _ = run([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY | T.FLAG_EXACT_KEYS)], [[col]])
    check(res[0], want)
    for _ in range(2):
        st.reset()
        st.update([col])
        check(st.finalize()[0], want)


def test_what_travels_between_states_is_the_keyed_fingerprint(pairs):
    """tgx_merge, blobs and ranks move keys as fingerprints (bytes never leave the device that was fed them): values
    that collide UNDER THE PLAN'S KEY and were fed to DIFFERENT states count once after the union -- stated in
    include/tgx.h at TGX_FLAG_EXACT_KEYS; values fed to one state stay apart through the union."""
    some = pairs[:30]
    a_vals = [p[0] for p in some] + [b"only-a-%d" % i for i in range(100)] + [some[0][1]]  # one pair complete in A
    b_vals = [p[1] for p in some] + [b"only-b-%d" % i for i in range(100)]
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0, flags=T.FLAG_EXACT_KEYS)], fingerprint_key=KEY)
    a, b = T.State(plan), T.State(plan)
    a.update([column_of(a_vals)[0]])
    b.update([column_of(b_vals)[0]])
    assert a.finalize()[0].distinct == len(set(a_vals)) and b.finalize()[0].distinct == len(set(b_vals))
    a.merge([b])
    # 30 values of B meet fingerprints in A: 29 their partners' (a different value: the deviation), one its own
    want = len(set(a_vals) | set(b_vals)) - (len(some) - 1)
    assert a.finalize()[0].distinct == want
    # ... and a blob holds fingerprints only: the pair that A kept apart by its bytes is one record's worth of key there
    u = T.State.deserialize(plan, a.serialize())
    assert u.finalize()[0].distinct == want - 1


# ---- under a key nobody chose: every input counts right on every route, exact or not ----------------------------------
def review_values(rng, n=50_000):
    vals = background(rng, n, 6_000)
    for k, v in enumerate(REVIEW_PAIR * 3):
        vals[1000 + 37 * k] = v
    return vals


@pytest.mark.parametrize("flags", [0, T.FLAG_EXACT_KEYS])
def test_review_pair_counts_twice_on_every_route(flags, monkeypatch):
    """The pair of VERDICT r5 ("What's weak" 1), derived against rounds 1-5's seedless function, under keys drawn from
    the operating system: the table, the lists, Utf8View, a dictionary, tuples, tgx_merge, serialize -> deserialize and
    threaded ranks all give the oracle's counts."""
    from test_gpu_dictionary import encode
    from test_gpu_distributed_sim import _run_ranks
    from test_gpu_utf8view import view_column

    rng = np.random.default_rng(15)
    vals = review_values(rng)
    offs, data, validity = orc.utf8_from_list(vals)
    want = orc.distinct_utf8(offs, data, validity)
    col, _ = column_of(vals)
    sp = [spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY | flags)]
    res, _, _ = run(sp, [[col]], key=None)
    check(res[0], want)
    monkeypatch.setenv("TGX_FP_LISTS_MIN_ROWS", "1000")
    res, _, _ = run(sp, [[col]], key=None)
    check(res[0], want)
    res, _, _ = run(sp, [[view_column(vals, rng, True)]], key=None)
    check(res[0], want)
    res, _, _ = run(sp, [[encode(vals, rng, device=True)]], key=None)
    check(res[0], want)
    # tuples: the pair next to one integer
    n = len(vals)
    ints = np.full(n, 7, np.int64)
    res, _, _ = run([spec(T.DISTINCT, 0, columns=[0, 1], flags=flags)], [[T.Column.int64(to_device(ints), None), col]], key=None)
    assert res[0].distinct == want.distinct + (1 if want.non_null < n else 0)
    # two states: merge, and blobs
    T.init()
    plan = T.Plan(sp)
    cut = 1000 + 37 * 2 + 1  # the first two planted values in A, the rest in B
    parts = [column_of(vals[:cut])[0], column_of(vals[cut:])[0]]
    a, b = T.State(plan), T.State(plan)
    a.update([parts[0]])
    b.update([parts[1]])
    blob_a, blob_b = a.serialize(), b.serialize()
    a.merge([b])
    check(a.finalize()[0], want)
    u = T.State.deserialize(plan, blob_a)
    u.merge([T.State.deserialize(plan, blob_b)])
    check(u.finalize()[0], want)
    # three threaded ranks (one plan: one key)
    cuts = [0, 1000 + 37, 1000 + 37 * 4, n]
    for res, _ in _run_ranks(3, T.Plan(sp), lambda r: [column_of(vals[cuts[r]:cuts[r + 1]])[0]]):
        check(res[0], want)


def test_states_of_different_keys_do_not_unite():
    """a blob carries the key it was made under; a plan with another key refuses it, and says which key to use"""
    vals = [b"a", b"bb", b"ccc"]
    T.init()
    sp = [spec(T.DISTINCT, 0)]
    p1, p2 = T.Plan(sp, fingerprint_key=KEY), T.Plan(sp, fingerprint_key=bytes(16))
    s = T.State(p1)
    s.update([column_of(vals)[0]])
    blob = s.serialize()
    with pytest.raises(T.TgxError, match="fingerprint key"):
        T.State.deserialize(p2, blob)
    assert T.blob_fingerprint_key(blob) == KEY
    p3 = T.Plan(sp, fingerprint_key=T.blob_fingerprint_key(blob))
    assert T.State.deserialize(p3, blob).finalize()[0].distinct == 3
    # the key is fixed once a state exists
    with pytest.raises(T.TgxError, match="fixed once a state"):
        p1.set_fingerprint_key(bytes(16))
    # numeric-only states carry no key: any plan reads them
    n1, n2 = T.Plan([spec(T.DISTINCT, 0)], fingerprint_key=KEY), T.Plan([spec(T.DISTINCT, 0)], fingerprint_key=bytes(16))
    s = T.State(n1)
    s.update([T.Column.int64(to_device(np.arange(100, dtype=np.int64)), None)])
    assert T.State.deserialize(n2, s.serialize()).finalize()[0].distinct == 100


def test_ranks_with_different_keys_all_refuse_and_with_one_key_agree():
    """tgx_allreduce: string keys travel between ranks as fingerprints, which mean the same everywhere only under ONE key
    (tgx_plan_set_fingerprint_key; term_amd.distributed.shared_fingerprint_key broadcasts rank 0's).  Ranks whose plans
    hold different keys all return TGX_INVALID_ARGUMENT from the facts round -- nobody is left in a collective -- and
    the same shards under one key give the oracle's counts."""
    import threading

    import torch

    from term_amd.distributed import ThreadGroup, sharded_suite_step, thread_comm

    rng = np.random.default_rng(21)
    vals = background(rng, 60_000, 9_000)
    offs, data, validity = orc.utf8_from_list(vals)
    want = orc.distinct_utf8(offs, data, validity)
    cuts = [0, 20_032, 41_024, len(vals)]
    sp = [spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY | T.FLAG_EXACT_KEYS), spec(T.COUNT, 0)]
    T.init()

    def run(keys):
        group, out = ThreadGroup(3), [None] * 3

        def worker(rank):
            try:
                torch.cuda.set_device(0)
                plan = T.Plan(sp, fingerprint_key=keys[rank])
                st = T.State(plan)
                comm = thread_comm(group, rank, device_buffers=True)
                shard = [column_of(vals[cuts[rank]:cuts[rank + 1]])[0]]
                out[rank] = ("ok", sharded_suite_step(plan, st, shard, comm))
            except T.TgxError as e:
                out[rank] = ("error", e)
            except Exception as e:  # noqa: BLE001
                out[rank] = ("crash", e)
                group.barrier.abort()

        threads = [threading.Thread(target=worker, args=(r,)) for r in range(3)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in threads), "a rank is stuck"
        return out

    out = run([KEY, KEY, bytes(16)])
    assert [o[0] for o in out] == ["error"] * 3, out
    assert all("fingerprint key" in str(o[1]) for o in out), out
    out = run([KEY, KEY, KEY])
    assert [o[0] for o in out] == ["ok"] * 3, out
    for _, res in out:
        check(res[0], want)
        assert (res[1].total, res[1].non_null) == (want.total, want.non_null)
