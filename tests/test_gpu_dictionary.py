"""-m gpu: checks on Dictionary<Int32, Utf8> columns vs the oracle run on the DECODED column.

A dictionary batch must give exactly the results of the equivalent plain Utf8 batch: COUNT from the index validity,
pattern checks evaluated once per dictionary entry and gathered, COUNT(DISTINCT) / value counts by VALUE (per-batch
dictionaries need not agree, may hold unused and repeated entries)."""
import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import pad_validity, run_plan, to_device
from test_gpu_regex import utf8_column
from test_gpu_strings import check, make_strings

pytestmark = pytest.mark.gpu

EMAIL = r"^[A-Za-z0-9._%+-]+@[A-Za-z0-9.-]+\.[A-Za-z]{2,}$"


def encode(vals, rng, extra_entries=(), repeat_entries=False, large=False, device=True, offset=0, length=None):
    """list[str|None] -> (dict Column, decoded list) with a shuffled dictionary; `extra_entries` are never
    referenced; `repeat_entries` stores some values twice so two indices mean the same string."""
    uniq = sorted({v for v in vals if v is not None})
    entries = list(uniq) + list(extra_entries)
    if repeat_entries:
        entries += uniq[: max(1, len(uniq) // 3)]
    order = rng.permutation(len(entries))
    entries = [entries[i] for i in order]
    where = {}
    for j, e in enumerate(entries):
        where.setdefault(e, []).append(j)
    idx = np.zeros(len(vals), np.int32)
    validity = np.zeros((len(vals) + 7) // 8, np.uint8)
    any_null = False
    for i, v in enumerate(vals):
        if v is None:
            idx[i] = rng.integers(0, max(1, len(entries)))  # garbage under a NULL slot is never interpreted
            any_null = True
        else:
            choices = where[v]
            idx[i] = choices[int(rng.integers(0, len(choices)))]
            validity[i >> 3] |= 1 << (i & 7)
    d_offs, d_data, _ = orc.utf8_from_list(entries if entries else [])
    dcol = utf8_column(d_offs, d_data, None, device, large=large)
    v = pad_validity(validity) if any_null else None
    if device:
        idx_b, v_b = to_device(idx), (to_device(v) if v is not None else None)
    else:
        idx_b, v_b = idx, v
    n = (len(vals) - offset) if length is None else length
    return T.Column.dict32_utf8(idx_b, dcol, validity=v_b, length=n, offset=offset)


def oracle_of(vals, pattern=None, flags=0):
    offs, data, validity = orc.utf8_from_list(vals)
    d = orc.distinct_utf8(offs, data, validity)
    m = None
    if pattern is not None:
        rx = orc.Regex(pattern, case_insensitive=bool(flags & T.FLAG_CASE_INSENSITIVE))
        m = rx.count_utf8(offs, data, validity, trim=bool(flags & T.FLAG_TRIM),
                          null_is_valid=bool(flags & T.FLAG_NULL_IS_VALID)).matches
    return d, m


@pytest.mark.parametrize("n,card,large,device", [(1000, 40, False, True), (250_000, 300, False, True),
                                                  (250_000, 60_000, True, True), (50_000, 2000, False, False)])
def test_dictionary_equals_plain(n, card, large, device):
    rng = np.random.default_rng(n + card)
    vals = make_strings(rng, n, card)
    for flags in (0, T.FLAG_TRIM | T.FLAG_NULL_IS_VALID):
        want_d, want_m = oracle_of(vals, EMAIL, flags)
        col = encode(vals, rng, extra_entries=["never-used@example.com", "unused"], repeat_entries=True, large=large,
                     device=device)
        res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.COUNT, 0),
                              spec(T.REGEX_MATCH, 0, flags=flags, pattern=EMAIL)], [[col]])
        check(res[0], want_d)
        assert (res[1].total, res[1].non_null) == (n, want_d.non_null)
        assert (res[2].total, res[2].matches) == (n, want_m)


def test_per_batch_dictionaries_and_slices():
    """every batch brings its own dictionary (different order, different entries); batches are slices with a
    non-zero Arrow offset; the union is by VALUE"""
    rng = np.random.default_rng(11)
    n = 90_000
    vals = make_strings(rng, n, 5000)
    want_d, want_m = oracle_of(vals, r"^user-\d+@", 0)
    cuts = [0, 1, 20_001, 20_001, 64_000, n]
    batches = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        lead = int(rng.integers(0, 70))  # rows before the slice, never read
        part = [None] * lead + vals[lo:hi]
        batches.append([encode(part, rng, offset=lead, length=hi - lo)])
    res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.REGEX_MATCH, 0, pattern=r"^user-\d+@")],
                         batches)
    check(res[0], want_d)
    assert (res[1].total, res[1].matches) == (n, want_m)


def test_edge_cases():
    rng = np.random.default_rng(5)
    # all rows NULL over an EMPTY dictionary
    col = encode([None] * 100, rng)
    res, _, _ = run_plan([spec(T.DISTINCT, 0), spec(T.COUNT, 0),
                          spec(T.REGEX_MATCH, 0, pattern="a", flags=T.FLAG_NULL_IS_VALID)], [[col]])
    assert (res[0].total, res[0].non_null, res[0].distinct) == (100, 0, 0)
    assert (res[1].total, res[1].non_null) == (100, 0)
    assert (res[2].total, res[2].matches) == (100, 100)
    # one-entry dictionary, 100k references: a single very hot entry
    vals = ["only"] * 100_000
    res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.REGEX_MATCH, 0, pattern="^on")],
                         [[encode(vals, rng)]])
    assert (res[0].distinct, res[0].groups_once, res[0].non_null, res[1].matches) == (1, 0, 100_000, 100_000)
    # empty batch
    res, _, _ = run_plan([spec(T.DISTINCT, 0), spec(T.REGEX_MATCH, 0, pattern="x")], [[encode([], rng)]])
    assert (res[0].total, res[0].distinct, res[1].total, res[1].matches) == (0, 0, 0, 0)
    # the empty string is a value like any other; the single NULL row is a group of one (GROUP BY keeps NULL)
    vals = ["", "", "a", None]
    res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.REGEX_MATCH, 0, pattern="^$")],
                         [[encode(vals, rng)]])
    assert (res[0].distinct, res[0].groups_once, res[0].non_null, res[1].matches) == (2, 2, 3, 2)


def test_from_arrow_dictionary():
    pa = pytest.importorskip("pyarrow")
    rng = np.random.default_rng(2)
    vals = make_strings(rng, 20_000, 700)
    arr = pa.array(vals, type=pa.string()).dictionary_encode()
    assert pa.types.is_int32(arr.type.index_type)
    want_d, want_m = oracle_of(vals, EMAIL, 0)
    sl = arr.slice(123, 15_000)
    want_s, _ = oracle_of(vals[123:15_123])
    res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.REGEX_MATCH, 0, pattern=EMAIL)],
                         [[T.Column.from_arrow(arr)]])
    check(res[0], want_d)
    assert res[1].matches == want_m
    res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)], [[T.Column.from_arrow(sl)]])
    check(res[0], want_s)


@pytest.mark.parametrize("mult", [False, True])
def test_dictionary_beyond_the_lds_budget(mult):
    """1.2 M entries: the per-entry bitmaps no longer fit LDS (1 Mi verdict bits, 512 Ki with multiplicity), so the
    rows go through the test-before-set global path; a NULL dictionary VALUE makes its rows NULL rows"""
    import torch

    rng = np.random.default_rng(31 + mult)
    card, n = 1_200_000, 2_000_000
    ids = np.arange(card)
    width = 7
    digits = ((ids[:, None] // 10 ** np.arange(width - 1, -1, -1)) % 10 + ord("0")).astype(np.uint8)
    data = np.concatenate([np.full((card, 1), ord("k"), np.uint8), digits], axis=1).reshape(-1)  # "k0000123"
    offs = (np.arange(card + 1) * (width + 1)).astype(np.int32)
    dvalid = np.ones(card, dtype=bool)
    dvalid[[5, 77, 1_100_000]] = False  # NULL dictionary values
    idx = rng.integers(0, card // 2, size=n).astype(np.int32)  # half of the entries are never referenced
    idx[:10] = [5, 5, 77, 1_100_000, 3, 3, 4, 1_199_999, 1_199_999, 1_199_998]
    mask = rng.random(n) >= 0.05
    mask[:10] = True
    dcol = T.Column.utf8(to_device(offs), to_device(np.concatenate([data, np.zeros(16, np.uint8)])),
                         validity=to_device(pad_validity(orc.pack_validity(dvalid))), length=card)
    col = T.Column.dict32_utf8(to_device(idx), dcol, validity=to_device(pad_validity(orc.pack_validity(mask))), length=n)
    flags = T.FLAG_MULTIPLICITY if mult else 0
    res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=flags), spec(T.REGEX_MATCH, 0, pattern=r"7$"),
                          spec(T.REGEX_MATCH, 0, pattern=r"7$", flags=T.FLAG_NULL_IS_VALID), spec(T.COUNT, 0)], [[col]])
    live = mask & dvalid[idx]                      # rows that carry a value
    used, counts = np.unique(idx[live], return_counts=True)
    assert (res[0].total, res[0].non_null, res[0].distinct) == (n, int(live.sum()), len(used))
    if mult:
        assert res[0].groups_once == int((counts == 1).sum()) + (1 if n - int(live.sum()) == 1 else 0)
    ends7 = (idx % 10 == 7) & live
    assert res[1].matches == int(ends7.sum())
    assert res[2].matches == int(ends7.sum()) + int((~live).sum())
    assert (res[3].total, res[3].non_null) == (n, int(live.sum()))  # COUNT: Arrow's logical nulls
    res, _, _ = run_plan([spec(T.COUNT, 0)], [[col]])  # COUNT alone (no DISTINCT to lean on)
    assert (res[0].total, res[0].non_null) == (n, int(live.sum()))


@pytest.mark.parametrize("mult", [False, True])
@pytest.mark.parametrize("n,card", [(70_000, 50), (300_000, 40_000)])
def test_many_checks_ride_on_the_distinct_pass(n, card, mult):
    """A dictionary column with a DISTINCT check and SIX pattern / length checks: up to four gathers are fused into the
    usage pass (kernels/dict.hip: dict_fused_kernel), the rest run on their own -- every count as on the plain column,
    over two batches with different dictionaries."""
    rng = np.random.default_rng(n * 3 + card + int(mult))
    vals = make_strings(rng, n, card)
    pats = [(EMAIL, 0), (r"@", T.FLAG_NULL_IS_VALID), (r"^[a-z]", T.FLAG_CASE_INSENSITIVE), (r"\d\d", T.FLAG_TRIM),
            (r"com$", T.FLAG_TRIM | T.FLAG_NULL_IS_VALID)]
    specs = [spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY if mult else 0)]
    specs += [spec(T.REGEX_MATCH, 0, flags=f, pattern=p) for p, f in pats]
    specs += [spec(T.LENGTH, 0, length_min=3, length_max=20), spec(T.COUNT, 0)]
    half = n // 2
    batches = [[encode(vals[:half], rng, extra_entries=["zz@zz.com"], repeat_entries=True)],
               [encode(vals[half:], rng, repeat_entries=False)]]
    res, _, _ = run_plan(specs, batches)
    offs, data, validity = orc.utf8_from_list(vals)
    want_d = orc.distinct_utf8(offs, data, validity)
    assert res[0].distinct == want_d.distinct and res[0].non_null == want_d.non_null
    if mult:
        check(res[0], want_d)
    for r, (p, f) in zip(res[1:], pats):
        rx = orc.Regex(p, case_insensitive=bool(f & T.FLAG_CASE_INSENSITIVE))
        want = rx.count_utf8(offs, data, validity, trim=bool(f & T.FLAG_TRIM),
                             null_is_valid=bool(f & T.FLAG_NULL_IS_VALID)).matches
        assert (r.total, r.matches) == (n, want), p
    want_len = orc.length_count_utf8(offs, data, validity, min_chars=3, max_chars=20).matches
    assert (res[6].total, res[6].matches) == (n, want_len)
    assert (res[7].total, res[7].non_null) == (n, want_d.non_null)
