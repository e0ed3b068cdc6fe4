"""-m gpu: COUNT(DISTINCT) / value-count checks on Utf8 columns (128-bit fingerprint sets) vs the oracle."""
import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import run_plan
from test_gpu_regex import utf8_column

pytestmark = pytest.mark.gpu


def check(res, d):
    assert (res.total, res.non_null, res.distinct, res.groups_once) == (d.total, d.non_null, d.distinct, d.groups_once)


def test_reference_uniqueness_vectors(golden):
    """constraints/uniqueness.rs:907-1070 (all on Utf8 columns) through the HIP path"""
    for case in golden["uniqueness"]:
        vals = case["values"]
        if not vals:
            continue
        offs, data, validity = orc.utf8_from_list(vals)
        res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)],
                             [[utf8_column(offs, data, validity, False)]])
        r, n = res[0], len(vals)
        kind = case["kind"]
        if kind in ("full_uniqueness", "distinctness"):
            assert r.distinct / n == case["metric"], case["ref"]
        elif kind == "unique_value_ratio":
            assert r.groups_once / n == case["metric"], case["ref"]
        elif kind == "unique_with_nulls_include":
            assert (r.distinct + (1 if r.non_null < n else 0)) / n == case["metric"], case["ref"]
        elif kind == "primary_key":
            nulls = n - r.non_null
            if "NULL" in case.get("message_contains", ""):
                assert nulls > 0
            elif "duplicate" in case.get("message_contains", ""):
                assert nulls == 0 and r.distinct != n
            else:
                assert nulls == 0 and r.distinct == n
    a = golden["analyzers"]["table"]["name"]
    offs, data, validity = orc.utf8_from_list(a)
    res, _, _ = run_plan([spec(T.DISTINCT, 0)], [[utf8_column(offs, data, validity, False)]])
    assert (res[0].non_null, res[0].distinct) == (4, 3)  # analyzers/basic/tests.rs:116-128


def make_strings(rng, n, card):
    keys = rng.integers(0, card, size=n)
    vals = []
    for i, k in enumerate(keys):
        r = k % 7
        if r == 0:
            vals.append("k%d" % k)
        elif r == 1:
            vals.append("user-%d@example.com" % k)
        elif r == 2:
            vals.append("x" * (k % 50) + str(k))
        elif r == 3:
            vals.append("Ünï-%d-ß" % k)
        elif r == 4:
            vals.append("" if k % 11 == 4 else "e%d" % k)
        elif r == 5:
            vals.append(None if i % 3 == 0 else "n%d" % k)
        else:
            vals.append("a much longer value that spans several eight byte words %d" % k)
    return vals


@pytest.mark.parametrize("n,card", [(1000, 50), (200_000, 30_000), (300_000, 10**9)])
def test_seeded_strings(n, card):
    rng = np.random.default_rng(n + card % 1000)
    vals = make_strings(rng, n, card)
    offs, data, validity = orc.utf8_from_list(vals)
    want = orc.distinct_utf8(offs, data, validity)
    for large in (False, True):
        res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.COUNT, 0)],
                             [[utf8_column(offs, data, validity, True, large=large)]])
        check(res[0], want)
        assert res[1].non_null == want.non_null


def test_same_value_at_every_alignment_hashes_equal():
    """the fingerprint reads aligned 8-byte words; it must not depend on where a value sits in memory"""
    vals = []
    for pad in range(9):
        vals.append("p" * pad)              # shifts the alignment of what follows
        vals.append("the-same-value-0123456789")
    offs, data, validity = orc.utf8_from_list(vals)
    res, _, _ = run_plan([spec(T.DISTINCT, 0)], [[utf8_column(offs, data, validity, True)]])
    assert res[0].distinct == orc.distinct_utf8(offs, data, validity).distinct == 10


def test_batches_merge_serialize_and_exchange():
    import torch

    rng = np.random.default_rng(3)
    n = 120_000
    vals = make_strings(rng, n, 40_000)
    offs, data, validity = orc.utf8_from_list(vals)
    want = orc.distinct_utf8(offs, data, validity)
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)])
    cuts = [0, 30_000, 30_001, 90_000, n]
    states = [T.State(plan), T.State(plan)]
    for i, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
        states[i % 2].update([utf8_column(offs, data, validity, True, offset=lo, length=hi - lo)])
    # (1) exact union through serialize -> deserialize -> merge
    a = T.State.deserialize(plan, states[0].serialize())
    a.merge([T.State.deserialize(plan, states[1].serialize())])
    check(a.finalize()[0], want)
    # (2) hash-owner exchange between the two "ranks", then count-only merge
    assert states[0].distinct_record_bytes(0) == 32
    world, exported = 2, []
    for st in states:
        ptr, counts = st.distinct_export(0, world)
        total = sum(counts)

        class P:
            __cuda_array_interface__ = {"shape": (max(total, 1) * 32,), "typestr": "|u1", "data": (ptr, False),
                                        "version": 2}

        recs = torch.as_tensor(P(), device="cuda")[: total * 32].clone()
        exported.append((recs, counts))
    for r in range(world):
        parts = []
        for recs, counts in exported:
            start = sum(counts[:r])
            parts.append(recs[32 * start: 32 * (start + counts[r])])
        mine = torch.cat(parts).contiguous()
        torch.cuda.synchronize()
        states[r].distinct_import(0, mine.data_ptr(), mine.numel() // 32)
    states[0].merge([states[1]])
    check(states[0].finalize()[0], want)


def test_host_slices_stage_only_their_window(golden):
    """HOST columns viewed at a large Arrow offset: the staged window (not the buffers from their start) must give
    the results of the same rows on the device -- numeric, Utf8, LargeUtf8 and pattern checks"""
    from gpu_util import numeric_column

    rng = np.random.default_rng(99)
    n = 300_000
    vals = make_strings(rng, n, 20_000)
    offs, data, validity = orc.utf8_from_list(vals)
    ints = rng.integers(-1000, 1000, size=n, dtype=np.int64)
    mask = rng.random(n) >= 0.1
    iv = orc.pack_validity(mask)
    for lo, m in ((0, 1000), (64, 70_000), (123_457, 100_001), (n - 77, 77)):
        specs = [spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.REGEX_MATCH, 0, pattern=r"^user-\d+@"),
                 spec(T.NUMERIC_STATS, 1), spec(T.DISTINCT, 1), spec(T.COUNT, 0)]
        for large in (False, True):
            got = run_plan(specs, [[utf8_column(offs, data, validity, False, offset=lo, length=m, large=large),
                                    numeric_column(ints, iv, False, offset=lo, length=m)]])[0]
            want = run_plan(specs, [[utf8_column(offs, data, validity, True, offset=lo, length=m, large=large),
                                     numeric_column(ints, iv, True, offset=lo, length=m)]])[0]
            for g, w in zip(got, want):
                assert (g.total, g.non_null, g.distinct, g.groups_once, g.matches, g.min_i, g.max_i, g.sum_i) == \
                       (w.total, w.non_null, w.distinct, w.groups_once, w.matches, w.min_i, w.max_i, w.sum_i)
            d = orc.distinct_utf8(offs, data, validity, n=m, offset=lo)
            check(got[0], d)


def test_distinct_over_column_tuples():
    """COUNT(DISTINCT (a, b, ...)) / GROUP BY a, b, ...: the tuple is one value, NULL components included
    (constraints/uniqueness.rs:557-562, 687-699, 709-715); checked against a Python Counter of the rows"""
    from collections import Counter

    from gpu_util import numeric_column

    rng = np.random.default_rng(21)
    n = 150_000
    strs = make_strings(rng, n, 300)
    ints = rng.integers(0, 40, size=n, dtype=np.int64)
    flts = rng.integers(0, 5, size=n).astype(np.float64)
    flts[rng.random(n) < 0.1] = -0.0  # by bit pattern: -0.0 and 0.0 are different components
    imask, fmask = rng.random(n) >= 0.1, rng.random(n) >= 0.2
    offs, data, svalid = orc.utf8_from_list(strs)
    iv, fv = orc.pack_validity(imask), orc.pack_validity(fmask)

    def rows(cols):
        out = []
        for i in range(n):
            t = []
            for c in cols:
                if c == 0:
                    t.append(strs[i])
                elif c == 1:
                    t.append(int(ints[i]) if imask[i] else None)
                else:
                    t.append(float(flts[i]).hex() if fmask[i] else None)
            out.append(tuple(t))
        return out

    for large in (False, True):
        columns = [utf8_column(offs, data, svalid, True, large=large), numeric_column(ints, iv, True),
                   numeric_column(flts, fv, True)]
        for cols in ([0, 1], [1, 2], [2, 1, 0]):
            cnt = Counter(rows(cols))
            res, plan, st = run_plan([spec(T.DISTINCT, cols[0], columns=cols, flags=T.FLAG_MULTIPLICITY),
                                      spec(T.DISTINCT, cols[0], columns=cols)], [columns])
            want = (n, sum(1 for t in rows(cols) if None not in t), len(cnt), sum(1 for v in cnt.values() if v == 1))
            assert (res[0].total, res[0].non_null, res[0].distinct, res[0].groups_once) == want
            assert (res[1].total, res[1].non_null, res[1].distinct, res[1].groups_once) == want[:3] + (0,)
    # batches + serialize / merge: the union is by tuple VALUE
    plan = T.Plan([spec(T.DISTINCT, 0, columns=[0, 1], flags=T.FLAG_MULTIPLICITY)])
    a, b = T.State(plan), T.State(plan)
    half = n // 2 // 64 * 64
    a.update([utf8_column(offs, data, svalid, True, length=half), numeric_column(ints, iv, True, length=half)])
    b.update([utf8_column(offs, data, svalid, True, offset=half, length=n - half),
              numeric_column(ints, iv, True, offset=half, length=n - half)])
    m = T.State.deserialize(plan, a.serialize())
    m.merge([T.State.deserialize(plan, b.serialize())])
    cnt = Counter(rows([0, 1]))
    r = m.finalize()[0]
    assert (r.total, r.distinct, r.groups_once) == (n, len(cnt), sum(1 for v in cnt.values() if v == 1))
    # too many columns / a column list on another check kind
    with pytest.raises(T.TgxError):
        T.Plan([spec(T.DISTINCT, 0, columns=list(range(9)))])
    with pytest.raises(T.TgxError):
        T.Plan([spec(T.COUNT, 0, columns=[0, 1])])


@pytest.mark.parametrize("layout", ["utf8", "large", "view", "dict"])
def test_length_check_counts_characters(layout):
    """LENGTH(col) bounds in CHARACTERS, NULL rows always satisfied (constraints/length.rs:36-45, 167-171)"""
    from test_gpu_dictionary import encode
    from test_gpu_utf8view import view_column

    rng = np.random.default_rng(17)
    n = 120_000
    alphabet = ["a", "Z", "é", "ß", "你", "🦀", " ", "0"]
    vals = []
    for i in range(n):
        r = rng.random()
        if r < 0.05:
            vals.append(None)
        else:
            k = int(rng.integers(0, 40)) if r < 0.9 else int(rng.integers(0, 600))
            vals.append("".join(alphabet[int(x)] for x in rng.integers(0, len(alphabet), size=k)))
    if layout == "dict":
        vals = [None if v is None else v[:30] for v in vals[:40_000]]
        n = len(vals)
    bounds = [(0, None), (1, None), (5, None), (0, 10), (3, 10), (5, 5), (0, 0), (100, 400), (13, 13)]
    specs = [spec(T.LENGTH, 0, length_min=lo, length_max=hi) for lo, hi in bounds]
    if layout == "view":
        col = view_column(vals, rng, True)
    elif layout == "dict":
        col = encode(vals, rng)
    else:
        offs, data, validity = orc.utf8_from_list(vals)
        col = utf8_column(offs, data, validity, True, large=(layout == "large"))
    res, _, _ = run_plan(specs, [[col]])
    o_offs, o_data, o_validity = orc.utf8_from_list(vals)
    for (lo, hi), r in zip(bounds, res):
        want = orc.length_count_utf8(o_offs, o_data, o_validity, min_chars=lo, max_chars=hi).matches
        assert want == sum(1 for v in vals if v is None or (len(v) >= lo and (hi is None or len(v) <= hi)))
        assert (r.total, r.matches) == (n, want), (layout, lo, hi)
    with pytest.raises(T.TgxError):
        T.Plan([spec(T.LENGTH, 0, length_min=5, length_max=4)])
