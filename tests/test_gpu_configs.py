"""-m gpu: BASELINE.json configs[2], [3] and [4] AS CONFIGS (configs[1] is tests/test_gpu_fullsize.py, the headline
1 G x 16 null + range + unique suite is bench.py's own closed-form verification):

  C3  has_pattern / contains_email on a LargeUtf8 column whose value bytes cross 2^31 (the 100 M-row x 28-byte
      column does: 2.8 GB) -- closed forms at full size + an oracle sample cut out of the far end of the buffer;
  C4  the exact plan (completeness + min / max / mean x16, uniqueness x2, KLL(k=200) x4, Pearson x2) vs the oracle at
      2.4 M rows, the same plan row-sharded over 8 simulated ranks through tgx_allreduce (KLL and co-moment states
      travel in the gathered blob), and once at 1 G rows through closed forms and batch-split invariance;
  C5  the exact plan (24 Int64 + 24 Float64 + 16 Dictionary<Int32, Utf8> columns, every column nullable, 144 checks
      in ONE plan) vs the oracle at 1 M rows, row-sharded over 4 simulated ranks, and once at 250 M rows through
      closed forms of the generator.
The oracle finishes the small sizes in seconds; the full sizes are checked through size-independent properties."""
import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from term_amd import synth
from term_amd.csrc_patterns import EMAIL
from test_gpu_distributed_sim import _run_ranks

pytestmark = pytest.mark.gpu

EPS_K200 = 1.65 / 200 ** 0.5  # KllSketch::relative_error_bound (kll_sketch.rs:397-399)


# =============================================================================================== C3
def _email_column(torch, n, device="cuda"):
    """tools/bench_regex.py's column: 'user%09d@example%03d.com' (28 bytes), 4 % with '#' for '@', 1 % NULL"""
    tmpl = torch.tensor(list(b"user000000000@example000.com"), dtype=torch.uint8, device=device)
    L = tmpl.numel()
    data = tmpl.repeat(n).view(n, L)
    rows = torch.arange(n, dtype=torch.int64, device=device)
    v = rows.clone()
    for pos in range(12, 3, -1):
        data[:, pos] = (48 + v % 10).to(torch.uint8)
        v //= 10
    d = rows % 1000
    for pos in range(23, 20, -1):
        data[:, pos] = (48 + d % 10).to(torch.uint8)
        d //= 10
    h = (rows * 2654435761) % 100
    data[h < 4, 13] = ord("#")
    valid = h != 99
    offsets = torch.arange(n + 1, dtype=torch.int64, device=device) * L
    pad = (-n) % 8
    bits = torch.cat([valid, torch.zeros(pad, dtype=torch.bool, device=device)]).view(-1, 8).to(torch.int32)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=device)
    validity = torch.cat([(bits * w).sum(dim=1).to(torch.uint8), torch.zeros(64, dtype=torch.uint8, device=device)])
    flat = torch.cat([data.view(-1), torch.zeros(64, dtype=torch.uint8, device=device)])
    expect = dict(nulls=int((~valid).sum()), with_at=int(((h >= 4) & valid).sum()))
    return offsets, flat, validity, L, expect


def test_c3_patterns_on_a_column_crossing_2_31_value_bytes():
    import torch

    T.init()
    n = 100_000_000  # C3's size; x 28 B = 2.8e9 value bytes > 2^31: int64 offsets, 64-bit byte addressing in the kernel
    offsets, data, validity, L, expect = _email_column(torch, n)
    assert n * L > 2**31
    col = T.Column(T.LARGE_UTF8, n, offsets=offsets, data=data, validity=validity)
    pats = [(r"@", 0), (r"^[^@]+@[^@]+\.[^@]+$", 0), (EMAIL, 0), (EMAIL, T.FLAG_NULL_IS_VALID), (r"#", 0),
            (r"^user\d{9}@example\d{3}\.com$", T.FLAG_TRIM)]
    plan = T.Plan([spec(T.REGEX_MATCH, 0, pattern=p, flags=f) for p, f in pats] +
                  [spec(T.LENGTH, 0, length_min=28, length_max=28), spec(T.COUNT, 0)])
    st = T.State(plan)
    st.update([col])
    res = st.finalize()
    with_at, nulls = expect["with_at"], expect["nulls"]
    assert all(r.total == n for r in res)
    assert [r.matches for r in res[:6]] == [with_at, with_at, with_at, with_at + nulls, n - nulls - with_at, with_at]
    assert res[6].matches == n and res[7].non_null == n - nulls  # LENGTH counts NULL rows; COUNT does not
    # the oracle on a window from the far end of the buffer (rows whose bytes lie beyond 2^31), same pointers sliced
    m, lo = 300_000, n - 300_000 - 64
    assert lo * L > 2**31
    sub = T.Column(T.LARGE_UTF8, m, offsets=offsets, data=data, validity=validity, offset=lo)
    s2 = T.State(plan)
    s2.update([sub])
    r2 = s2.finalize()
    h_off = (offsets[lo: lo + m + 1] - offsets[lo]).to(torch.int32).cpu().numpy()
    h_data = data[lo * L: (lo + m) * L].cpu().numpy()
    bits = validity.cpu().numpy()
    mask = orc.unpack_validity(bits, lo + m)[lo:]
    h_valid = orc.pack_validity(mask)
    for (p, f), r in zip(pats, r2):
        want = orc.Regex(p).count_utf8(h_off, h_data, h_valid, trim=bool(f & T.FLAG_TRIM),
                                       null_is_valid=bool(f & T.FLAG_NULL_IS_VALID))
        assert (r.total, r.matches) == (want.total, want.matches), p
    # additivity at full size: two ragged halves merged == the whole
    cut = 51_234_560
    a, b = T.State(plan), T.State(plan)
    a.update([T.Column(T.LARGE_UTF8, cut, offsets=offsets, data=data, validity=validity)])
    b.update([T.Column(T.LARGE_UTF8, n - cut, offsets=offsets, data=data, validity=validity, offset=cut)])
    a.merge([b])
    assert [(r.total, r.matches, r.non_null) for r in a.finalize()] == [(r.total, r.matches, r.non_null) for r in res]
    del offsets, data, validity
    torch.cuda.empty_cache()


# =============================================================================================== C4
LAYOUT16 = synth.COLUMNS_16
F_COLS = [ci for ci, (k, _) in enumerate(LAYOUT16) if k.startswith("f_")]
KLL_COLS = F_COLS[:4]
PAIRS = list(zip(F_COLS[0:4:2], F_COLS[1:4:2]))


def c4_specs():
    specs = []
    for ci in range(len(LAYOUT16)):
        specs += [spec(T.COUNT, ci), spec(T.NUMERIC_STATS, ci)]
    specs += [spec(T.DISTINCT, ci) for ci in synth.UNIQUE_COLUMNS_16]
    specs += [spec(T.KLL, ci, kll_k=200) for ci in KLL_COLS]
    specs += [spec(T.COMOMENTS, a, column2=b) for a, b in PAIRS]
    return specs


def columns16(table, lo, n):
    return [(T.Column.float64 if k.startswith("f_") else T.Column.int64)(v, b, length=n, offset=lo)
            for (k, _), (v, b) in zip(LAYOUT16, table)]


def _rank_error(sorted_vals, value, phi):
    """distance of `value`'s rank interval in the sorted data from phi"""
    lo = np.searchsorted(sorted_vals, value, side="left") / len(sorted_vals)
    hi = np.searchsorted(sorted_vals, value, side="right") / len(sorted_vals)
    return 0.0 if lo <= phi <= hi else min(abs(lo - phi), abs(hi - phi))


def check_c4_against_oracle(specs, res, state, host, n):
    """host: [(values ndarray, validity ndarray or None)] of the same rows"""
    by = {}
    for si, (s, r) in enumerate(zip(specs, res)):
        by[(s.kind, s.column, s.column2)] = (si, r)
    for ci, (kind, _) in enumerate(LAYOUT16):
        v, b = host[ci]
        o = orc.stats(v, b, n=n)
        c, r = by[(T.COUNT, ci, -1)][1], by[(T.NUMERIC_STATS, ci, -1)][1]
        assert (c.total, c.non_null, r.total, r.non_null) == (n, o.non_null, n, o.non_null), ci
        if kind.startswith("f_"):
            assert (r.min_f, r.max_f) == (o.min_f, o.max_f)
            assert abs(r.sum_f - o.sum_hi) <= 1e-6 * abs(o.sum_hi) and abs(r.mean - o.sum_hi / o.non_null) <= 1e-6 * abs(o.mean)
        else:
            assert (r.min_i, r.max_i, r.sum_i) == (o.min_i, o.max_i, o.sum_i_wrapping)
    for ci in synth.UNIQUE_COLUMNS_16:
        v, b = host[ci]
        o = orc.distinct_bits64(v.view(np.uint64), b, n=n)
        r = by[(T.DISTINCT, ci, -1)][1]
        assert (r.total, r.non_null, r.distinct) == (o.total, o.non_null, o.distinct), ci
    for ci in KLL_COLS:
        v, b = host[ci]
        mask = np.ones(n, bool) if b is None else orc.unpack_validity(b, n)
        kept = np.sort(v[:n][mask])
        si, r = by[(T.KLL, ci, -1)]
        assert r.kll_n == len(kept)  # total weight == non-NULL rows, exactly
        summ = state.kll_summary(si)
        assert (summ["n"], summ["min"], summ["max"]) == (len(kept), kept[0], kept[-1])
        for phi in (0.5, 0.95, 0.99):
            q = state.kll_quantile(si, phi)
            assert _rank_error(kept, q, phi) < EPS_K200, (ci, phi, q)
        assert state.kll_quantile(si, 0.0) == kept[0] and state.kll_quantile(si, 1.0) == kept[-1]
    for a, c in PAIRS:
        o = orc.comoments(host[a][0], host[c][0], host[a][1], host[c][1], n=n)
        r = by[(T.COMOMENTS, a, c)][1]
        assert (r.total, r.non_null) == (n, o.n)
        for got, want in ((r.sum_x, o.sum_x), (r.sum_y, o.sum_y), (r.sum_x2, o.sum_x2), (r.sum_y2, o.sum_y2),
                          (r.sum_xy, o.sum_xy)):
            assert abs(got - want) <= 1e-9 * max(abs(want), 1.0)


def _host_table(table, n):
    return [(np.ascontiguousarray(v[:n].cpu().numpy()), None if b is None else b[: n // 8 + 64].cpu().numpy())
            for v, b in table]


def test_c4_plan_vs_oracle_and_over_8_ranks():
    n = 2_400_000 + 64 * 3
    T.init(distinct_capacity_hint=n)
    table = synth.make_table(LAYOUT16, 0, n, n, 0x7E570004, "cuda")
    host = _host_table(table, n)
    specs = c4_specs()
    plan = T.Plan(specs)
    one = T.State(plan)
    one.update(columns16(table, 0, n))
    check_c4_against_oracle(specs, one.finalize(), one, host, n)
    # streamed in three ragged batches: the same verdicts (KLL: same weight, same error bound)
    three = T.State(plan)
    for lo, hi in ((0, 800_000), (800_000, 800_064), (800_064, n)):
        three.update(columns16(table, lo, hi - lo))
    check_c4_against_oracle(specs, three.finalize(), three, host, n)
    # row-sharded over 8 ranks through tgx_allreduce: KLL sketches and co-moments travel in the gathered blob
    world = 8

    def shards_of(rank):
        from term_amd.distributed import shard_rows

        lo, hi = shard_rows(n, world, rank)
        return columns16(table, lo, hi - lo)

    results = _run_ranks(world, plan, shards_of, steps=2)
    for res, st in results:
        check_c4_against_oracle(specs, res, st, host, n)
    # rank-ordered merge: every rank holds the same bits
    blobs = {bytes(st.serialize()) for _, st in results}
    assert len(blobs) == 1


def test_c4_plan_with_its_spearman_pair_vs_oracle_and_over_8_ranks():
    """SURVEY.md 8d lists a Spearman pair in C4 (timed separately in the bench): the whole plan of C4 PLUS the rank
    correlation of its first pair in ONE plan -- one state, and row shards over 8 ranks, where the ranking is a
    distributed sort inside tgx_allreduce; the rank sums bit-exact against the oracle (UInt64 wrapping as the reference)"""
    n = 1_200_000 + 64
    T.init(distinct_capacity_hint=n)
    table = synth.make_table(LAYOUT16, 0, n, n, 0x7E570004, "cuda")
    host = _host_table(table, n)
    specs = c4_specs() + [spec(T.SPEARMAN, PAIRS[0][0], column2=PAIRS[0][1])]
    plan = T.Plan(specs)
    (a, c) = PAIRS[0]
    want = orc.spearman_state(host[a][0], host[c][0], host[a][1], host[c][1], n=n)

    def check(res, st):
        check_c4_against_oracle(specs[:-1], res[:-1], st, host, n)
        r = res[-1]
        assert (r.total, r.non_null) == (n, want.n)
        assert (r.sum_x, r.sum_y, r.sum_x2, r.sum_y2, r.sum_xy) == (want.sum_x, want.sum_y, want.sum_x2, want.sum_y2, want.sum_xy)

    one = T.State(plan)
    one.update(columns16(table, 0, n))
    check(one.finalize(), one)
    world = 8

    def shards_of(rank):
        from term_amd.distributed import shard_rows

        lo, hi = shard_rows(n, world, rank)
        return columns16(table, lo, hi - lo)

    for res, st in _run_ranks(world, plan, shards_of):
        check(res, st)


def test_c4_full_size_properties():
    import torch

    n = 1_000_000_000 // 64 * 64
    T.init(distinct_capacity_hint=n)
    table = synth.make_table(LAYOUT16, 0, n, n, 0x7E570004, "cuda")
    specs = c4_specs()
    plan = T.Plan(specs)
    st = T.State(plan)
    st.update(columns16(table, 0, n))
    res = st.finalize()
    by = {(s.kind, s.column, s.column2): (si, r) for si, (s, r) in enumerate(zip(specs, res))}
    # closed forms of the generator
    assert by[(T.DISTINCT, 0, -1)][1].distinct == n
    s0 = by[(T.NUMERIC_STATS, 0, -1)][1]
    assert (s0.min_i, s0.max_i, s0.sum_i) == (0, n - 1, n * (n - 1) // 2)
    assert 0.99 * (n // 10) < by[(T.DISTINCT, 1, -1)][1].distinct <= n // 10
    for ci, (kind, has_validity) in enumerate(LAYOUT16):
        c = by[(T.COUNT, ci, -1)][1]
        assert c.total == n and (c.non_null == n if not has_validity else abs(c.non_null / n - 0.95) < 1e-4)
    for ci in KLL_COLS:
        si, r = by[(T.KLL, ci, -1)]
        assert r.kll_n == by[(T.COUNT, ci, -1)][1].non_null
        stt = by[(T.NUMERIC_STATS, ci, -1)][1]
        summ = st.kll_summary(si)
        assert (summ["min"], summ["max"]) == (stt.min_f, stt.max_f)
        qs = [st.kll_quantile(si, q) for q in (0.5, 0.95, 0.99)]
        assert qs[0] <= qs[1] <= qs[2]
        if LAYOUT16[ci][0] == "f_uniform":  # uniform [0, 1000): the quantile IS its rank
            for q, v in zip((0.5, 0.95, 0.99), qs):
                assert abs(v / 1000.0 - q) < EPS_K200
        if LAYOUT16[ci][0] == "f_normal":
            assert abs(qs[0]) < 0.05 and abs(qs[1] - 1.6449) < 0.1 and abs(qs[2] - 2.3263) < 0.15
    for a, c in PAIRS:  # independent columns: n = rows with both valid, Sxy ~ Sx Sy / n
        r = by[(T.COMOMENTS, a, c)][1]
        assert abs(r.non_null / n - 0.95 * 0.95) < 1e-4
        sx, sy = by[(T.NUMERIC_STATS, a, -1)][1], by[(T.NUMERIC_STATS, c, -1)][1]
        corr = (r.non_null * r.sum_xy - r.sum_x * r.sum_y) / np.sqrt(
            (r.non_null * r.sum_x2 - r.sum_x ** 2) * (r.non_null * r.sum_y2 - r.sum_y ** 2))
        assert abs(corr) < 1e-3
        assert abs(r.sum_x / r.non_null - sx.mean) < 1e-3 * max(1.0, abs(sx.mean))
        assert abs(r.sum_y / r.non_null - sy.mean) < 1e-2
    # batch-split invariance at full size: four ragged batches give the same integers, the same float aggregates to
    # 1e-12 and KLL sketches of the same weight within the same bound
    st4 = T.State(plan)
    cuts = [0, 250_000_064, 333_333_312, 900_000_000, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        st4.update(columns16(table, lo, hi - lo))
    res4 = st4.finalize()
    for (s, a), b in zip(zip(specs, res), res4):
        assert (a.total, a.non_null, a.min_i, a.max_i, a.sum_i, a.distinct, a.kll_n) == \
            (b.total, b.non_null, b.min_i, b.max_i, b.sum_i, b.distinct, b.kll_n), s.kind
        assert (a.min_f, a.max_f) == (b.min_f, b.max_f)
        for x, y in ((a.sum_f, b.sum_f), (a.sum_xy, b.sum_xy), (a.sum_x2, b.sum_x2)):
            assert abs(x - y) <= 1e-12 * max(1.0, abs(x))
    for ci in KLL_COLS:
        si = by[(T.KLL, ci, -1)][0]
        if LAYOUT16[ci][0] == "f_uniform":
            for q in (0.5, 0.95, 0.99):
                assert abs(st4.kll_quantile(si, q) / 1000.0 - q) < EPS_K200
    del table, st, st4
    torch.cuda.empty_cache()


# =============================================================================================== C5
C5_NUMERIC = [("i_wide", True)] * 24 + [("f_uniform", True)] * 12 + [("f_normal", True)] * 12
C5_CARDS = [1000, 1000, 4096, 10_000, 10_000, 50_000, 100_000, 100_000, 250_000, 250_000, 500_000, 500_000,
            1_000_000, 1_000_000, 1_000_000, 1_000_000]
C5_SEED = 0x7E570005


def _c5_dictionary(card):
    """`card` strings, 1 in 16 of them not an e-mail address (tools/bench_configs.py run_c5)"""
    entries = [("user%07d@example%03d.com" % (e, e % 997)) if e % 16 else ("not-an-email-%d" % e) for e in range(card)]
    data = np.frombuffer("".join(entries).encode(), dtype=np.uint8)
    offs = np.zeros(card + 1, dtype=np.int32)
    offs[1:] = np.cumsum([len(e) for e in entries])
    return offs, data


def build_c5(torch, n, row0=0, n_total=None, cards=C5_CARDS):
    """columns (device), specs, and per dictionary column (indices tensor, validity tensor, card)"""
    n_total = n if n_total is None else n_total
    table = synth.make_table(C5_NUMERIC, row0, n, n_total, C5_SEED, "cuda")
    columns = [(T.Column.float64 if k.startswith("f_") else T.Column.int64)(v, b, length=n)
               for (k, _), (v, b) in zip(C5_NUMERIC, table)]
    dicts = []
    for k, card in enumerate(cards):
        ci = len(C5_NUMERIC) + k
        offs, data = _c5_dictionary(card)
        dcol = T.Column.utf8(torch.from_numpy(offs).cuda(),
                             torch.cat([torch.from_numpy(data.copy()), torch.zeros(64, dtype=torch.uint8)]).cuda())
        idx = synth.gen_column("k_mod10", ci, row0, n, 10 * card, C5_SEED, "cuda").to(torch.int32)  # uniform in [0, card)
        validity = synth.gen_validity(ci, row0, n, C5_SEED, "cuda")
        columns.append(T.Column.dict32_utf8(idx, dcol, validity=validity, length=n))
        dicts.append((idx, validity, card, offs, data))
    specs = [spec(T.COUNT, ci) for ci in range(len(columns))]
    specs += [spec(T.NUMERIC_STATS, ci) for ci in range(len(C5_NUMERIC))]
    for k in range(len(cards)):
        ci = len(C5_NUMERIC) + k
        specs += [spec(T.DISTINCT, ci), spec(T.REGEX_MATCH, ci, pattern=EMAIL, flags=T.FLAG_NULL_IS_VALID)]
    return table, columns, specs, dicts


def check_c5_against_oracle(specs, res, table, dicts, n, dict_cols=None):
    by = {(s.kind, s.column): r for s, r in zip(specs, res)}
    for ci, (kind, _) in enumerate(C5_NUMERIC):
        v = np.ascontiguousarray(table[ci][0][:n].cpu().numpy())
        b = table[ci][1][: n // 8 + 64].cpu().numpy()
        o = orc.stats(v, b, n=n)
        c, r = by[(T.COUNT, ci)], by[(T.NUMERIC_STATS, ci)]
        assert (c.total, c.non_null, r.non_null) == (n, o.non_null, o.non_null)
        if kind.startswith("f_"):
            assert (r.min_f, r.max_f) == (o.min_f, o.max_f) and abs(r.sum_f - o.sum_hi) <= 1e-6 * abs(o.sum_hi)
        else:
            assert (r.min_i, r.max_i, r.sum_i) == (o.min_i, o.max_i, o.sum_i_wrapping)
    rx = orc.Regex(EMAIL)
    for k, (idx, validity, card, offs, data) in enumerate(dicts):
        if dict_cols is not None and k not in dict_cols:
            continue
        ci = len(C5_NUMERIC) + k
        # the decoded column, as Utf8, through the oracle
        h_idx = idx[:n].cpu().numpy().astype(np.int64)
        h_valid = validity[: n // 8 + 64].cpu().numpy()
        lens = (offs[1:] - offs[:-1]).astype(np.int64)
        row_len = lens[h_idx]
        d_off = np.zeros(n + 1, dtype=np.int64)
        d_off[1:] = np.cumsum(row_len)
        assert d_off[-1] < 2**31
        starts = offs[:-1].astype(np.int64)[h_idx]
        flat = np.repeat(starts - d_off[:-1], row_len) + np.arange(d_off[-1], dtype=np.int64)
        d_data = data[flat]
        d_off32 = d_off.astype(np.int32)
        od = orc.distinct_utf8(d_off32, d_data, h_valid, n=n)
        om = rx.count_utf8(d_off32, d_data, h_valid, n=n, null_is_valid=True)
        got = (by[(T.COUNT, ci)].total, by[(T.COUNT, ci)].non_null, by[(T.DISTINCT, ci)].distinct,
               by[(T.REGEX_MATCH, ci)].total, by[(T.REGEX_MATCH, ci)].matches)
        assert got == (n, od.non_null, od.distinct, om.total, om.matches), (k, card)


def test_c5_plan_vs_oracle_and_over_4_ranks():
    import torch

    T.init(distinct_capacity_hint=1 << 20)
    n = 1_000_000 + 64
    table, columns, specs, dicts = build_c5(torch, n)
    assert len(columns) == 64 and len(specs) == 144
    plan = T.Plan(specs)
    st = T.State(plan)
    st.update(columns)
    res = st.finalize()
    check_c5_against_oracle(specs, res, table, dicts, n, dict_cols=(0, 2, 5, 12))
    # row shards over 4 ranks: every rank brings its own window of the indices and the same dictionaries
    world = 4

    def shards_of(rank):
        from term_amd.distributed import shard_rows

        lo, hi = shard_rows(n, world, rank)
        out = []
        for c in columns:
            out.append(c.sliced(lo, hi - lo))
        return out

    for got, _ in _run_ranks(world, plan, shards_of):
        for s, a, b in zip(specs, got, res):
            assert (a.total, a.non_null, a.min_i, a.max_i, a.sum_i, a.distinct, a.matches) == \
                (b.total, b.non_null, b.min_i, b.max_i, b.sum_i, b.distinct, b.matches), (s.kind, s.column)
            assert (a.min_f, a.max_f) == (b.min_f, b.max_f) and abs(a.sum_f - b.sum_f) <= 1e-12 * max(1.0, abs(b.sum_f))


def test_c5_full_size_properties():
    import torch

    T.init(distinct_capacity_hint=1 << 20)
    n = 250_000_000 // 64 * 64
    table, columns, specs, dicts = build_c5(torch, n)
    plan = T.Plan(specs)
    st = T.State(plan)
    st.update(columns)
    res = st.finalize()
    by = {(s.kind, s.column): r for s, r in zip(specs, res)}
    for ci in range(len(columns)):
        c = by[(T.COUNT, ci)]
        assert c.total == n and abs(c.non_null / n - 0.95) < 2e-4
    for ci, (kind, _) in enumerate(C5_NUMERIC):
        r = by[(T.NUMERIC_STATS, ci)]
        assert r.non_null == by[(T.COUNT, ci)].non_null
        if kind == "i_wide":
            assert -(2**40) <= r.min_i < -(2**40) + 2**20 and 2**40 - 2**20 < r.max_i < 2**40
        elif kind == "f_uniform":
            assert 0.0 <= r.min_f < 1e-3 and 999.999 < r.max_f < 1000.0 and abs(r.mean - 500.0) < 0.1
        else:
            assert abs(r.mean) < 1e-3
    for k, (idx, validity, card, _, _) in enumerate(dicts):
        ci = len(C5_NUMERIC) + k
        # 250 M uniform draws from <= 1 M entries: every entry is referenced (P[miss] < 1e-100); an entry is an
        # address unless e % 16 == 0 -- closed forms of the generator, checked against the device's own counts
        assert by[(T.DISTINCT, ci)].distinct == card
        valid_rows = by[(T.COUNT, ci)].non_null
        m = by[(T.REGEX_MATCH, ci)]
        nulls = n - valid_rows
        bad_entries = (card + 15) // 16
        assert m.total == n
        assert abs((m.matches - nulls) / valid_rows - (1 - bad_entries / card)) < 5e-4
    # the dictionary columns again, streamed in three ragged batches with the numeric columns: same integers
    st3 = T.State(plan)
    cuts = [0, 100_000_064, 100_000_128, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        st3.update([c.sliced(lo, hi - lo) for c in columns])
    for s, a, b in zip(specs, st3.finalize(), res):
        assert (a.total, a.non_null, a.min_i, a.max_i, a.sum_i, a.distinct, a.matches) == \
            (b.total, b.non_null, b.min_i, b.max_i, b.sum_i, b.distinct, b.matches), (s.kind, s.column)
    del table, columns, dicts, st, st3
    torch.cuda.empty_cache()
