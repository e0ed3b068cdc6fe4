"""The oracle against implementations that are NOT this repo's: Arrow C++ compute kernels (pyarrow 25: min_max, sum,
mean, variance, count_distinct, value_counts, utf8_length), RE2 (pyarrow's match_substring_regex -- the engine whose
syntax and leftmost semantics Rust's regex crate follows; TG/constraints/format.rs:756-776 builds its patterns for
that family), NumPy and SciPy (corrcoef, cov, spearmanr).  None of them is the reference (DataFusion is Rust; it is not
in this image), but they are independent restatements of the same SQL aggregates, and where the reference's own
literal vectors are small they are what stands between the oracle and a shared mistake: the
round-4 verdict's "pin strength".  Seeded inputs; edge cases the two sides are known to treat alike (NULLs, empty
input, -0.0 / 0.0 as two values, one NaN payload as one value, multi-byte characters) are in, the ones they are known to
differ on are left out and named where they are left out."""
import math

import numpy as np
import pytest

import oracle_binding as orc

pa = pytest.importorskip("pyarrow")
pc = pytest.importorskip("pyarrow.compute")


def _arrow(values, mask):
    return pa.array(values, mask=None if mask is None else ~mask)


def _validity(mask):
    return None if mask is None else orc.pack_validity(mask)


@pytest.mark.parametrize("n", [0, 1, 2, 1000, 100_003])
@pytest.mark.parametrize("null_rate", [0.0, 0.3, 1.0])
def test_int64_aggregates_against_arrow_compute(n, null_rate):
    rng = np.random.default_rng([n, int(null_rate * 10)])
    v = rng.integers(-2**40, 2**40, size=n, dtype=np.int64)
    mask = None if null_rate == 0.0 else rng.random(n) >= null_rate
    a = _arrow(v, mask)
    s = orc.stats(v, _validity(mask), n)
    live = n if mask is None else int(mask.sum())
    assert (s.total, s.non_null) == (n, live) == (len(a), len(a) - a.null_count)
    if live == 0:
        assert not s.has_value
        return
    mm = pc.min_max(a).as_py()
    assert (s.min_i, s.max_i) == (mm["min"], mm["max"])
    assert s.sum_i_wrapping == pc.sum(a).as_py()
    assert s.mean == pytest.approx(pc.mean(a).as_py(), rel=1e-12)
    if live >= 2:
        assert s.has_variance
        assert s.var_samp == pytest.approx(pc.variance(a, ddof=1).as_py(), rel=1e-9)
        assert s.stddev_samp == pytest.approx(pc.stddev(a, ddof=1).as_py(), rel=1e-9)


@pytest.mark.parametrize("n", [1, 7, 1000, 100_003])
@pytest.mark.parametrize("null_rate", [0.0, 0.25])
def test_float64_aggregates_against_arrow_compute(n, null_rate):
    rng = np.random.default_rng([n, 5, int(null_rate * 100)])
    v = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 6, size=n)
    v[rng.random(n) < 0.05] = 0.0
    v[rng.random(n) < 0.05] = -0.0
    mask = None if null_rate == 0.0 else rng.random(n) >= null_rate
    a = _arrow(v, mask)
    s = orc.stats(v, _validity(mask), n)
    live = len(a) - a.null_count
    assert (s.total, s.non_null) == (n, live)
    if live == 0:
        return
    mm = pc.min_max(a).as_py()
    # (the sign of a zero extreme is not compared: SQL MIN over {0.0, -0.0} may name either)
    assert (s.min_f, s.max_f) == (mm["min"], mm["max"])
    assert s.sum_f == pytest.approx(pc.sum(a).as_py(), rel=1e-11, abs=1e-6)
    assert s.mean == pytest.approx(pc.mean(a).as_py(), rel=1e-11, abs=1e-9)
    if live >= 2:
        assert s.var_samp == pytest.approx(pc.variance(a, ddof=1).as_py(), rel=1e-9)


def _once(a):
    counts = pc.value_counts(a)
    return sum(1 for c, v in zip(counts.field("counts").to_pylist(), counts.field("values").to_pylist()) if c == 1 and v is not None)


@pytest.mark.parametrize("card", [1, 50, 5000, 10**9])
def test_count_distinct_int64_against_arrow_compute(card):
    rng = np.random.default_rng(card % 1000 + 3)
    n = 60_000
    v = rng.integers(-card, card, size=n, dtype=np.int64)
    v[:3] = [-1, np.iinfo(np.int64).min, np.iinfo(np.int64).max]
    mask = rng.random(n) >= 0.1
    a = _arrow(v, mask)
    d = orc.distinct_bits64(v.view(np.uint64), _validity(mask), n)
    assert d.non_null == len(a) - a.null_count
    assert d.distinct == pc.count_distinct(a, mode="only_valid").as_py()
    assert d.groups_once == _once(a)


def test_count_distinct_float64_zeros_and_nan_against_arrow_compute():
    """-0.0 and 0.0 are two values, NaN (one payload) is one: Arrow C++ and the oracle's bit patterns agree there.
    (NaNs of DIFFERENT payloads are left out: Arrow C++ folds them, DataFusion's hash of the bits does not.)"""
    rng = np.random.default_rng(17)
    n = 40_000
    pool = np.concatenate([rng.standard_normal(3000), [0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324]])
    v = pool[rng.integers(0, len(pool), size=n)]
    mask = rng.random(n) >= 0.05
    a = _arrow(v, mask)
    d = orc.distinct_bits64(v.view(np.uint64), _validity(mask), n)
    assert d.distinct == pc.count_distinct(a, mode="only_valid").as_py()
    assert d.groups_once == _once(a)


_WORDS = ["", "a", "ab", "straße", "STRASSE", "héllo", "Ωmega", "日本語", "naïve", "x" * 70, "😀", "á", "İstanbul", " pad "]


def _strings(rng, n, null_rate=0.1):
    vals = [_WORDS[int(i)] + (str(int(k)) if k % 3 == 0 else "") for i, k in
            zip(rng.integers(0, len(_WORDS), size=n), rng.integers(0, 40, size=n))]
    return [None if rng.random() < null_rate else v for v in vals]


def test_count_distinct_and_lengths_of_strings_against_arrow_compute():
    rng = np.random.default_rng(23)
    vals = _strings(rng, 30_000)
    a = pa.array(vals, pa.large_string())
    offsets, data, validity = orc.utf8_from_list(vals)
    d = orc.distinct_utf8(offsets, data, validity, len(vals))
    assert d.distinct == pc.count_distinct(a, mode="only_valid").as_py()
    assert d.groups_once == _once(a)
    # LENGTH counts characters (code points), as character_length / Arrow's utf8_length do -- not bytes, not graphemes;
    # a NULL row counts as passing (TG/constraints/length.rs:169: `{condition} OR {column} IS NULL`)
    lengths = pc.utf8_length(a).to_pylist()
    for lo, hi in [(0, None), (1, None), (0, 0), (2, 6), (7, 7), (70, 80)]:
        want = sum(1 for k in lengths if k is None or (k >= lo and (hi is None or k <= hi)))
        got = orc.length_count_utf8(offsets, data, validity, len(vals), min_chars=lo, max_chars=hi)
        assert (got.total, got.matches) == (len(vals), want), (lo, hi)


# patterns inside what RE2 and Rust's regex agree on: no \d \w \s \b over non-ASCII text (Rust's are Unicode-aware, RE2's
# ASCII), no look-around or back-references (neither has them), Unicode classes and (?i) only over characters that
# have been in Unicode since 6.0 (the two engines ship different Unicode versions)
_PATTERNS_ANY_TEXT = [
    r"@", r"^a", r"b$", r"^$", r"^.+$", r"é", r"ß", r"^[^@]+@[^@]+\.[a-z]{2,}$", r"(ab)+", r"a|日本", r"x{3,5}", r"x{70}",
    r"\p{L}+[0-9]", r"^\p{Lu}", r"\p{Greek}", r"\p{Han}{2}", r"[α-ω]", r"[^\x00-\x7F]", r"(?i)strasse", r"(?i)^héLLO",
    r"(?i)ωMEGA", r"^(?:a|ab)(?:c|bcd)?$", r"a.c", r"a\.c", r"[[:alpha:]]+@", r"^\PL*$",
]
_PATTERNS_ASCII_TEXT = [r"\d+", r"^\w+@\w+\.\w+$", r"\s", r"\bab\b", r"^\S+$", r"(?i)[a-c]\d", r"^[\w.+-]+@[\w-]+(\.[\w-]+)+$"]


def _check_patterns(patterns, vals):
    a = pa.array(vals, pa.large_string())
    for pat in patterns:
        want = pc.match_substring_regex(a, pat).to_pylist()
        rx = orc.Regex(pat)
        got = [None if v is None else rx.is_match(v) for v in vals]
        assert got == want, pat


def test_patterns_against_re2_over_any_text():
    rng = np.random.default_rng(31)
    base = ["a@b.co", "ab", "abab", "abc", "a.c", "abcd", "日本語", "Ωmega", "ωmega", "straße", "STRASSE", "Strasse", "héllo",
            "HÉLLO", "x" * 4, "x" * 70, "", "😀", "a1", "é9", "A", "user@example.com", "user@@example.com", "no at", "\n", "a\nb"]
    vals = base + [v for v in _strings(rng, 400, 0.05)]
    _check_patterns(_PATTERNS_ANY_TEXT, vals)


def test_perl_classes_against_re2_over_ascii_text():
    rng = np.random.default_rng(37)
    alphabet = list("ab1_ @.-+\t") + ["ab", "x9"]
    vals = ["".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), size=int(k))) for k in rng.integers(0, 12, size=600)]
    vals += ["john.doe+tag@mail-host.example.org", "ab", " ab ", "xab", "a_b@c_d.e_f", None]
    _check_patterns(_PATTERNS_ASCII_TEXT + _PATTERNS_ANY_TEXT[:12], vals)


def test_correlations_against_numpy_and_scipy():
    stats = pytest.importorskip("scipy.stats")
    rng = np.random.default_rng(41)
    n = 20_000
    x = rng.standard_normal(n)
    y = 0.6 * x + 0.8 * rng.standard_normal(n)
    xi = rng.permutation(n).astype(np.int64)  # (no ties: SQL RANK() and SciPy's average ranks are the same numbers)
    c = orc.comoments(x, y)
    assert orc.pearson(c) == pytest.approx(np.corrcoef(x, y)[0, 1], rel=1e-10)
    assert orc.covariance(c) == pytest.approx(np.cov(x, y, ddof=1)[0, 1], rel=1e-10)
    online = orc.corr_online(x, y)
    assert online.corr == pytest.approx(np.corrcoef(x, y)[0, 1], rel=1e-10)
    assert online.covar_samp == pytest.approx(np.cov(x, y, ddof=1)[0, 1], rel=1e-10)
    s = orc.spearman_state(xi, y)
    assert orc.pearson(s) == pytest.approx(stats.spearmanr(xi, y).statistic, rel=1e-9)
    # NULL in either column drops the pair on both sides
    xm, ym = rng.random(n) >= 0.1, rng.random(n) >= 0.1
    both = xm & ym
    c = orc.comoments(x, y, orc.pack_validity(xm), orc.pack_validity(ym))
    assert c.n == int(both.sum())
    assert orc.pearson(c) == pytest.approx(np.corrcoef(x[both], y[both])[0, 1], rel=1e-10)
