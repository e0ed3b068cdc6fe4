"""CPU tests of the two pattern front-ends: the oracle's Pike VM (oracle/regex_oracle.c) and the product's
pattern -> DFA compiler (term_amd/csrc/regex, reached through tgx_regex_validate / tgx_regex_is_match, a
host-side walk of the compiled automaton -- no GPU involved and no data path)."""
import ctypes as C
import json
import os

import pytest

import oracle_binding as orc
import term_amd as T
from term_amd._lib import _Error

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def product_is_match(pattern, value, flags=0):
    err, m = _Error(), C.c_int32()
    pb, vb = pattern.encode(), value.encode()
    rc = T.lib().tgx_regex_is_match(pb, len(pb), flags, vb, len(vb), C.byref(m), C.byref(err))
    if rc != 0:
        raise T.TgxError(rc, err.msg.decode())
    return bool(m.value)


def product_validate(pattern, flags=0):
    err = _Error()
    pb = pattern.encode()
    rc = T.lib().tgx_regex_validate(pb, len(pb), flags, C.byref(err))
    return rc, err.msg.decode()


def pattern_of(case, patterns):
    fmt = case["format"]
    if fmt == "regex":
        return case["pattern"]
    if fmt == "url":
        return patterns["url_localhost" if case.get("allow_localhost") else "url"]
    if fmt in ("phone", "postal_code"):
        return patterns[fmt + "_" + case["country"]]
    return patterns[fmt]


@pytest.fixture(scope="module")
def crosscheck():
    with open(os.path.join(ROOT, "tests", "golden", "regex_crosscheck.json")) as f:
        return json.load(f)


def test_reference_format_vectors_oracle_and_product(golden):
    """constraints/format.rs:917-1508: every ratio the reference's tests assert, through both engines"""
    for case in golden["format"]:
        vals = case["values"]
        if not vals:
            continue
        pat = pattern_of(case, golden["patterns"])
        ci = case.get("case_sensitive") is False
        trim = bool(case.get("trim"))
        niv = case.get("null_is_valid", True)
        offs, data, validity = orc.utf8_from_list(vals)
        out = orc.Regex(pat, ci).count_utf8(offs, data, validity, trim=trim, null_is_valid=niv)
        assert out.total == len(vals)
        assert out.matches / out.total == case["metric"], case["ref"]
        flags = (T.FLAG_TRIM if trim else 0) | (T.FLAG_CASE_INSENSITIVE if ci else 0)
        m = sum((niv if v is None else product_is_match(pat, v, flags)) for v in vals)
        assert m / len(vals) == case["metric"], case["ref"]


def test_every_builtin_pattern_compiles(golden):
    """format.rs:1310-1342 test_all_format_types_have_patterns"""
    for name, pat in golden["patterns"].items():
        assert product_validate(pat)[0] == 0, name
        orc.Regex(pat)


def test_crosscheck_vectors_oracle(crosscheck):
    cache = {}
    for c in crosscheck["cases"]:
        key = (c["pattern"], c["flags"])
        if key not in cache:
            cache[key] = orc.Regex(c["pattern"], bool(c["flags"] & 8))
        assert cache[key].is_match(c["input"]) == c["match"], (c["pattern"], c["input"])


def test_crosscheck_vectors_product(crosscheck):
    for c in crosscheck["cases"]:
        assert product_is_match(c["pattern"], c["input"], c["flags"]) == c["match"], (c["pattern"], c["input"])


def test_invalid_and_rejected_patterns(crosscheck):
    for p in crosscheck["invalid"]:
        rc, msg = product_validate(p)
        assert rc in (1, 2), p  # TGX_INVALID_ARGUMENT ("Invalid regex pattern: ...") or TGX_UNSUPPORTED
        with pytest.raises(ValueError):
            orc.Regex(p)
    # SqlSecurity::validate_regex_pattern (security.rs:152-183, 258-281)
    for p in crosscheck["rejected_by_validate_regex_pattern"]:
        rc, msg = product_validate(p)
        assert rc == 1
        assert "ReDoS" in msg or "too long" in msg
    rc, msg = product_validate("a\0b")
    assert rc == 1 and "null bytes" in msg
    with pytest.raises(T.TgxError) as e:
        T.Plan([T._lib.spec(T.REGEX_MATCH, 0, pattern="(unclosed")])
    assert e.value.status == "TGX_INVALID_ARGUMENT" and "Invalid regex pattern" in str(e.value)


def test_semantic_traps():
    """unanchored search, `$` only at the very end, Unicode \\d, TRIM strips U+0020 only"""
    for engine in (lambda p, v, f=0: product_is_match(p, v, f),
                   lambda p, v, f=0: orc.Regex(p, bool(f & 8)).is_match(v)):
        assert engine(r"\d{3}", "abc123def")            # unanchored
        assert not engine(r"^\d{3}$", "123\n")          # `$` does not match before a trailing newline
        assert engine(r"^\d+$", "٣٤")                   # Unicode decimal digits
        assert not engine(r"^[0-9]+$", "٣٤")
        assert engine(r"^\s$", " ")                # White_Space
        assert engine(r"(?i)^straße$", "STRAẞE")        # simple case folding: ß <-> ẞ only
        assert not engine(r"(?i)^straße$", "STRASSE")
        assert engine(r"^.$", "😀") and not engine(r"^.$", "\n")
        assert engine(r"", "") and engine(r"^$", "") and not engine(r"^$", "x")
    assert product_is_match(r"^x$", "  x  ", T.FLAG_TRIM)
    assert not product_is_match(r"^x$", "\tx", T.FLAG_TRIM)  # tab is not trimmed


def product_group_mask(patterns, flags, value):
    n = len(patterns)
    pbs = [p.encode() for p in patterns]
    arr = (C.c_char_p * n)(*pbs)
    lens = (C.c_size_t * n)(*[len(b) for b in pbs])
    fl = (C.c_uint32 * n)(*flags)
    vb = value.encode()
    mask, grouped, err = C.c_uint32(), C.c_int32(), _Error()
    rc = T.lib().tgx_regex_match_group(arr, lens, fl, n, vb, len(vb), C.byref(mask), C.byref(grouped), C.byref(err))
    if rc != 0:
        raise T.TgxError(rc, err.msg.decode())
    return mask.value, bool(grouped.value)


def test_product_automaton_of_several_patterns_decides_each_of_them(crosscheck, golden):
    """Several pattern checks of one column share ONE walk on the device: the product of their automata
    (regex_compile.cpp dfa_product).  Host-side walk of that product against the PyPI-`regex` truth of the cross-check
    vectors: groups of 2 .. 4 patterns, every input any member of the group has a vector for."""
    T.lib().tgx_regex_match_group.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32),
                                              C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32),
                                              C.POINTER(C.c_int32), C.POINTER(_Error)]
    truth, inputs = {}, {}
    for c in crosscheck["cases"]:
        key = (c["pattern"], c["flags"])
        truth[(key, c["input"])] = c["match"]
        inputs.setdefault(key, []).append(c["input"])
    keys = sorted(inputs)
    checked = grouped_groups = 0
    for g0 in range(0, len(keys) - 1, 3):
        group = keys[g0:g0 + 2 + (g0 // 3) % 3]            # sizes 2, 3, 4 in turn (overlapping windows)
        group = [k for k in group if (k[1] & T.FLAG_TRIM) == (group[0][1] & T.FLAG_TRIM)]
        if len(group) < 2:
            continue
        pats, fl = [k[0] for k in group], [k[1] for k in group]
        pool = sorted({s for k in group for s in inputs[k]})[:: max(1, sum(len(inputs[k]) for k in group) // 24)]
        for s in pool:
            mask, grouped = product_group_mask(pats, fl, s)
            grouped_groups += grouped
            for bit, k in enumerate(group):
                want = truth.get((k, s))
                if want is None:
                    want = product_is_match(k[0], s, k[1])   # no vector for this pair: the single automaton decides
                assert bool((mask >> bit) & 1) == want, (pats, s, bit)
                checked += 1
    assert checked > 1000 and grouped_groups > 0
    # the trio of BASELINE.json configs[2] fits one table
    P = golden["patterns"]
    trio = [r"@", r"^[^@]+@[^@]+\.[^@]+$", P["email"]]
    for s, want in (("user000000001@example001.com", 0b111), ("user#example.com", 0), ("a@b", 0b101), ("a@b.c", 0b111),
                    ("a@@b.c", 0b001), ("", 0)):
        assert product_group_mask(trio, [0, 0, 0], s) == (want, True), s
    with pytest.raises(T.TgxError):
        product_group_mask(["(unclosed", "a"], [0, 0], "a")


@pytest.fixture(scope="module")
def crosscheck_r3():
    with open(os.path.join(ROOT, "tests", "golden", "regex_crosscheck_r3.json")) as f:
        return json.load(f)


def test_multiline_word_boundary_and_set_operation_vectors(crosscheck_r3):
    """(?m) anchors, the ASCII word boundaries and Perl classes of (?-u), class set operations (&& -- ~~): the PyPI
    `regex` truth of tests/golden/regex_crosscheck_r3.json through the oracle's VM, the product's automaton, and the
    product automaton of groups of these patterns (the form in which a column's patterns share one walk)."""
    by_pattern = {}
    for c in crosscheck_r3["cases"]:
        by_pattern.setdefault(c["pattern"], []).append(c)
    for pat, cases in by_pattern.items():
        rx = orc.Regex(pat)
        assert product_validate(pat)[0] == 0, pat
        for c in cases:
            assert rx.is_match(c["input"]) == c["match"], ("oracle", pat, c["input"])
            assert product_is_match(pat, c["input"]) == c["match"], ("product", pat, c["input"])
    T.lib().tgx_regex_match_group.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32),
                                              C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32),
                                              C.POINTER(C.c_int32), C.POINTER(_Error)]
    pats = sorted(by_pattern)
    grouped_any = 0
    for g0 in range(0, len(pats) - 2, 3):
        group = pats[g0:g0 + 3]
        truth = {(c["pattern"], c["input"]): c["match"] for p in group for c in by_pattern[p]}
        for s in sorted({c["input"] for p in group for c in by_pattern[p]})[::7]:
            mask, grouped = product_group_mask(group, [0, 0, 0], s)
            grouped_any += grouped
            for bit, p in enumerate(group):
                want = truth.get((p, s))
                if want is None:  # (no vector for this pair: the oracle decides)
                    want = orc.Regex(p).is_match(s)
                assert bool((mask >> bit) & 1) == want, (group, s, bit)
    assert grouped_any > 0
    # what stays outside the engine is refused, not mis-evaluated
    for p in crosscheck_r3["unsupported"]:
        rc, msg = product_validate(p)
        assert rc == 2, (p, rc, msg)


@pytest.fixture(scope="module")
def crosscheck_r4():
    with open(os.path.join(ROOT, "tests", "golden", "regex_crosscheck_r4.json")) as f:
        return json.load(f)


def test_unicode_word_boundary_vectors(crosscheck_r4):
    """`\\b` / `\\B` as Rust's `regex` takes them by default (security.rs:152-183 accepts whatever that crate compiles):
    the PyPI `regex` truth of tests/golden/regex_crosscheck_r4.json through the oracle's VM (code points on either
    side), the product's automaton (a context that follows a character's bytes, an obligation carried until the
    character behind the assertion is complete) and the product automaton of groups of these patterns."""
    by_pattern = {}
    for c in crosscheck_r4["cases"]:
        by_pattern.setdefault(c["pattern"], []).append(c)
    assert len(by_pattern) >= 25
    for pat, cases in by_pattern.items():
        rx = orc.Regex(pat)
        assert product_validate(pat)[0] == 0, pat
        for c in cases:
            assert rx.is_match(c["input"]) == c["match"], ("oracle", pat, c["input"])
            assert product_is_match(pat, c["input"]) == c["match"], ("product", pat, c["input"])
    pats = sorted(by_pattern)
    grouped_any = 0
    for g0 in range(0, len(pats) - 2, 3):
        group = pats[g0:g0 + 3]
        truth = {(c["pattern"], c["input"]): c["match"] for p in group for c in by_pattern[p]}
        for s in sorted({c["input"] for p in group for c in by_pattern[p]})[::5]:
            mask, grouped = product_group_mask(group, [0, 0, 0], s)
            grouped_any += grouped
            for bit, p in enumerate(group):
                want = truth.get((p, s))
                if want is None:  # (no vector for this pair: the oracle decides)
                    want = orc.Regex(p).is_match(s)
                assert bool((mask >> bit) & 1) == want, (group, s, bit)
    # (an automaton with Unicode word boundaries carries the walk through \w's UTF-8 forms: ~650 states x ~100 byte
    #  classes, four times the LDS table -- these patterns run from a table in global memory and are not grouped)
    assert grouped_any == 0


@pytest.fixture(scope="module")
def crosscheck_r5():
    with open(os.path.join(ROOT, "tests", "golden", "regex_crosscheck_r5.json")) as f:
        return json.load(f)


def test_case_insensitive_vectors_follow_simple_case_folding(crosscheck_r5):
    """`~*` / `(?i)` (format.rs:756-776) is Rust's SIMPLE case folding -- CaseFolding.txt status C + S at Unicode 16.0, no
    Turkic line, no full folding: tests/golden/regex_crosscheck_r5.json (the PyPI `regex` module asked about the
    Turkic i's, the Kelvin / long-s / Angstrom signs, capital sharp s, 15.1's status-S lines and cased pairs of
    Unicode 14 / 16 / 17, corrected where the module is not Rust -- make_regex_crosscheck_r5.py) through the oracle's
    VM and the product's automaton, whose fold tables have different origins (tests/test_unicode_tables.py).  The
    tables both engines shared up to round 4 fail 95 of these vectors."""
    by_pattern = {}
    for c in crosscheck_r5["cases"]:
        by_pattern.setdefault((c["pattern"], c["flags"]), []).append(c)
    assert len(by_pattern) >= 40
    hit = set()
    for (pat, flags), cases in by_pattern.items():
        rx = orc.Regex(pat, bool(flags & T.FLAG_CASE_INSENSITIVE))
        assert product_validate(pat, flags)[0] == 0, pat
        for c in cases:
            assert rx.is_match(c["input"]) == c["match"], ("oracle", pat, flags, c["input"])
            assert product_is_match(pat, c["input"], flags) == c["match"], ("product", pat, flags, c["input"])
            hit.update(c["input"])
    for ch in "ıİKſẞⰯꟁ\U00010597ﬅΐɤ\U00010d70ᲊ꟏\U00016ebb":
        assert ch in hit, hex(ord(ch))
    # the verdict's probes, spelled out
    ci = T.FLAG_CASE_INSENSITIVE
    for engine in (lambda p, v, f=0: product_is_match(p, v, f), lambda p, v, f=0: orc.Regex(p, bool(f & 8)).is_match(v)):
        assert not engine(r"(?i)^i$", "ı") and not engine(r"^[a-z]+$", "ı", ci) and not engine(r"^I$", "ı", ci)
        assert not engine(r"^i$", "İ", ci) and not engine("^İ$", "i", ci) and engine("^İ$", "İ", ci)
        for a, b in (("Ⱟ", "ⱟ"), ("Ꟁ", "ꟁ"), ("\U00010570", "\U00010597"), ("Ɤ", "ɤ"),
                     ("\U00010d50", "\U00010d70"), ("Ᲊ", "ᲊ"), ("k", "K"), ("s", "ſ"), ("ß", "ẞ")):
            assert engine("^%s$" % a, b, ci) and engine("(?i)^%s$" % b, a), (a, b)
            assert not engine("^%s$" % a, b)
        for a, b in (("꟎", "꟏"), ("\U00016ea0", "\U00016ebb")):   # Unicode 17.0: not in regex-syntax 0.8.8
            assert not engine("^%s$" % a, b, ci)
