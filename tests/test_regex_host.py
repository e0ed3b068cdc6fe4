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
