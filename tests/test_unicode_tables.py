"""The Unicode tables of the two pattern engines, held against each other and against further sources.

    product   term_amd/csrc/regex/unicode_tables.h   tools/gen_unicode_tables.py: the PyPI `regex` module's database
                                                      (Unicode 17.0) probed, 17.0's additions and the Turkic pairs taken out
    oracle    oracle/unicode_oracle_tables.h         oracle/gen_unicode_oracle.py: ICU 70 (Unicode 14.0) + the written-down
                                                      CaseFolding lines of 15.1 / 16.0 (+ the `regex` module for the
                                                      classes of the code points 15.0 - 16.0 added)

The target is Rust's regex-syntax 0.8.8 (Unicode 16.0; `(?i)` = simple case folding, CaseFolding.txt status C + S:
/root/reference/term-guard/src/constraints/format.rs:756-760 makes `case_sensitive = false` the operator `~*`).  The
round-4 verdict found both engines folding i with dotless i from ONE shared table; now the tables have two origins and
this file is where they meet: the fold tables must be the same set, and equal to a THIRD reconstruction made here from
CPython's `unicodedata` (Unicode 13.0) + the C / S lines 14.0, 15.1 and 16.0 added."""
import ctypes
import os
import re
import sys
import unicodedata

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import unicode_versions as UV  # noqa: E402

MAXCP = 0x110000
PRODUCT = os.path.join(ROOT, "term_amd", "csrc", "regex", "unicode_tables.h")
ORACLE = os.path.join(ROOT, "oracle", "unicode_oracle_tables.h")

# CaseFolding-14.0.0.txt's additions to 13.0 (all status C): Glagolitic, Latin Extended-D, Vithkuqi
CASEFOLDING_14_0 = ([(0x2C2F, 0x2C5F), (0xA7C0, 0xA7C1), (0xA7D0, 0xA7D1), (0xA7D6, 0xA7D7), (0xA7D8, 0xA7D9)] +
                    [(cp, cp + 39) for cp in range(0x10570, 0x10596) if cp not in (0x1057B, 0x1058B, 0x10593)])


def load_tables(path):
    text = open(path).read()
    arrays = {}
    for m in re.finditer(r"static const tgx_urange (tgx_ur_\w+)\[\] = \{(.*?)\};", text, re.S):
        pairs = re.findall(r"\{0x([0-9A-Fa-f]+),0x([0-9A-Fa-f]+)\}", m.group(2))
        arrays[m.group(1)] = [(int(a, 16), int(b, 16)) for a, b in pairs]
    tables = {}
    for m in re.finditer(r'\{"(\w+)",\s*(tgx_ur_\w+),\s*(\d+)\}', text):
        name, arr, count = m.group(1), m.group(2), int(m.group(3))
        assert len(arrays[arr]) == count, name
        tables[name] = arrays[arr]
    return tables


def load_fold_pairs(path):
    text = open(path).read()
    m = re.search(r"tgx_fold_pairs\[\]\[2\] = \{(.*?)\};", text, re.S)
    pairs = [(int(a, 16), int(b, 16)) for a, b in re.findall(r"\{0x([0-9A-Fa-f]+),0x([0-9A-Fa-f]+)\}", m.group(1))]
    n = int(re.search(r"tgx_n_fold_pairs = (\d+);", text).group(1))
    assert n == len(pairs) and pairs == sorted(pairs)
    return set(pairs)


def membership(ranges):
    out = np.zeros(MAXCP, dtype=bool)
    for lo, hi in ranges:
        assert 0 <= lo <= hi < MAXCP
        out[lo:hi + 1] = True
    return out


def pairs_of_folds(fold):
    """{code point: its fold} -> every ordered pair of each class of code points with one fold"""
    classes = {}
    for cp, f in fold.items():
        assert f not in fold, hex(f)
        classes.setdefault(f, {f}).add(cp)
    return {(a, b) for members in classes.values() for a in members for b in members if a != b}


def cpython_simple_folds():
    """CaseFolding-13.0.0's C + S lines out of CPython: `casefold()` is the full folding (C + F) -- where it yields one
    character that is the C line; where it yields several (an F line) the S line, if there is one, is the character's
    simple lower-case mapping (U+1E9E -> U+00DF, the Greek capitals with prosgegrammeni); the Turkic lines (T) are in
    neither.  U+0130's lower() is two characters: no simple fold, as in CaseFolding.txt."""
    assert unicodedata.unidata_version == "13.0.0"
    fold = {}
    for cp in range(MAXCP):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        c = chr(cp)
        f = c.casefold()
        if len(f) != 1:
            f = c.lower()
        if len(f) == 1 and f != c:
            fold[cp] = ord(f)
    return fold


def test_fold_tables_agree_and_are_simple_case_folding_of_unicode_16():
    product, oracle = load_fold_pairs(PRODUCT), load_fold_pairs(ORACLE)
    assert product == oracle, sorted(product ^ oracle)[:8]  # (two origins: `regex` module probed / ICU 70 + written lines)
    fold = cpython_simple_folds()
    for code, mapping in CASEFOLDING_14_0:
        assert code not in fold
        fold[code] = mapping
    for code, status, mapping in UV.CASEFOLDING_15_1 + UV.CASEFOLDING_16_0:
        assert status in "CS" and code not in fold
        fold[code] = mapping
    want = pairs_of_folds(fold)
    assert product == want, (sorted(product - want)[:8], sorted(want - product)[:8])
    # the defects the verdict named, as facts about the table
    for a, b in ((0x49, 0x131), (0x69, 0x131), (0x69, 0x130), (0x49, 0x130)):   # no Turkic line
        assert (a, b) not in product and (b, a) not in product
    assert not any(0x130 in p or 0x131 in p for p in product)
    for a, b in ((0x4B, 0x212A), (0x6B, 0x212A), (0x73, 0x17F), (0x53, 0x17F), (0xDF, 0x1E9E), (0xE5, 0x212B),
                 (0x2C2F, 0x2C5F), (0xA7C0, 0xA7C1), (0x10570, 0x10597),             # 14.0
                 (0xFB05, 0xFB06), (0x390, 0x1FD3), (0x3B0, 0x1FE3),                 # 15.1 (status S)
                 (0xA7CB, 0x264), (0x10D50, 0x10D70), (0x1C89, 0x1C8A), (0xA7DC, 0x19B)):  # 16.0
        assert (a, b) in product and (b, a) in product, (hex(a), hex(b))
    for a, b in ((0xA7CE, 0xA7CF), (0xA7D2, 0xA7D3), (0xA7D4, 0xA7D5), (0x16EA0, 0x16EBB)):  # 17.0: not regex-syntax 0.8.8's
        assert (a, b) not in product
    assert (0xDF, 0x73) not in product and (0x73, 0xDF) not in product  # (no full folding: sharp s is not "ss")


def icu70():
    try:
        lib = ctypes.CDLL("libicuuc.so.70")
    except OSError:
        return None
    return lib


def test_fold_table_against_icu_and_the_count_of_17_additions():
    """Where ICU 70 can be loaded (this image has it): its simple folding + the later lines IS the table, and the code
    points the `regex`-derived product table calls assigned but ICU does not are exactly 15.0 + 15.1 + 16.0's additions
    -- the arithmetic that pins tools/unicode_versions.py's list of what 17.0 added."""
    lib = icu70()
    if lib is None:
        pytest.skip("no libicuuc.so.70")
    fold_fn = lib.u_foldCase_70
    fold_fn.restype, fold_fn.argtypes = ctypes.c_int32, [ctypes.c_int32, ctypes.c_uint32]
    ctype = lib.u_charType_70
    ctype.restype, ctype.argtypes = ctypes.c_int8, [ctypes.c_int32]
    fold = {}
    icu_unassigned = np.zeros(MAXCP, dtype=bool)
    for cp in range(MAXCP):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        f = fold_fn(cp, 0)
        if f != cp:
            fold[cp] = f
        icu_unassigned[cp] = ctype(cp) == 0
    for code, status, mapping in UV.CASEFOLDING_15_1 + UV.CASEFOLDING_16_0:
        fold[code] = mapping
    assert load_fold_pairs(PRODUCT) == pairs_of_folds(fold)
    product = load_tables(PRODUCT)
    assigned_product = ~membership(product["gc_Cn"])
    late = assigned_product & icu_unassigned
    assert int(late.sum()) == 4489 + 627 + 5185  # Unicode 15.0, 15.1, 16.0
    added17 = np.zeros(MAXCP, dtype=bool)
    for lo, hi in UV.UNICODE_17_ADDITIONS:
        added17[lo:hi + 1] = True
    assert int(added17.sum()) == UV.UNICODE_17_COUNT == 4803
    assert not (added17 & assigned_product).any() and (added17 & icu_unassigned).sum() == 4803


def test_class_tables_of_the_two_origins():
    product, oracle = load_tables(PRODUCT), load_tables(ORACLE)
    assert set(product) == set(oracle)
    # what the pattern engines use without being asked: \d \s \w and the word boundary's classifier -- identical
    for name in ("perl_digit", "perl_space", "perl_word", "gc_Nd", "gc_Pc", "gc_M", "gc_L", "gc_N", "gc_P", "gc_S", "gc_Z",
                 "gc_C", "gc_Cn", "prop_White_Space", "prop_Uppercase"):
        assert product[name] == oracle[name], name
    for name in product:
        if name.startswith("script_"):
            assert product[name] == oracle[name], name
    # older characters whose properties changed between Unicode 14 (oracle) and 17 (product): which of the two 16.0 has
    # cannot be told here -- 40-odd code points, none of them in \w \d \s
    changed = {}
    for name in product:
        diff = np.flatnonzero(membership(product[name]) != membership(oracle[name]))
        if len(diff):
            changed[name] = set(int(c) for c in diff)
    alpha = set(range(0x363, 0x370)) | {0xC04, 0xF82, 0xF83} | set(range(0x1DD3, 0x1DE7)) | {0x11080, 0x11081}
    assert changed == {"gc_Ll": {0x295}, "gc_Lo": {0x295}, "gc_Mc": {0x1171E}, "gc_Mn": {0x1171E},
                       "prop_Alphabetic": alpha,  # (combining marks: in \w on either side, as M)
                       "prop_Lowercase": {0x295, 0x10FC, 0xA7F2, 0xA7F3, 0xA7F4, 0xAB69}}, changed


def test_product_tables_agree_with_unicodedata_where_both_know_the_character():
    """The product's classes against CPython's `unicodedata` (Unicode 13.0) on the code points both know: a generator
    error (an off-by-one in a range, a dropped block, a swapped table) moves thousands, a Unicode version a handful."""
    product = load_tables(PRODUCT)
    cat = np.array([unicodedata.category(chr(cp)) for cp in range(MAXCP)])
    assigned_py = cat != "Cn"
    assigned_tbl = ~membership(product["gc_Cn"])
    both = assigned_py & assigned_tbl & (cat != "Cs")  # (surrogates are no scalar values: the engine's classes leave them out)
    assert both.sum() > 140_000  # Unicode 13 assigns 143 859 characters (+ surrogates / private use)
    # ranges are sorted, disjoint and non-adjacent-overlapping in every table
    for name, ranges in product.items():
        for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
            assert a1 < b0, name
    budget = 64  # characters whose General_Category moved between Unicode 13 and the table's version
    moved = 0
    for name, ranges in product.items():
        if not name.startswith("gc_") or name == "gc_Cn":
            continue
        short = name[3:]
        want = (cat == short) if len(short) == 2 else np.char.startswith(cat, short)
        diff = (membership(ranges) != want) & both
        moved += int(diff.sum())
        assert diff.sum() <= budget, (name, [hex(c) for c in np.flatnonzero(diff)[:10]])
    assert moved <= 4 * budget
    # the Perl classes of Rust's `regex`: \d = Nd, \s = White_Space, \w = Alphabetic + M + Nd + Pc + Join_Control
    nd = membership(product["gc_Nd"])
    assert np.array_equal(membership(product["perl_digit"]), nd)
    space = membership(product["perl_space"])
    assert np.array_equal(space, membership(product["prop_White_Space"]))
    for cp in (0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x20, 0x85, 0xA0, 0x1680, 0x2000, 0x200A, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000):
        assert space[cp], hex(cp)
    assert space.sum() == 25
    word = membership(product["perl_word"])
    expect_word = (membership(product["prop_Alphabetic"]) | membership(product["gc_M"]) | nd | membership(product["gc_Pc"]))
    expect_word[[0x200C, 0x200D]] = True
    assert np.array_equal(word, expect_word)
    # derived properties against the second source on the characters both know: Lowercase / Uppercase contain Ll / Lu
    # (U+0295 is Ll up to Unicode 14 and Lo in 17: the same handful of moved characters as above)
    assert ((cat == "Ll") & both & ~membership(product["prop_Lowercase"])).sum() <= 8
    assert ((cat == "Lu") & both & ~membership(product["prop_Uppercase"])).sum() <= 8
    # scripts: spot checks of blocks that have not moved since Unicode 1
    for name, lo, hi in (("script_Greek", 0x3B1, 0x3C9), ("script_Cyrillic", 0x410, 0x44F), ("script_Hebrew", 0x5D0, 0x5EA),
                         ("script_Hiragana", 0x3041, 0x3096), ("script_Katakana", 0x30A1, 0x30FA), ("script_Han", 0x4E00, 0x9FA5),
                         ("script_Latin", 0x61, 0x7A), ("script_Arabic", 0x621, 0x63A)):
        assert membership(product[name])[lo:hi + 1].all(), name
    # 17.0's additions are unassigned for both engines
    for lo, hi in UV.UNICODE_17_ADDITIONS:
        assert not assigned_tbl[lo:hi + 1].any(), hex(lo)
