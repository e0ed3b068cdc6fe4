"""The Unicode class tables of the pattern engine (term_amd/csrc/regex/unicode_tables.h, generated from the PyPI `regex`
module's database; oracle/unicode_tables.h is the same file) against a SECOND source: CPython's `unicodedata`
(Unicode 13.0).  The two databases are different Unicode versions, so only code points that are ASSIGNED in both are
compared, and a handful of characters whose category changed between the versions is allowed -- a generator error
(an off-by-one in a range, a dropped block, a swapped table) moves thousands."""
import os
import re
import unicodedata

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAXCP = 0x110000


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


def membership(ranges):
    out = np.zeros(MAXCP, dtype=bool)
    for lo, hi in ranges:
        assert 0 <= lo <= hi < MAXCP
        out[lo:hi + 1] = True
    return out


def test_tables_agree_with_unicodedata_where_both_know_the_character():
    product = load_tables(os.path.join(ROOT, "term_amd", "csrc", "regex", "unicode_tables.h"))
    oracle = load_tables(os.path.join(ROOT, "oracle", "unicode_tables.h"))
    assert product == oracle  # (one generated file, kept in both trees)
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
    # (U+0295 moved from Ll to Lo in Unicode 14: the same handful of moved characters as above)
    assert ((cat == "Ll") & both & ~membership(product["prop_Lowercase"])).sum() <= 8
    assert ((cat == "Lu") & both & ~membership(product["prop_Uppercase"])).sum() <= 8
    # scripts: spot checks of blocks that have not moved since Unicode 1
    for name, lo, hi in (("script_Greek", 0x3B1, 0x3C9), ("script_Cyrillic", 0x410, 0x44F), ("script_Hebrew", 0x5D0, 0x5EA),
                         ("script_Hiragana", 0x3041, 0x3096), ("script_Katakana", 0x30A1, 0x30FA), ("script_Han", 0x4E00, 0x9FA5),
                         ("script_Latin", 0x61, 0x7A), ("script_Arabic", 0x621, 0x63A)):
        assert membership(product[name])[lo:hi + 1].all(), name
