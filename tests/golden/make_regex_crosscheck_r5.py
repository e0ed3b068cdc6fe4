#!/usr/bin/env python3
"""Writes tests/golden/regex_crosscheck_r5.json: vectors for `(?i)` / `~*` / `case_sensitive = false`
(/root/reference/term-guard/src/constraints/format.rs:756-776) on the characters where engines and Unicode versions
part ways -- the Turkic i's, the Kelvin and long-s signs, capital sharp s, the status-S lines of CaseFolding-15.1.0
(U+FB05 ~ U+FB06, U+1FD3 ~ U+0390, U+1FE3 ~ U+03B0), one cased pair of each of Unicode 14.0 / 16.0, and pairs 17.0 added
(which Rust's regex-syntax 0.8.8, Unicode 16.0, does not know).

Expected values come from the `regex` PyPI module (Unicode 17.0 here), asked under regex.V0 (simple case folding) --
with two corrections, made by SUBSTITUTION before the module is asked, because the module is not Rust there:
  * the module folds I ~ U+0131 and i ~ U+0130 (CaseFolding.txt's status-T lines); Rust's simple folding has neither.
    U+0131 is replaced by U+0138 (LATIN SMALL LETTER KRA: Ll, Latin, Alphabetic, no case mapping at all) and U+0130 by
    U+03D2 (GREEK UPSILON WITH HOOK SYMBOL: Lu, no case mapping) in pattern and input alike: the same classes, no
    folding.  (No pattern below asks for a script.)
  * code points Unicode 17.0 added are unassigned for regex-syntax 0.8.8: each is replaced by a code point that is
    unassigned in 17.0 as well (U+0378, U+0379, U+0380, U+0381).

Left out on purpose: Unicode property classes under `(?i)` -- the module then takes \\p{Lu}, \\p{Ll}, \\p{Lt} for one
class of cased letters, while regex-syntax closes a class under the fold pairs and only then negates it (the engines'
rule: `(?i)^\\p{Lu}$` matches U+0345 through U+0399 and does not match U+0390, whose partner U+1FD3 is lower case too);
U+0295, whose General_Category differs between Unicode 14 and 17 (tests/test_unicode_tables.py).

An independent cross-check of oracle/regex_oracle.c (tables: ICU 70 + written CaseFolding lines) and
term_amd/csrc/regex/regex_compile.cpp (tables: the `regex` module probed, corrected the same way but by code that
shares nothing with this script).

    python tests/golden/make_regex_crosscheck_r5.py
"""
import json
import os
import random

import regex

HERE = os.path.dirname(os.path.abspath(__file__))

RUST_VIEW = {0x131: 0x138, 0x130: 0x3D2, 0xA7CE: 0x378, 0xA7CF: 0x379, 0x16EA0: 0x380, 0x16EBB: 0x381}
for cp in (0x138, 0x3D2):
    assert regex.findall("(?i)" + chr(cp), "".join(map(chr, range(0x20, 0x3000))), flags=regex.V0) == [chr(cp)]
for cp in (0x378, 0x379, 0x380, 0x381):
    assert regex.fullmatch(r"\p{Cn}", chr(cp))
assert regex.fullmatch(r"\p{Ll}", "ĸ") and regex.fullmatch(r"\p{Lu}", "ϒ")


def rust_view(s):
    return s.translate(RUST_VIEW)


FLAG_CI = 8
# (rust pattern, python pattern, flags) -- `$` is `\Z` for the module; flags = 8 is TGX_FLAG_CASE_INSENSITIVE, what `~*` sets
PATTERNS = [
    (r"(?i)^i$", r"(?i)^i\Z", 0),
    (r"^i$", r"^i\Z", FLAG_CI),
    (r"^i$", r"^i\Z", 0),
    (r"(?i)^[a-z]+$", r"(?i)^[a-z]+\Z", 0),
    (r"^[A-Z]+$", r"^[A-Z]+\Z", FLAG_CI),
    (r"(?i)İ", "(?i)İ", 0),
    (r"(?i)ı", "(?i)ı", 0),
    ("^İstanbul$", "^İstanbul\\Z", FLAG_CI),
    (r"(?i)k", r"(?i)k", 0),
    (r"(?i)^[k]$", r"(?i)^[k]\Z", 0),
    ("(?i)^K$", "(?i)^K\\Z", 0),
    (r"(?i)[^k]", r"(?i)[^k]", 0),
    (r"(?i)[^i]", r"(?i)[^i]", 0),
    (r"(?i)s+t", r"(?i)s+t", 0),
    ("(?i)^ſ", "(?i)^ſ", 0),
    ("straße", "straße", FLAG_CI),
    ("(?i)^ẞ$", "(?i)^ẞ\\Z", 0),
    ("(?i)[ß]", "(?i)[ß]", 0),
    ("(?i)ⱟ", "(?i)ⱟ", 0),
    ("Ꟁ", "Ꟁ", FLAG_CI),
    ("(?i)\U00010570", "(?i)\U00010570", 0),
    ("(?i)^[\U00010570-\U0001057a]+$", "(?i)^[\U00010570-\U0001057a]+\\Z", 0),
    ("(?i)ﬅ", "(?i)ﬅ", 0),
    ("ﬆ", "ﬆ", FLAG_CI),
    ("(?i)^ΐ$", "(?i)^ΐ\\Z", 0),
    ("(?i)ΰ", "(?i)ΰ", 0),
    ("(?i)Ɤ", "(?i)Ɤ", 0),
    ("ɤ", "ɤ", FLAG_CI),
    ("(?i)\U00010d50", "(?i)\U00010d50", 0),
    ("(?i)^[\U00010d50-\U00010d65]+$", "(?i)^[\U00010d50-\U00010d65]+\\Z", 0),
    ("(?i)Ᲊ", "(?i)Ᲊ", 0),
    ("(?i)ꟛ", "(?i)ꟛ", 0),
    ("(?i)ƛ", "(?i)ƛ", 0),
    ("(?i)꟎", "(?i)꟎", 0),
    ("\U00016ea0", "\U00016ea0", FLAG_CI),
    (r"^\p{Ll}+$", r"^\p{Ll}+\Z", 0),
    (r"^\p{Lu}+$", r"^\p{Lu}+\Z", 0),
    (r"(?i)^\w+$", r"(?i)^\w+\Z", 0),
    (r"^\w+$", r"^\w+\Z", 0),
    (r"(?i)DIYARBAKIR", r"(?i)DIYARBAKIR", 0),
    (r"(?i)^(?:k|s)+$", r"(?i)^(?:k|s)+\Z", 0),
    (r"(?i)a(?-i)b", r"(?i)a(?-i)b", 0),
    (r"(?i)[Ā-įĹ-ſ]", r"(?i)[Ā-įĹ-ſ]", 0),   # (Latin Extended-A without the two Turkic letters and the stand-in U+0138)
    (r"(?i)^[ɐ-ʯ]$", r"(?i)^[ɐ-ʯ]\Z", 0),
]

SPECIAL = ["i", "I", "ı", "İ", "k", "K", "K", "s", "S", "ſ", "ß", "ẞ", "Ⱟ", "ⱟ",
           "Ꟁ", "ꟁ", "\U00010570", "\U00010597", "\U00010571", "\U00010598", "ﬅ", "ﬆ", "ΐ",
           "ΐ", "ΰ", "ΰ", "Ɤ", "ɤ", "\U00010d50", "\U00010d70", "\U00010d65", "\U00010d85",
           "Ᲊ", "ᲊ", "Ꟛ", "ꟛ", "Ƛ", "ƛ", "꟎", "꟏", "\U00016ea0", "\U00016ebb",
           "å", "Å", "Å", "ͅ", "ι", "Ι", "ι", "Ǆ", "ǅ", "ǆ", "σ",
           "ς", "Σ"]
ALPHABET = list("abikstxzAIKST 1_") + SPECIAL
SEEDS = SPECIAL + ["", "x", "ix", "ss", "SS", "st", "ST", "sst", "ſt", "straße", "STRASSE", "STRAẞE", "Straße",
                   "İstanbul", "istanbul", "ISTANBUL", "i̇stanbul", "ıspanak", "Ispanak", "DIYARBAKIR",
                   "DİYARBAKIR", "diyarbakır", "diyarbakir", "kelvin K", "KkK", "ab", "aB", "AB", "Ab",
                   "\U00010570\U00010571", "\U00010597\U00010598", "\U00010d50\U00010d85", "\U00010d70x", "a꟎b",
                   "Ǆǅǆ", "Σσς", "åÅÅ", "1", "_", " "]


def mutate(rng, s):
    s = list(s)
    for _ in range(rng.randint(0, 2)):
        op, pos = rng.randint(0, 2), rng.randint(0, len(s))
        if op == 0:
            s.insert(pos, rng.choice(ALPHABET))
        elif op == 1 and s:
            del s[min(pos, len(s) - 1)]
        elif s:
            s[min(pos, len(s) - 1)] = rng.choice(ALPHABET)
    return "".join(s)


def main():
    rng = random.Random(20261005)
    cases = []
    for rust, py, flags in PATTERNS:
        rx = regex.compile(rust_view(py), flags=regex.V0 | (regex.I if flags & FLAG_CI else 0))
        inputs = set(SEEDS)
        for s in SEEDS:
            inputs.add(mutate(rng, s))
        for s in sorted(inputs):
            cases.append({"pattern": rust, "flags": flags, "input": s, "match": rx.search(rust_view(s)) is not None})
    path = os.path.join(HERE, "regex_crosscheck_r5.json")
    with open(path, "w") as f:
        json.dump({"cases": cases}, f, ensure_ascii=True, indent=0)
    print("wrote", path, len(cases), "cases,", sum(c["match"] for c in cases), "matching")


if __name__ == "__main__":
    main()
