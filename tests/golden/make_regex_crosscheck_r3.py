#!/usr/bin/env python3
"""Writes tests/golden/regex_crosscheck_r3.json: vectors for the constructs added in round 3 -- multi-line anchors
(?m), ASCII word boundaries (?-u:\\b) / (?-u:\\B), (?-u) Perl classes and class set operations (&& -- ~~).

Expected values come from the `regex` PyPI module, each pattern written twice: as Rust's `regex` crate takes it and as
the `regex` module takes it (V1 for the set operations, scoped `a` flag for the ASCII forms, `\\Z` for an unflagged
`$`).  An independent cross-check of oracle/regex_oracle.c and term_amd/csrc/regex/regex_compile.cpp.

    python tests/golden/make_regex_crosscheck_r3.py
"""
import json
import os
import random

import regex

HERE = os.path.dirname(os.path.abspath(__file__))

# (rust pattern, python pattern)
PATTERNS = [
    (r"(?m)^abc$", r"(?m)^abc$"),
    (r"(?m)^\d+$", r"(?m)^\d+$"),
    (r"(?m)^$", r"(?m)^$"),
    (r"(?m)a$", r"(?m)a$"),
    (r"(?m)^b", r"(?m)^b"),
    (r"x(?m:^y)", r"x(?m:^y)"),
    (r"(?m)^a.c$", r"(?m)^a.c$"),
    (r"(?m:^)ab(?m:$)\n", r"(?m:^)ab(?m:$)\n"),
    (r"(?m)(^x|y$)", r"(?m)(^x|y$)"),
    (r"(?ms)^a.*z$", r"(?ms)^a.*z$"),
    (r"(?m)^[^@\n]+@[^@\n]+$", r"(?m)^[^@\n]+@[^@\n]+$"),
    (r"(?m)^\s*$", r"(?m)^\s*$"),
    (r"^a(?m)$", r"^a(?m:$)"),
    (r"(?-u:\b)foo(?-u:\b)", r"(?a:\b)foo(?a:\b)"),
    (r"(?-u:\bfoo\b)", r"(?a:\bfoo\b)"),
    (r"(?-u:\B)oo", r"(?a:\B)oo"),
    (r"(?-u:\b\d{3}\b)", r"(?a:\b\d{3}\b)"),
    (r"(?-u:\w+)@(?-u:\w+)", r"(?a:\w+)@(?a:\w+)"),
    (r"^(?-u:\b)", r"^(?a:\b)"),
    (r"(?-u:\b)$", r"(?a:\b)\Z"),
    (r"(?-u:\B)$", r"(?a:\B)\Z"),
    (r"(?i)(?-u:\b)select(?-u:\b)", r"(?i)(?a:\b)select(?a:\b)"),
    (r"é(?-u:\b)", r"é(?a:\b)"),
    (r"(?m)^(?-u:\b)x", r"(?m)^(?a:\b)x"),
    (r"^[a-z&&[^m]]+$", r"(?V1)^[a-z&&[^m]]+\Z"),
    (r"^[a-z--m]+$", r"(?V1)^[a-z--m]+\Z"),
    (r"^[a-f~~d-k]+$", r"(?V1)^[a-f~~d-k]+\Z"),
    (r"^[\w--\d]+$", r"(?V1)^[\w--\d]+\Z"),
    (r"^[\p{L}&&\p{ASCII}]+$", r"(?V1)^[\p{L}&&\p{ASCII}]+\Z"),
    (r"^[a-z&&b-y--m]+$", r"(?V1)^[[[a-z]&&[b-y]]--[m]]+\Z"),
    (r"^[^a-z--m]+$", r"(?V1)^[^[a-z--m]]+\Z"),
    (r"(?i)^[a-z--m]+$", r"(?V1i)^[a-z--m]+\Z"),
    (r"^[0-9&&[[:digit:]]--5]+$", r"(?V1)^[[0-9&&[[:digit:]]]--5]+\Z"),
]

ALPHABET = list("abcfmoxyzABMZ019 .-_@\n\t") + ["é", "ß", "٣", "Ω", "😀"]
SEEDS = ["abc", "abc\n", "\nabc", "x\nabc\ny", "xabc", "abc\nabcd", "", "\n", "\n\n", "a", "a\n", "b\nb", "ab\n", "x\ny",
         "xy", "12\n34", "12a\n34", "a c", "a\nc", "axc\n", "a@b\nc", "a@b", "  \n", "x\n \ny", "a\nb\nz", "az", "a1z\nz",
         "foo", "a foo b", "afoo", "foob", "foo.", "éfoo", "fooé", "_foo", "foo_", "1foo", "food", "oo", "foo oo",
         "123", "a123", "123 456", "1234", "abc@def", "é@é", "select", "SELECT x", "selects", "é", "aé", "éa", "x", " x",
         "abcxyz", "amz", "m", "dk", "abgk", "defgk", "a1", "9", "٣", "Ω", "aΩ", "bcx", "n", "MZ", "bly", "05", "12", "5"]


def mutate(rng, s):
    s = list(s)
    for _ in range(rng.randint(0, 3)):
        op, pos = rng.randint(0, 2), rng.randint(0, len(s))
        if op == 0:
            s.insert(pos, rng.choice(ALPHABET))
        elif op == 1 and s:
            del s[min(pos, len(s) - 1)]
        elif s:
            s[min(pos, len(s) - 1)] = rng.choice(ALPHABET)
    return "".join(s)


def main():
    rng = random.Random(20261003)
    cases = []
    for rust, py in PATTERNS:
        rx = regex.compile(py)
        inputs = set(SEEDS)
        for s in SEEDS:
            for _ in range(2):
                inputs.add(mutate(rng, s))
        for s in sorted(inputs):
            cases.append({"pattern": rust, "flags": 0, "input": s, "match": rx.search(s) is not None})
    # still outside the engine: byte classes that could match invalid UTF-8, CRLF mode, the \b{..} variants, ASCII and
    # Unicode word boundaries in one pattern (Unicode word boundaries themselves: round 4, make_regex_crosscheck_r4.py)
    unsupported = [r"(?-u:.)", r"(?-u:\W)", r"(?-u:[^a])", r"(?mR)^a$", r"\b{start}x", r"\bfoo(?-u:\b)"]
    path = os.path.join(HERE, "regex_crosscheck_r3.json")
    with open(path, "w") as f:
        json.dump({"cases": cases, "unsupported": unsupported}, f, ensure_ascii=True, indent=0)
    print("wrote", path, len(cases), "cases,", sum(c["match"] for c in cases), "matching")


if __name__ == "__main__":
    main()
