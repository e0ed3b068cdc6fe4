#!/usr/bin/env python3
"""Writes tests/golden/regex_crosscheck.json: (pattern, flags, input, expected is_match) vectors.

Expected values come from the `regex` PyPI module (Unicode-aware \\d and \\s like Rust's `regex`), with the
two documented semantic differences of Python removed: an unescaped `$` is translated to `\\Z` (Rust's `$`
matches only at the very end of the haystack) and inputs are searched unanchored (`regex.search`).  This is
an independent cross-check of the pattern front-ends (oracle/regex_oracle.c and
term_amd/csrc/regex/regex_compile.cpp); the reference's own vectors live in reference_vectors.json.

    python tests/golden/make_regex_crosscheck.py
"""
import json
import os
import random

import regex

HERE = os.path.dirname(os.path.abspath(__file__))


def to_python(pattern):
    """translate Rust-regex syntax to the Python `regex` module: `$` -> `\\Z` outside classes"""
    out = []
    i, depth = 0, 0
    while i < len(pattern):
        c = pattern[i]
        if c == "\\":
            nxt = pattern[i + 1]
            if nxt in "xu" and pattern[i + 2:i + 3] == "{":
                end = pattern.index("}", i)
                out.append(regex.escape(chr(int(pattern[i + 3:end], 16))))
                i = end + 1
                continue
            if nxt == "z":
                out.append("\\Z")
            elif nxt == "A":
                out.append("\\A")
            else:
                out.append(pattern[i:i + 2])
            i += 2
            continue
        if c == "[":
            depth += 1
        elif c == "]" and depth:
            depth -= 1
        if c == "$" and depth == 0:
            out.append("\\Z")
        else:
            out.append(c)
        i += 1
    return "".join(out)


EXTRA_PATTERNS = [
    (r"@", 0), (r"^[^@]+@[^@]+\.[^@]+$", 0), (r"\d+", 0), (r"^\d{2,4}-\d+$", 0), (r"^\s*$", 0),
    (r"^(foo|bar|ba)z*$", 0), (r"(?i)^select\s+\d", 0), (r"^a.c$", 0), (r"(?s)^a.c$", 0), (r"a|b|", 0),
    (r"^(ab)*$", 0), (r"^(a|ab)(c|bcd)(d*)$", 0), (r"x{0}y", 0), (r"^[[:alpha:]]+[[:digit:]]{2}$", 0),
    (r"^[^\d\s]+$", 0), (r"^\p{Lu}\p{Ll}+$", 0), (r"^[\p{Greek}]+$", 0), (r"café", 0), (r"(?i)straße", 0),
    (r"^[a-f0-9]{8}$", 8), (r"hello", 8), (r"^[A-Z]{3}\d{3}$", 8), (r"^\x41\x{42}C$", 0), (r"^\.\*\+\?$", 0),
    (r"^(?:a(?:b(?:c)?)?)?$", 0), (r"^a{2,}$", 0), (r"(?x) ^ a b \s c $ # comment", 0), (r"^[]a]+$", 0),
    (r"^[a\-z]+$", 0), (r"^[a-]+$", 0), (r"\$\d+\.\d{2}", 0), (r"^$", 0), (r"^.*$", 0), (r"\Aab\z", 0),
    (r"^\D+$", 0), (r"^\S+@\S+$", 0), (r"(^a|b$)", 0), (r"^(?i:ab)c$", 0), (r"^[\t\n ]x$", 0), (r"\u{1F600}", 0),
]

ALPHABET = list("abcABCxyz019 .-_@:/+()[]{}'\"\\\t\n") + ["é", "ß", "ẞ", "٣", "४", "Ω", "ω", "я", "Я", "😀", " ",
                                                             " ", "K", "k", "K", "ſ", "S", "s"]
SEEDS = ["test@example.com", "user@domain.org", "invalid-email", "https://example.com/path?q=1", "http://localhost",
         "4111-1111-1111-1111", "(555) 123-4567", "12345-6789", "550e8400-e29b-41d4-a716-446655440000",
         "192.168.1.1", "256.256.256.256", "2001:db8:85a3::8a2e:370:7334", "::1", '{"key": "value"}', "[1, 2, 3]",
         "2023-12-25T10:30:00.123Z", "2023-12-25T10:30:00+05:30", "123-45-6789", "078-05-1120", "ABC123", "abc123",
         "SELECT 1", "select  42", "foozz", "baz", "abab", "abcd", "abbcdd", "y", "xy", "aa", "a\nc", "abc", "ab1",
         "Straße", "STRASSE", "strasse", "STRAẞE", "café", "CAFÉ", "Ωmega", "ωΩ", "Hello", "hello world", "$12.50",
         "٣٣", "12٣4-5", " \t ", "", "a b c", "😀", "K", "]a]", "a-z", ".*+?", "ABC", "x", " x", "\nx"]


def mutate(rng, s):
    s = list(s)
    for _ in range(rng.randint(0, 3)):
        op = rng.randint(0, 2)
        pos = rng.randint(0, len(s))
        if op == 0:
            s.insert(pos, rng.choice(ALPHABET))
        elif op == 1 and s:
            del s[min(pos, len(s) - 1)]
        elif s:
            s[min(pos, len(s) - 1)] = rng.choice(ALPHABET)
    return "".join(s)


def main():
    with open(os.path.join(HERE, "reference_vectors.json")) as f:
        builtin = json.load(f)["patterns"]
    patterns = [(p, 0) for p in sorted(set(builtin.values()))] + EXTRA_PATTERNS
    patterns += [(builtin["email"], 8), (builtin["uuid"], 8), (builtin["postal_code_UK"], 8)]
    rng = random.Random(20261002)
    cases = []
    for pat, flags in patterns:
        py = regex.compile(to_python(pat), regex.IGNORECASE if flags & 8 else 0)
        inputs = set(SEEDS)
        for s in SEEDS:
            for _ in range(3):
                inputs.add(mutate(rng, s))
        for s in sorted(inputs):
            cases.append({"pattern": pat, "flags": flags, "input": s, "match": py.search(s) is not None})
    invalid = [r"(", r")", r"a**b(", r"[a", r"a{2,1}", r"\q", r"(?=a)", r"(?<!a)b", r"\1", r"*a", r"a{", r"[z-a]",
               r"(?P<n>a)(?P=n)", r"(?z)a", r"\p{NoSuchProperty}x"]
    out = {"cases": cases, "invalid": invalid,
           "rejected_by_validate_regex_pattern": ["(.*)*", "x(.*)+y", "(a+)+", "(a*)*b", "a" * 1001]}
    path = os.path.join(HERE, "regex_crosscheck.json")
    with open(path, "w") as f:
        json.dump(out, f, ensure_ascii=True, indent=0)
    print("wrote", path, len(cases), "cases,", sum(c["match"] for c in cases), "matching")


if __name__ == "__main__":
    main()
