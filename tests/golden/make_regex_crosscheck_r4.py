#!/usr/bin/env python3
"""Writes tests/golden/regex_crosscheck_r4.json: vectors for Unicode word boundaries -- `\\b` / `\\B` as Rust's `regex`
crate takes them by default: a position with a `\\w` character on exactly one side (`\\w` = Alphabetic, M, Nd, Pc,
Join_Control; the ends of the haystack count as "not a word character").

Expected values come from the `regex` PyPI module (Unicode is its default for str patterns; `\\Z` for an unflagged
`$`).  An independent cross-check of oracle/regex_oracle.c and term_amd/csrc/regex/regex_compile.cpp (a context per
automaton state that follows a character's bytes through a classifier of \\w, and an obligation the thread carries
until the character behind the assertion is complete).

    python tests/golden/make_regex_crosscheck_r4.py
"""
import json
import os
import random

import regex

HERE = os.path.dirname(os.path.abspath(__file__))

# (rust pattern, python pattern)
PATTERNS = [
    (r"\bword\b", r"\bword\b"),
    (r"\bé", r"\bé"),
    (r"é\b", r"é\b"),
    (r"\B", r"\B"),
    (r"\b", r"\b"),
    (r"\bfoo\b|\bbar", r"\bfoo\b|\bbar"),
    (r"a\b\B", r"a\b\B"),
    (r"\b\d+\b", r"\b\d+\b"),
    (r"(?i)\bSTRASSE\b", r"(?i)\bSTRASSE\b"),
    (r"^\b", r"^\b"),
    (r"\b$", r"\b\Z"),
    (r"\B$", r"\B\Z"),
    (r"\w\b\W", r"\w\b\W"),
    (r"\B\w+\B", r"\B\w+\B"),
    (r"x\b|\by", r"x\b|\by"),
    (r"\b[[:alpha:]]{3}\b", r"\b[A-Za-z]{3}\b"),  # (POSIX classes are ASCII in Rust)
    (r"\b\p{Greek}+\b", r"\b\p{Greek}+\b"),
    (r"(?m)^\b\w+\b$", r"(?m)^\b\w+\b$"),
    (r"\bnaïve\b", r"\bnaïve\b"),
    (r"\b日本\b", r"\b日本\b"),
    (r"\Bon\b", r"\Bon\b"),
    (r"\b(?:select|drop)\b", r"\b(?:select|drop)\b"),
    (r"\b\w{1,3}\b \b", r"\b\w{1,3}\b \b"),
    (r"\b\b\w", r"\b\b\w"),
    (r"\b\B", r"\b\B"),
]

ALPHABET = list("abdorwxyzW019 .-_,\n") + ["é", "ï", "ß", "٣", "Ω", "日", "本", "\u200d", "\u0301", "😀"]
SEEDS = ["word", "a word here", "swordfish", "word1", "wörd", "éword", "word é", "日本word", "word,", "_word", " word\n",
         "", "é", "aé", "a é", "éa", "日é", ".é", "é a", "é日", "é.", "a", "ab", " ", "日本", "a b", ".", "foo", "xfoo",
         "bar", "xbar", "foo bar", "foobar", "barx", "éfoo", "foo日", "12", "a12", "12a", "x 12 y", "١٢", "x١٢",
         "strasse", "Strasse 5", "xstrasse", "straße", " a", "a ", "a .", "é.", "日 ", "abc", "日本語", "x", "xy", "y",
         "ay", "x é", "éy", "abc def", "abcd", "Ωμέγα", "αβγ δ", "xαβγ", "one\ntwo", "one two\n", "naïve", "naïvely",
         "a naïve b", "日本 語", "x日本", "on", "upon", "on it", "bonbon", "select", "drop x", "dropped", "xselect",
         "ab cd", "abcd ef", "a b c", "e\u0301", "a\u200db", "😀a", "a😀", "_", "__", "a_b", "1_2"]


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
    rng = random.Random(20261004)
    cases = []
    for rust, py in PATTERNS:
        rx = regex.compile(py)
        inputs = set(SEEDS)
        for s in SEEDS:
            for _ in range(2):
                inputs.add(mutate(rng, s))
        for s in sorted(inputs):
            cases.append({"pattern": rust, "flags": 0, "input": s, "match": rx.search(s) is not None})
    path = os.path.join(HERE, "regex_crosscheck_r4.json")
    with open(path, "w") as f:
        json.dump({"cases": cases}, f, ensure_ascii=True, indent=0)
    print("wrote", path, len(cases), "cases,", sum(c["match"] for c in cases), "matching")


if __name__ == "__main__":
    main()
