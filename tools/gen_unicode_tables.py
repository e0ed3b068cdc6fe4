#!/usr/bin/env python3
"""Generates the Unicode range tables of the PRODUCT's pattern compiler:
    term_amd/csrc/regex/unicode_tables.h   (pattern -> byte DFA compiler)

(The oracle's tables are NOT this file: oracle/gen_unicode_oracle.py derives oracle/unicode_oracle_tables.h from other
databases by another method, and tests/test_unicode_tables.py holds the two against each other.)

Target: what Rust's regex-syntax 0.8.8 ships (the reference's dependency, /root/reference/Cargo.lock:3637-3661):
Unicode 16.0 tables, `(?i)` = SIMPLE case folding -- the lines of CaseFolding.txt with status C or S; the Turkic lines
(T) and the full foldings (F) are not part of it (format.rs:756-760 turns `case_sensitive = false` into `~*`).

Source of the data: the `regex` PyPI module's database (Unicode 17.0 in this image's regex 2026.7.19), probed:
  * classes: every scalar value against `\\p{..}`;
  * fold orbits: every cased / case-changing code point as the pattern `(?i)c` under regex.V0 (the module's simple
    case folding) against all the others -- then the two Turkic pairs the module adds (I ~ U+0131, i ~ U+0130) are taken
    out;
  * Unicode 17.0 -> 16.0: the 4 803 code points 17.0 added (tools/unicode_versions.py, a hand-written list whose
    arithmetic closes) are unassigned here: out of every class but Cn / C, their 28 fold pairs dropped.
Properties of OLDER characters that 17.0 changed cannot be told from this image and stay as 17.0 has them (DESIGN.md
lists non-ASCII pattern parity as unpinned for that reason).

    python tools/gen_unicode_tables.py
"""
import os
import sys

import regex

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import unicode_versions as UV

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAXCP = 0x10FFFF


def is_scalar(cp):
    return not (0xD800 <= cp <= 0xDFFF)


def ranges_of(pred):
    out = []
    start = None
    for cp in range(MAXCP + 2):
        ok = cp <= MAXCP and is_scalar(cp) and pred(cp)
        if ok and start is None:
            start = cp
        elif not ok and start is not None:
            out.append((start, cp - 1))
            start = None
    return out


def class_ranges(pattern, unassigned_too=False):
    """`unassigned_too`: the class holds unassigned code points (Cn, C) -- 17.0's additions belong to it; they are out
    of every other class"""
    rx = regex.compile(pattern)
    if unassigned_too:
        return ranges_of(lambda cp: UV.added_in_17(cp) or rx.fullmatch(chr(cp)) is not None)
    return ranges_of(lambda cp: not UV.added_in_17(cp) and rx.fullmatch(chr(cp)) is not None)


def fold_orbits():
    """{code point: the set of code points `(?i)` makes it equal to (itself included)} for every code point whose set
    has more than one member -- Unicode 16.0, status C + S"""
    cand_rx = regex.compile(r"[\p{Cased}\p{CWCF}\p{CWCM}\p{CWL}\p{CWU}\p{CWT}\p{Case_Ignorable}]")
    cands = [cp for cp in range(MAXCP + 1) if is_scalar(cp) and cand_rx.fullmatch(chr(cp))]
    hay = "".join(chr(c) for c in cands)
    orb = {}
    for cp in cands:
        hits = regex.findall("(?i)" + regex.escape(chr(cp)), hay, flags=regex.V0)
        members = set(ord(x) for x in hits)
        if members != {cp}:
            orb[cp] = members
    # nothing outside the candidates is equal to one of them
    every = regex.compile("(?i)[" + "".join(regex.escape(chr(c)) for c in sorted(orb)) + "]", flags=regex.V0)
    stray = [cp for cp in range(MAXCP + 1) if is_scalar(cp) and cp not in orb and every.fullmatch(chr(cp))]
    assert not stray, [hex(c) for c in stray[:8]]
    # the Turkic lines (status T): the module has I ~ dotless i and i ~ dotted I
    assert orb[0x49] == {0x49, 0x69, 0x131} and orb[0x69] == {0x49, 0x69, 0x130}, "the module's Turkic rule changed"
    assert orb[0x130] == {0x130, 0x69} and orb[0x131] == {0x131, 0x49}
    orb[0x49] = orb[0x69] = {0x49, 0x69}
    del orb[0x130], orb[0x131]
    # Unicode 17.0's pairs
    for cp in [c for c, members in orb.items() if any(UV.added_in_17(m) for m in members)]:
        assert all(UV.added_in_17(m) for m in orb[cp]) or cp in (0xA7D3, 0xA7D5) or orb[cp] & {0xA7D3, 0xA7D5}, hex(cp)
        del orb[cp]
    # what is left is an equivalence relation
    for cp, members in orb.items():
        for m in members:
            assert orb[m] == members, (hex(cp), hex(m))
    return orb


def main():
    tables = {}
    # Perl classes as Rust's regex defines them (regex-syntax: \d = Nd, \s = White_Space,
    # \w = Alphabetic + M + Nd + Pc + Join_Control)
    tables["perl_digit"] = class_ranges(r"\p{Nd}")
    tables["perl_space"] = class_ranges(r"\p{White_Space}")
    tables["perl_word"] = class_ranges(r"[\p{Alphabetic}\p{M}\p{Nd}\p{Pc}\p{Join_Control}]")
    gcs = ["L", "Lu", "Ll", "Lt", "Lm", "Lo", "M", "Mn", "Mc", "Me", "N", "Nd", "Nl", "No", "P", "Pc", "Pd",
           "Ps", "Pe", "Pi", "Pf", "Po", "S", "Sm", "Sc", "Sk", "So", "Z", "Zs", "Zl", "Zp", "C", "Cc", "Cf",
           "Co", "Cn"]
    for gc in gcs:
        tables["gc_" + gc] = class_ranges(r"\p{%s}" % gc, unassigned_too=gc in ("C", "Cn"))
    for prop in ["Alphabetic", "White_Space", "Lowercase", "Uppercase"]:
        tables["prop_" + prop] = class_ranges(r"\p{%s}" % prop)
    for script in ["Latin", "Greek", "Cyrillic", "Han", "Hiragana", "Katakana", "Arabic", "Hebrew"]:
        tables["script_" + script] = class_ranges(r"\p{Script=%s}" % script)

    fold_pairs = []  # (cp, other) for every ordered pair in an orbit, sorted by cp
    for cp, members in fold_orbits().items():
        for other in members:
            if other != cp:
                fold_pairs.append((cp, other))
    fold_pairs.sort()

    names = sorted(tables.keys())
    lines = []
    lines.append("// GENERATED by tools/gen_unicode_tables.py -- do not edit.")
    lines.append("// Unicode data: regex module %s (Unicode 17.0) minus the code points 17.0 added = Unicode 16.0;" %
                 regex.__version__)
    lines.append("// fold pairs: CaseFolding.txt status C + S (no Turkic T, no full F), as regex-syntax 0.8.8 folds.")
    lines.append("#pragma once")
    lines.append("#include <stdint.h>")
    lines.append("typedef struct { uint32_t lo, hi; } tgx_urange;")
    lines.append("typedef struct { const char *name; const tgx_urange *ranges; uint32_t count; } tgx_utable;")
    for n in names:
        r = tables[n]
        body = ",".join("{0x%X,0x%X}" % (a, b) for a, b in r)
        lines.append("static const tgx_urange tgx_ur_%s[] = {%s};" % (n, body))
    lines.append("static const tgx_utable tgx_utables[] = {")
    for n in names:
        lines.append('  {"%s", tgx_ur_%s, %d},' % (n, n, len(tables[n])))
    lines.append("};")
    lines.append("static const uint32_t tgx_n_utables = %d;" % len(names))
    lines.append("// simple case folding: every (code point, equivalent code point) pair, sorted by the first")
    lines.append("static const uint32_t tgx_fold_pairs[][2] = {%s};" %
                 ",".join("{0x%X,0x%X}" % p for p in fold_pairs))
    lines.append("static const uint32_t tgx_n_fold_pairs = %d;" % len(fold_pairs))
    text = "\n".join(lines) + "\n"
    for rel in ("term_amd/csrc/regex/unicode_tables.h",):
        path = os.path.join(ROOT, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)
        print("wrote", rel, len(text), "bytes")


if __name__ == "__main__":
    main()
