#!/usr/bin/env python3
"""Three-way differential fuzzer for the pattern engines, on the CPU: random VALID patterns from a grammar inside what
Rust's regex and RE2 agree on (no \\d \\w \\s \\b over non-ASCII subjects, no look-around / back-references, Unicode
classes and (?i) only over characters as old as Unicode 6) against random subjects, decided by
  * the product's compiler + automaton (term_amd/csrc/regex, through tgx_regex_is_match),
  * the oracle's backtracking-free VM (oracle/regex_oracle.c, through orc.Regex),
  * RE2 (pyarrow.compute.match_substring_regex) -- an engine neither of the two shares a line with.
Any disagreement is printed with the pattern and the subject.

    python tools/fuzz_regex_diff.py [--seconds 60] [--seed 1] [--ascii]"""
import argparse
import ctypes as C
import os
import random
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LITERALS_ASCII = list("abcxyzABZ019@.-_ /:")
LITERALS_UNI = ["é", "ß", "ω", "Ω", "д", "Д", "日", "本", "ﬁ", "ı", "İ", "K", "ſ"]
CLASSES_ASCII = ["[a-c]", "[^a-c]", "[0-9]", "[A-Za-z]", "[^@]", "[a-cx-z]", "[[:alpha:]]", "[[:digit:]]", "[^[:space:]]", ".", r"\.", r"\-"]
CLASSES_PERL = [r"\d", r"\w", r"\s", r"\D", r"\W", r"\S"]           # ASCII subjects only
CLASSES_UNI = [r"\p{L}", r"\p{Lu}", r"\p{Ll}", r"\p{N}", r"\PL", r"\p{Greek}", r"\p{Cyrillic}", r"\p{Han}", "[α-ω]", "[а-я]", "[^\\x00-\\x7F]"]
QUANTS = ["", "", "", "", "*", "+", "?", "{2}", "{1,3}", "{0,2}", "{2,}", "*?", "+?", "??"]


def atom(rng, ascii_only, depth):
    r = rng.random()
    if r < 0.45:
        pool = LITERALS_ASCII if (ascii_only or rng.random() < 0.7) else LITERALS_UNI
        c = rng.choice(pool)
        return "\\" + c if c in ".-/" and rng.random() < 0.5 and c != "/" else ("\\." if c == "." else c)
    if r < 0.75:
        pool = list(CLASSES_ASCII)
        if ascii_only:
            pool += CLASSES_PERL
        else:
            pool += CLASSES_UNI
        return rng.choice(pool)
    if depth >= 2:
        return rng.choice(LITERALS_ASCII[:9])
    inner = alternation(rng, ascii_only, depth + 1)
    return ("(?:%s)" if rng.random() < 0.6 else "(%s)") % inner


def concat(rng, ascii_only, depth):
    return "".join(atom(rng, ascii_only, depth) + rng.choice(QUANTS) for _ in range(rng.randint(1, 3 if depth else 4)))


def alternation(rng, ascii_only, depth):
    return "|".join(concat(rng, ascii_only, depth) for _ in range(1 if rng.random() < 0.7 else rng.randint(2, 3)))


def pattern(rng, ascii_only):
    p = alternation(rng, ascii_only, 0)
    if rng.random() < 0.4:
        p = "^" + p if "|" not in p else "^(?:" + p + ")"
    if rng.random() < 0.4:
        p = p + "$" if "|" not in p else "(?:" + p + ")$"
    if rng.random() < 0.15:
        p = "(?i)" + p
    if rng.random() < 0.05:
        p = "(?s)" + p
    if ascii_only and rng.random() < 0.15:  # word boundaries: Rust's are Unicode-aware, RE2's ASCII -- the same over ASCII subjects
        p = rng.choice([r"\b", r"\B"]) + p if rng.random() < 0.5 else p + rng.choice([r"\b", r"\B"])
    if rng.random() < 0.08:
        p = "(?m)" + p  # ^ and $ at line feeds too
    return p


def subject(rng, ascii_only):
    pool = LITERALS_ASCII + (["\t", "\n", "\n"] if rng.random() < 0.3 else []) + ([] if ascii_only else LITERALS_UNI + ["Ж", "漢", "😀"])
    return "".join(rng.choice(pool) for _ in range(rng.randint(0, 10)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ascii", action="store_true", help="ASCII subjects, Perl classes in the grammar")
    ap.add_argument("--subjects", type=int, default=40)
    ap.add_argument("--groups", type=float, default=0.0, help="after the single patterns: this fraction of --seconds on product automata of 2 - 4 patterns")
    ap.add_argument("--counted", type=float, default=0.0, help="first: this fraction of --seconds on counted repeats {m,n} (n <= 100) of \\w \\p{L} \\d ... over non-ASCII subjects, against the PyPI regex module and the oracle (RE2's \\w \\d are ASCII)")
    args = ap.parse_args()
    import pyarrow as pa
    import pyarrow.compute as pc

    import oracle_binding as orc
    import term_amd as T

    rng = random.Random(args.seed)
    t0 = time.time()
    n_pat = n_cmp = bad = skipped = too_big = 0
    # ---- counted repeats of Unicode classes (tests/test_regex_counted_classes.py has the generators)
    if args.counted:
        import regex as pyregex
        from test_regex_counted_classes import subjects as counted_subjects

        classes = [r"\w", r"\p{L}", r"\d", r"[\p{L} ]", r"[\p{L}\p{M}\s'-]", r"[^\W\d]", r"\p{Lu}", r"[α-ωa-z]", r"\S"]
        shapes = ["^%s{%d,%d}$", "^%s{%d,%d}", "%s{%d,%d}$", "%s{%d,%d}", "^x%s{%d,%d}$", "^%s{%d,%d}@", "^(?:%s{%d,%d})$",
                  r"\A%s{%d,%d}\z", "^(%s{%d,%d}|-)$", "^%s{%d,%d}-%s{%d,%d}$"]
        n_counted = 0
        while time.time() - t0 < args.seconds * args.counted:
            c, shape = rng.choice(classes), rng.choice(shapes)
            m = rng.choice([0, 1, 2, 3, 5, 17, 40, 64])
            n = min(100, m + rng.choice([0, 1, 3, 10, 24, 36, 60]))
            pat = shape % ((c, m, n) * shape.count("%s"))
            pb = pat.encode()
            err = T._lib._Error()
            if T.lib().tgx_regex_validate(pb, len(pb), 0, C.byref(err)) != 0:
                msg = err.msg.decode(errors="replace")
                if ("DFA states" in msg or "NFA states" in msg) and not (shape.startswith("^%s") and shape.endswith("}$") and shape.count("%s") == 1):
                    too_big += 1
                    continue
                print("PRODUCT REFUSES %r: %s" % (pat, msg))
                bad += 1
                continue
            rx = orc.Regex(pat)
            py = pat.replace(r"\z", r"\Z").replace("$", r"\Z")
            n_pat += 1
            n_counted += 1
            for s in counted_subjects(rng, m, n, args.subjects):
                sb = s.encode()
                mm = C.c_int32()
                rc = T.lib().tgx_regex_is_match(pb, len(pb), 0, sb, len(sb), C.byref(mm), C.byref(err))
                got_p = None if rc != 0 else bool(mm.value)
                want = pyregex.search(py, s, pyregex.V0) is not None
                n_cmp += 1
                if got_p != want or rx.is_match(s) != want:
                    bad += 1
                    print("DISAGREE (counted class) pattern %r subject %r: product %s oracle %s regex module %s" % (pat, s, got_p, rx.is_match(s), want))
        print("%d counted-class patterns" % n_counted)
    while time.time() - t0 < args.seconds:
        ascii_only = args.ascii or rng.random() < 0.5
        pat = pattern(rng, ascii_only)
        subs = [subject(rng, ascii_only) for _ in range(args.subjects)]
        try:
            want = pc.match_substring_regex(pa.array(subs, pa.large_string()), pat).to_pylist()
        except Exception:  # (RE2 refuses it: outside the common subset after all)
            skipped += 1
            continue
        pb = pat.encode()
        err = T._lib._Error()
        if T.lib().tgx_regex_validate(pb, len(pb), 0, C.byref(err)) != 0:
            msg = err.msg.decode(errors="replace")
            if "DFA states" in msg:  # (too big for the device's table: TGX_UNSUPPORTED, the caller's fall-back -- not a verdict)
                too_big += 1
                continue
            print("PRODUCT REFUSES %r: %s" % (pat, msg))
            bad += 1
            continue
        try:
            rx = orc.Regex(pat)
        except ValueError as e:
            print("ORACLE REFUSES %r: %s" % (pat, e))
            bad += 1
            continue
        n_pat += 1
        for s, w in zip(subs, want):
            sb = s.encode()
            m = C.c_int32()
            rc = T.lib().tgx_regex_is_match(pb, len(pb), 0, sb, len(sb), C.byref(m), C.byref(err))
            got_p = None if rc != 0 else bool(m.value)
            got_o = rx.is_match(s)
            n_cmp += 1
            if got_p != w or got_o != w:
                bad += 1
                print("DISAGREE pattern %r subject %r: product %s oracle %s RE2 %s" % (pat, s, got_p, got_o, w))
        # FormatOptions::trim_before_check: the pattern sees TRIM(value) -- SQL TRIM takes the blanks (0x20) off both ends
        want_trim = pc.match_substring_regex(pa.array([s.strip(" ") for s in subs], pa.large_string()), pat).to_pylist()
        for s, w in zip(subs, want_trim):
            sb = s.encode()
            m = C.c_int32()
            rc = T.lib().tgx_regex_is_match(pb, len(pb), T.FLAG_TRIM, sb, len(sb), C.byref(m), C.byref(err))
            got_p = None if rc != 0 else bool(m.value)
            n_cmp += 1
            if got_p != w:
                bad += 1
                print("DISAGREE (flag: trim) pattern %r subject %r: product %s RE2 on the trimmed value %s" % (pat, s, got_p, w))
        # the case-insensitive FLAG (`~*`, FormatOptions::case_sensitive(false)) is the pattern under (?i)
        ci_ok = T.lib().tgx_regex_validate(pb, len(pb), T.FLAG_CASE_INSENSITIVE, C.byref(err)) == 0
        if not ci_ok:
            too_big += 1  # (folded, the automaton is over the device's table limit: a refusal, not a verdict)
        if ci_ok and not pat.startswith("(?"):
            want_ci = pc.match_substring_regex(pa.array(subs, pa.large_string()), "(?i)" + pat).to_pylist()
            rx_ci = orc.Regex(pat, case_insensitive=True)
            for s, w in zip(subs, want_ci):
                sb = s.encode()
                m = C.c_int32()
                rc = T.lib().tgx_regex_is_match(pb, len(pb), T.FLAG_CASE_INSENSITIVE, sb, len(sb), C.byref(m), C.byref(err))
                got_p = None if rc != 0 else bool(m.value)
                got_o = rx_ci.is_match(s)
                n_cmp += 1
                if got_p != w or got_o != w:
                    bad += 1
                    print("DISAGREE (flag: case-insensitive) pattern %r subject %r: product %s oracle %s RE2 %s" % (pat, s, got_p, got_o, w))
    # ---- the PRODUCT automaton of 2 - 4 patterns (what several pattern checks of one column share on the device): the
    # mask it gives a value against RE2 pattern by pattern
    L = T.lib()
    L.tgx_regex_match_group.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.c_size_t,
                                        C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_void_p]
    t1 = time.time()
    n_groups = n_grouped = 0
    while args.groups and time.time() - t1 < args.seconds * args.groups:
        ascii_only = args.ascii or rng.random() < 0.5
        k = rng.randint(2, 4)
        pats = [pattern(rng, ascii_only) for _ in range(k)]
        err = T._lib._Error()
        if any(L.tgx_regex_validate(p.encode(), len(p.encode()), 0, C.byref(err)) != 0 for p in pats):
            continue
        subs = [subject(rng, ascii_only) for _ in range(args.subjects)]
        want = [pc.match_substring_regex(pa.array(subs, pa.large_string()), p).to_pylist() for p in pats]
        enc = [p.encode() for p in pats]
        arr = (C.c_char_p * k)(*enc)
        lens = (C.c_size_t * k)(*[len(e) for e in enc])
        fl = (C.c_uint32 * k)(*([0] * k))
        n_groups += 1
        for si, s in enumerate(subs):
            sb = s.encode()
            mask, grouped = C.c_uint32(), C.c_int32()
            rc = L.tgx_regex_match_group(arr, lens, fl, k, sb, len(sb), C.byref(mask), C.byref(grouped), C.byref(err))
            n_grouped += 1 if (rc == 0 and grouped.value and si == 0) else 0
            wmask = sum((1 << j) for j in range(k) if want[j][si])
            n_cmp += 1
            if rc != 0 or mask.value != wmask:
                bad += 1
                print("DISAGREE (product automaton) patterns %r subject %r: mask %s (rc %d) RE2 %s" % (pats, s, bin(mask.value), rc, bin(wmask)))
    if args.groups:
        print("%d groups of 2 - 4 patterns (%d walked as ONE automaton)" % (n_groups, n_grouped))
    print("%d patterns, %d comparisons, %d disagreements, %d patterns RE2 refused, %d too big for the device table, %.0f s"
          % (n_pat, n_cmp, bad, skipped, too_big, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
