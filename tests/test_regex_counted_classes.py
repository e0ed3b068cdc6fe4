"""CPU: counted repeats of Unicode classes -- "a name of 2 to 50 letters", "a user name of 1 to 64 word characters" -- in
the product's compiler (term_amd/csrc/regex/regex_compile.cpp) against the PyPI `regex` module and the oracle's VM.
Round 5 refused them ("more than 20000 DFA states" / "too many NFA states"): \\w is Unicode in Rust's regex
(TG/security.rs:152-183 lets every pattern of <= 1000 characters through that compiles), and every repetition brought a
copy of the class's UTF-8 trie.  Now a class is a minimal acyclic automaton with shared suffixes, and `^C{m,n}$` is the
automaton of `^C*$` plus a character count (Dfa::len_min / len_max).  Subjects stay within characters as old as Unicode
6 (the module is at 17.0, the tables at 16.0)."""
import random

import pytest

import oracle_binding as orc
from test_regex_host import product_is_match, product_validate

regex = pytest.importorskip("regex")

VERDICT_PATTERNS = [r"^\w{1,64}$", r"^[\p{L} ]{2,50}$", r"^[\p{L}\p{M}\s'-]{1,100}$", r"^\p{L}{1,100}$"]
LETTERS = list("azAZ") + ["é", "ß", "ω", "Ω", "д", "Ж", "日", "本", "한", "ﬁ"]
DIGITS = list("09") + ["٣", "５", "߉"]          # Nd outside ASCII
MARKS = ["́", "̈", "ा"]          # Mn / Mc
CONNECT = ["_", "‿"]                       # Pc
SPACES = [" ", "\t", " ", " "]
OTHER = list("-'!.@") + ["€", "😀", "​"]


def subjects(rng, m, n, count):
    pools = [LETTERS, LETTERS + DIGITS, LETTERS + MARKS + SPACES + ["'", "-"], LETTERS + DIGITS + MARKS + CONNECT + OTHER]
    out = ["", " ", "a", "é"]
    for _ in range(count):
        pool = rng.choice(pools)
        # lengths around the bounds, where the verdict changes
        target = rng.choice([0, 1, max(m - 1, 0), m, m + 1, (m + n) // 2, max(n - 1, 0), n, n + 1, n + 7])
        s = "".join(rng.choice(pool) for _ in range(target))
        if rng.random() < 0.3 and s:
            k = rng.randrange(len(s))
            s = s[:k] + rng.choice(OTHER + SPACES) + s[k + 1:]
        if rng.random() < 0.1:
            s = " " + s + " "
        out.append(s)
    return out


def module_search(pat, s, flags=0):
    return regex.search(pat, s, flags | regex.V0) is not None


@pytest.mark.parametrize("pat", VERDICT_PATTERNS)
def test_the_patterns_of_the_review_compile_and_agree(pat):
    rc, msg = product_validate(pat)
    assert rc == 0, msg
    rng = random.Random(hash(pat) & 0xFFFF)
    import re

    m, n = (int(x) for x in re.search(r"\{(\d+),(\d+)\}", pat).groups())
    rx = orc.Regex(pat)
    for s in subjects(rng, m, n, 400):
        want = module_search(pat.replace("$", r"\Z"), s)
        assert product_is_match(pat, s) == want, (pat, s)
        assert rx.is_match(s) == want, (pat, s)


def test_counted_classes_in_every_position():
    """anchored on both sides (the count form), on one side, on none, next to literals, nested, (?i), under TRIM"""
    import term_amd as T

    rng = random.Random(5)
    classes = [r"\w", r"\p{L}", r"\d", r"[\p{L} ]", r"[\p{L}\p{M}'-]", r"[^\W\d]", r"\p{Lu}", r"[α-ωa-z]"]
    shapes = ["^%s{%d,%d}$", "^%s{%d,%d}", "%s{%d,%d}$", "%s{%d,%d}", "^x%s{%d,%d}$", "^%s{%d,%d}@", r"^(?:%s{%d,%d})$",
              r"\A%s{%d,%d}\z", "^(%s{%d,%d}|-)$"]
    # ((?i) over PROPERTY classes is where the module is not Rust -- it takes \p{Lu} for "any cased letter", Rust closes
    #  the class under simple case folding: tests/golden/make_regex_crosscheck_r5.py -- so the flag has its own test below)
    n_pat = n_refused = 0
    for _ in range(60):
        c, shape = rng.choice(classes), rng.choice(shapes)
        m = rng.choice([0, 1, 2, 3, 5, 17, 40])
        n = m + rng.choice([0, 1, 3, 10, 24, 60])
        n = min(n, 100)
        pat = shape % (c, m, n)
        rc, msg = product_validate(pat)
        if rc != 0:  # a size refusal is the caller's fall-back, never a wrong verdict -- but the anchored form never is one
            assert "DFA states" in msg or "NFA states" in msg, (pat, msg)
            assert not (shape in ("^%s{%d,%d}$", r"^(?:%s{%d,%d})$", r"\A%s{%d,%d}\z")), (pat, msg)
            n_refused += 1
            continue
        n_pat += 1
        py = pat.replace(r"\z", r"\Z").replace("$", r"\Z")
        for s in subjects(rng, m, n, 60):
            want = module_search(py, s)
            assert product_is_match(pat, s) == want, (pat, s)
            assert product_is_match(pat, s, T.FLAG_TRIM) == module_search(py, s.strip(" ")), (pat, s)
    assert n_pat >= 40, (n_pat, n_refused)


def test_case_insensitive_flag_on_the_count_form():
    import term_amd as T

    pat = r"^[a-zß]{2,40}$"
    rc, msg = product_validate(pat, T.FLAG_CASE_INSENSITIVE)
    assert rc == 0, msg
    for s, want in (("AB", True), ("aB" * 20, True), ("aB" * 20 + "c", False), ("ẞẞ", True), ("a", False), ("a1", False)):
        assert product_is_match(pat, s, T.FLAG_CASE_INSENSITIVE) == want, s
