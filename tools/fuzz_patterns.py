#!/usr/bin/env python3
"""Pattern fuzzer for the HOST side of the regex path (csrc/regex/regex_compile.cpp behind tgx_regex_validate /
tgx_regex_is_match / tgx_regex_match_group): random and mutated patterns -- most of them malformed -- and random
subjects.  Nothing here needs a GPU.  Meant to run against the sanitizer build (tools/run_host_asan.sh): the assertion
is that the compiler neither crashes nor trips ASan / UBSan, that every outcome is a clean status, and that a pattern
that compiles gives the same verdict alone and inside a product automaton.

    python tools/fuzz_patterns.py [--seconds 60] [--seed 1]
"""
import argparse
import ctypes as C
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

ATOMS = ["a", "b", "Z", "0", "9", "@", ".", "\\.", "\\d", "\\w", "\\s", "\\D", "\\W", "\\S", "[a-z]", "[^@]", "[0-9a-fA-F]",
         "[[:alpha:]]", "\\p{L}", "\\pN", "é", "中", "\\x41", "\\u{1F600}", ".", "(?:ab)", "(a|b)", "^", "$", "\\b", "\\B",
         "(?i)", "(?m)", "(?s)", "(?x)", "[a&&b]", "[a-z&&[^m]]", "\\A", "\\z", "(", ")", "[", "]", "{", "}", "|", "\\",
         "(?P<n>x)", "(?<n>x)", "\\1", "(?=a)", "(?!a)", "[]", "[^]", "[z-a]", "\\", "\\Q", "\\E", "\xff", "\x00"]
QUANT = ["", "", "", "*", "+", "?", "{2}", "{1,3}", "{0,}", "{,3}", "{61}", "{1000}", "{3,1}", "*?", "+?", "??", "**", "{"]
SEEDS = [r"^[a-zA-Z0-9._%+-]+@[a-zA-Z0-9.-]+\.[a-zA-Z]{2,}$", r"^\d{3}-\d{2}-\d{4}$", r"^https?://[^\s]+$",
         r"^[0-9a-fA-F]{8}-[0-9a-fA-F]{4}-[0-9a-fA-F]{4}-[0-9a-fA-F]{4}-[0-9a-fA-F]{12}$", r"^(\d{1,3}\.){3}\d{1,3}$",
         r"(a+)+$", r"(a|a)*", r"[^@]+@[^@]+\.[^@]+", r"^\s*\{.*\}\s*$"]


def random_pattern(rng):
    if rng.random() < 0.3:
        p = list(rng.choice(SEEDS))
        for _ in range(rng.randint(1, 4)):  # mutate a known-good pattern
            k = rng.randrange(len(p) + 1)
            op = rng.random()
            if op < 0.4 and p:
                del p[min(k, len(p) - 1)]
            elif op < 0.8:
                p.insert(k, rng.choice(ATOMS + QUANT))
            else:
                p.insert(k, chr(rng.randint(1, 0x2FF)))
        return "".join(p)
    return "".join(rng.choice(ATOMS) + rng.choice(QUANT) for _ in range(rng.randint(0, 12)))


def random_subject(rng):
    n = rng.choice([0, 1, 2, 5, 17, 64, 300])
    alphabet = "ab@.09 Zé中\n\t-_:/{}x"
    return "".join(rng.choice(alphabet) for _ in range(n)).encode("utf-8", "surrogatepass") + (b"\xff\xfe" if rng.random() < 0.05 else b"")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from term_amd._lib import _Error, lib

    L = lib()
    L.tgx_regex_validate.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_void_p]
    L.tgx_regex_is_match.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.c_void_p]
    L.tgx_regex_match_group.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.c_size_t,
                                        C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_void_p]
    rng = random.Random(args.seed)
    t_end = time.time() + args.seconds
    n = ok = grouped_runs = 0
    statuses = {}
    good = []
    while time.time() < t_end:
        pat = random_pattern(rng).encode("utf-8", "surrogatepass")
        flags = rng.choice([0, 0, 4, 8, 12])  # TRIM / CASE_INSENSITIVE
        err = _Error()
        st = L.tgx_regex_validate(pat, len(pat), flags, C.byref(err))
        statuses[st] = statuses.get(st, 0) + 1
        assert st in (0, 1, 2), (st, pat)  # OK / INVALID_ARGUMENT / UNSUPPORTED -- never INTERNAL, never a crash
        n += 1
        if st != 0:
            continue
        ok += 1
        good.append((pat, flags))
        for _ in range(4):
            sub = random_subject(rng)
            m = C.c_int32(-1)
            st = L.tgx_regex_is_match(pat, len(pat), flags, sub, len(sub), C.byref(m), C.byref(err))
            assert st == 0 and m.value in (0, 1), (st, pat, sub)
            if len(good) >= 3 and rng.random() < 0.2:
                grp = [good[-1], rng.choice(good), rng.choice(good)]
                pats = (C.c_char_p * 3)(*[g[0] for g in grp])
                lens = (C.c_size_t * 3)(*[len(g[0]) for g in grp])
                fl = (C.c_uint32 * 3)(*[g[1] for g in grp])
                mask, grouped = C.c_uint32(0), C.c_int32(0)
                st = L.tgx_regex_match_group(pats, lens, fl, 3, sub, len(sub), C.byref(mask), C.byref(grouped), C.byref(err))
                assert st == 0, (st, err.msg)
                for k, (gp, gf) in enumerate(grp):  # the product automaton agrees with each pattern alone
                    one = C.c_int32(-1)
                    assert L.tgx_regex_is_match(gp, len(gp), gf, sub, len(sub), C.byref(one), C.byref(err)) == 0
                    assert ((mask.value >> k) & 1) == one.value, (gp, sub, mask.value, grouped.value)
                grouped_runs += 1
        if len(good) > 200:
            del good[:100]
    print("fuzz_patterns: %d patterns (%d compiled, %d product-automaton checks), statuses %s" % (n, ok, grouped_runs, statuses))


if __name__ == "__main__":
    main()
