"""CPU: random valid patterns (the grammar of tools/fuzz_regex_diff.py: inside what Rust's regex and RE2 agree on)
against random subjects, decided three ways -- the product's compiler + automaton (tgx_regex_is_match), the oracle's VM
(oracle/regex_oracle.c) and RE2 (pyarrow's match_substring_regex, an engine neither shares a line with).  A seeded
slice of the fuzzer; the long runs are in profiles/ (r05_fuzz_regex_diff.txt)."""
import ctypes as C
import os
import random
import sys

import pytest

import oracle_binding as orc
import term_amd as T

pa = pytest.importorskip("pyarrow")
pc = pytest.importorskip("pyarrow.compute")

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


@pytest.mark.parametrize("ascii_only,seed", [(False, 101), (True, 202)])
def test_product_oracle_and_re2_agree(ascii_only, seed):
    import fuzz_regex_diff as F

    rng = random.Random(seed)
    err = T._lib._Error()
    compared = 0
    for _ in range(120):
        pat = F.pattern(rng, ascii_only)
        subs = [F.subject(rng, ascii_only) for _ in range(25)]
        want = pc.match_substring_regex(pa.array(subs, pa.large_string()), pat).to_pylist()
        pb = pat.encode()
        if T.lib().tgx_regex_validate(pb, len(pb), 0, C.byref(err)) != 0:
            assert b"DFA states" in err.msg, (pat, err.msg)  # (too big for the device's table is the only refusal)
            continue
        rx = orc.Regex(pat)
        for s, w in zip(subs, want):
            sb = s.encode()
            m = C.c_int32()
            assert T.lib().tgx_regex_is_match(pb, len(pb), 0, sb, len(sb), C.byref(m), C.byref(err)) == 0, (pat, s)
            assert (bool(m.value), rx.is_match(s)) == (w, w), (pat, s, bool(m.value), rx.is_match(s), w)
            compared += 1
    assert compared > 2000
