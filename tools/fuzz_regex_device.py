#!/usr/bin/env python3
"""The pattern kernels on the GPU against RE2 (pyarrow.compute): random valid patterns (tools/fuzz_regex_diff.py's
grammar), each counted by regex_match_kernel over a column of a few thousand random values -- short and long ones
(a value may span several staging steps), NULLs, as Utf8 on the device, as Utf8View, as a dictionary column; up to
four patterns per plan, so that they share ONE walk (product automaton) where their tables fit -- and by RE2 value by
value.  Counts must be equal.

    python tools/fuzz_regex_device.py [--seconds 120] [--seed 1]"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--rows", type=int, default=6000)
    args = ap.parse_args()
    import ctypes as C

    import numpy as np
    import pyarrow as pa
    import pyarrow.compute as pc

    import fuzz_regex_diff as F
    import term_amd as T
    from term_amd._lib import spec

    T.init()
    rng = random.Random(args.seed)
    t0 = time.time()
    plans = cmps = bad = 0
    err = T._lib._Error()
    while time.time() - t0 < args.seconds:
        ascii_only = rng.random() < 0.5
        pats = []
        while len(pats) < rng.randint(1, 4):
            p = F.pattern(rng, ascii_only)
            pb = p.encode()
            if T.lib().tgx_regex_validate(pb, len(pb), 0, C.byref(err)) == 0:
                pats.append(p)
        vals = []
        for _ in range(args.rows):
            r = rng.random()
            if r < 0.06:
                vals.append(None)
            elif r < 0.85:
                vals.append(F.subject(rng, ascii_only))
            else:  # long values: several staging steps, multi-byte characters across their borders
                vals.append("".join(F.subject(rng, ascii_only) for _ in range(rng.randint(5, 60))))
        layout = rng.choice(["utf8", "large", "view", "dict"])
        arr = pa.array(vals, pa.large_string() if layout == "large" else pa.string())
        if layout == "view":
            fed = arr.cast(pa.string_view())
        elif layout == "dict":
            fed = arr.dictionary_encode()
        else:
            fed = arr
        col = T.Column.from_arrow(fed)
        plan = T.Plan([spec(T.REGEX_MATCH, 0, pattern=p) for p in pats])
        st = T.State(plan)
        st.update([col])
        res = st.finalize()
        plans += 1
        for p, r in zip(pats, res):
            want = pc.sum(pc.match_substring_regex(arr, p)).as_py() or 0
            cmps += 1
            if r.matches != want or r.total != len(vals):
                bad += 1
                print("DISAGREE %s pattern %r: device %d RE2 %d (of %d rows, with %r)" % (layout, p, r.matches, want, len(vals), pats))
        del st, plan
    print("%d plans, %d pattern counts, %d disagreements, %.0f s" % (plans, cmps, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
