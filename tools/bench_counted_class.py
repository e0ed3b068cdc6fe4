#!/usr/bin/env python3
"""A counted Unicode class between anchors on the C3 column (100 M x 28 B): `^[\\w.@+-]{1,64}$` -- the automaton of
`^[\\w.@+-]*$` (316 states: walked from L2, or from LDS with TGX_REGEX_LDS_ENTRIES=32768) + a character count.

    python tools/bench_counted_class.py [--rows N] [--steps K]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd._lib import spec
    from bench_regex import make_column

    T.init()
    offsets, data, validity, L, expect = make_column(torch, args.rows)
    col = T.Column(T.LARGE_UTF8, args.rows, offsets=offsets, data=data, validity=validity)
    for pat in (r"^[\w.@+-]{1,64}$", r"^[\w.@+-]*$", r"^\w{1,64}@"):
        plan = T.Plan([spec(T.REGEX_MATCH, 0, pattern=pat)])
        st = T.State(plan)
        st.update([col])
        res = st.finalize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st.reset()
            st.update([col])
            res = st.finalize()
        dt = (time.perf_counter() - t0) / args.steps
        print(json.dumps({"pattern": pat, "ms_per_step": dt * 1e3, "matches": res[0].matches, "total": res[0].total,
                          "lds_entries": os.environ.get("TGX_REGEX_LDS_ENTRIES", "16384")}))


if __name__ == "__main__":
    main()
