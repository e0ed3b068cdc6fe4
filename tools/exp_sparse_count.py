#!/usr/bin/env python3
"""COUNT(DISTINCT) of M sparse Int64 ids through the key lists, alone (what tools/bench_distinct.py runs among other
things): a short program to put under rocprofv3 (tools/prof_any.sh, tools/pmc_any.sh).

    python tools/exp_sparse_count.py [--rows M] [--steps K]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd._lib import spec

    m = (args.rows // 64) * 64
    T.init(distinct_capacity_hint=1 << 20)
    g = torch.Generator(device="cuda").manual_seed(3)
    ids = torch.randint(-2**62, 2**62, (m,), dtype=torch.int64, device="cuda", generator=g)
    st = T.State(T.Plan([spec(T.DISTINCT, 0)]))
    col = T.Column.int64(ids, None, length=m)
    for it in range(args.steps + 2):
        if it == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        st.reset()
        st.update([col])
        res = st.finalize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps * 1e3
    print("rows %d: distinct=%d  wall %.2f ms/step = %.1f G rows/s" % (m, res[0].distinct, dt, m / dt / 1e6))


if __name__ == "__main__":
    main()
