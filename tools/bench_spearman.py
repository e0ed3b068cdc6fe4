#!/usr/bin/env python3
"""Spearman rank correlation of one Float64 pair (analyzer a13) on one MI355X: python tools/bench_spearman.py [--rows N]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd._lib import spec

    T.init()
    n = args.rows
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    y = 0.5 * x + 0.5 * torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    cols = [T.Column.float64(x, None, length=n), T.Column.float64(y, None, length=n)]
    plan = T.Plan([spec(T.SPEARMAN, 0, column2=1)])
    st = T.State(plan)
    st.update(cols)
    st.finalize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st.reset()
        st.update(cols)
        st.finalize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"workload": "Spearman, %d rows, 1 pair" % n, "ms_per_step": dt * 1e3, "rows_per_s": n / dt}))


if __name__ == "__main__":
    main()
