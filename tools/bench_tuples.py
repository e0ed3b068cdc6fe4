#!/usr/bin/env python3
"""COUNT(DISTINCT (a, b)) over two Int64 columns (multi-column uniqueness, constraints/uniqueness.rs:557-562) on one
MI355X: through the partitioned lists and, with TGX_FP_LISTS_MIN_ROWS raised out of reach, through the 128-bit table.

    python tools/bench_tuples.py [--rows N] [--steps K]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd._lib import spec

    n = args.rows // 64 * 64
    T.init()
    a = torch.arange(n, dtype=torch.int64, device="cuda") // 1000          # 1000 rows share a value of a ...
    b = torch.arange(n, dtype=torch.int64, device="cuda") % 1000           # ... and differ in b: every tuple is unique
    cols = [T.Column.int64(a, None, length=n), T.Column.int64(b, None, length=n)]
    for path, env in (("lists", None), ("128-bit table", str(1 << 62))):
        if env is None:
            os.environ.pop("TGX_FP_LISTS_MIN_ROWS", None)
        else:
            os.environ["TGX_FP_LISTS_MIN_ROWS"] = env
        plan = T.Plan([spec(T.DISTINCT, 0, columns=[0, 1])])
        st = T.State(plan)
        for it in range(args.steps + 2):
            if it == 2:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            st.reset()
            st.update(cols)
            res = st.finalize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps * 1e3
        assert res[0].distinct == n, res[0].distinct
        print("(Int64, Int64) tuples, %d rows, %s: %.2f ms/step = %.1f G rows/s" % (n, path, dt, n / dt / 1e6))
    os.environ.pop("TGX_FP_LISTS_MIN_ROWS", None)


if __name__ == "__main__":
    main()
