#!/usr/bin/env python3
"""Cost of the 32-bit numeric types (widened on the device) next to their 64-bit twins: min / max / sum / count of one
column, 1 G rows, device-resident.   python tools/bench_numeric32.py [--rows N]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd._lib import spec

    T.init()
    n = args.rows // 64 * 64
    plan = T.Plan([spec(T.COUNT, 0), spec(T.NUMERIC_STATS, 0)])
    for name, dtype, ctor in (("Int64", torch.int64, T.Column.int64), ("Int32", torch.int32, T.Column.int32),
                              ("Float64", torch.float64, T.Column.float64), ("Float32", torch.float32, T.Column.float32)):
        x = (torch.arange(n, dtype=torch.int32, device="cuda") % 1000003).to(dtype)
        col = ctor(x, None, length=n)
        st = T.State(plan)
        st.update([col])
        st.finalize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st.reset()
            st.update([col])
            st.finalize()
        dt = (time.perf_counter() - t0) / args.steps
        print(json.dumps({"type": name, "rows": n, "ms_per_step": dt * 1e3, "rows_per_s": n / dt,
                          "column_GBs": n * x.element_size() / dt / 1e9}))
        del st, col, x


if __name__ == "__main__":
    main()
