#!/usr/bin/env python3
"""Times the exact COUNT(DISTINCT) path alone (range-partitioned bitmap) on the bench table's two unique columns.

    python tools/bench_distinct.py [--rows N] [--steps K]
The value range is declared up front (tgx_distinct_range_hint), so no scan runs: the numbers are the partition +
bucket-apply kernels only (tgx_profile_get("distinct"))."""
import argparse
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
    from term_amd import synth
    from term_amd._lib import spec

    n = (args.rows // 64) * 64
    T.init(distinct_capacity_hint=n)
    layout = synth.COLUMNS_16[:2]
    table = synth.make_table(layout, 0, n, n, 0x7E570004, "cuda")
    for ci, (vals, validity) in enumerate(table):
        col = T.Column.int64(vals, validity, length=n)
        lo, hi = int(vals.min().item()), int(vals.max().item())
        plan = T.Plan([spec(T.DISTINCT, 0)])
        st = T.State(plan)
        res = None
        for it in range(args.steps + 2):
            if it == 2:
                st.profile_enable(True)
                st.profile_reset()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            st.reset()
            st.distinct_range_hint(0, lo, hi)
            st.update([col])
            res = st.finalize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps * 1e3
        prof = st.profile_get("distinct")
        print("col %d (%s, validity=%s): distinct=%d  wall %.2f ms/step, kernels %.2f ms/step" %
              (ci, layout[ci][0], validity is not None, res[0].distinct, dt, prof["total_ms"] / args.steps))


if __name__ == "__main__":
    main()
