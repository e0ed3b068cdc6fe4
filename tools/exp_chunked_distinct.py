#!/usr/bin/env python3
"""Experiment (round 5): does the dense uniqueness pass get cheaper when a key column is fed in row chunks small enough
for a chunk's bucket lists to stay in the 256 MiB Infinity Cache between the partition kernel that writes them and the
replay that reads them?  Same state, same column, chunk sizes from the whole column down to 4 M rows.

    python tools/exp_chunked_distinct.py [--rows 1000000000]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd import synth
    from term_amd._lib import spec

    n = args.rows // 64 * 64
    T.init(distinct_capacity_hint=n)
    layout = synth.COLUMNS_16[:2]
    table = synth.make_table(layout, 0, n, n, 0x7E570004, "cuda")
    for ci, (vals, validity) in enumerate(table):
        lo, hi = int(vals.min().item()), int(vals.max().item())
        plan = T.Plan([spec(T.DISTINCT, 0), spec(T.NUMERIC_STATS, 0)])
        st = T.State(plan)
        for chunk in (n, 1 << 28, 1 << 26, 1 << 25, 1 << 24, 1 << 23, 1 << 22):
            chunk = min(chunk, n) // 64 * 64
            cols = [T.Column.int64(vals, validity, length=min(chunk, n - lo_row), offset=lo_row) for lo_row in range(0, n, chunk)]
            best = None
            for it in range(4):
                st.reset()
                st.distinct_range_hint(0, lo, hi)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for c in cols:
                    st.update([c])
                res = st.finalize()
                dt = (time.perf_counter() - t0) * 1e3
                best = dt if best is None or dt < best else best
            print("col %d (%s): %4d chunks of %10d rows: %.2f ms  distinct=%d" % (ci, layout[ci][0], len(cols), chunk, best, res[0].distinct), flush=True)


if __name__ == "__main__":
    main()
