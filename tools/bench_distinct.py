#!/usr/bin/env python3
"""Times the exact COUNT(DISTINCT) paths alone.

    python tools/bench_distinct.py [--rows N] [--steps K] [--sparse-rows M]
(1) the range-partitioned bitmap on the bench table's two unique columns: the value range is declared up front
    (tgx_distinct_range_hint), so no scan runs -- the numbers are the partition + bucket-apply kernels only
    (tgx_profile_get("distinct"));
(2) keys WITHOUT a dense range (sparse Int64 ids, Float64 values), M rows: through the key lists (mixed keys
    partitioned twice, deduplicated in LDS) and, with TGX_FP_LISTS_MIN_ROWS raised out of reach, through the hash
    table (one memory-side atomic per key)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--sparse-rows", type=int, default=1_000_000_000)
    ap.add_argument("--ordered-only", action="store_true", help="only (1b): the dense pass over keys in order")
    ap.add_argument("--dense-only", action="store_true", help="only (1): the bench table's two unique columns")
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd import synth
    from term_amd._lib import spec

    n = (args.rows // 64) * 64
    T.init(distinct_capacity_hint=n)
    # ---- (1b) the dense pass over keys IN ORDER (ids that grow with the row number): a wave's rows share a bucket ----
    ids = torch.arange(n if not args.dense_only else 64, dtype=torch.int64, device="cuda") + 1000
    for name, specs in (() if args.dense_only else (("uniqueness alone", [spec(T.DISTINCT, 0)]),
                        ("uniqueness + min/max/mean", [spec(T.DISTINCT, 0), spec(T.NUMERIC_STATS, 0)]))):
        plan = T.Plan(specs)
        st = T.State(plan)
        col = T.Column.int64(ids, None, length=n)
        for it in range(args.steps + 2):
            if it == 2:
                st.profile_enable(True)
                st.profile_reset()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            st.reset()
            st.update([col])
            res = st.finalize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps * 1e3
        prof = st.profile_get("distinct")
        ok = res[0].distinct == n and (len(specs) == 1 or (res[1].min_i, res[1].max_i) == (1000, n + 999))
        print("ids in order, %d rows, %s: distinct=%d verified=%s  wall %.2f ms/step, kernels %.2f ms/step" %
              (n, name, res[0].distinct, ok, dt, prof["total_ms"] / args.steps))
    del ids
    if args.ordered_only:
        return
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
    del table
    if args.dense_only:
        return
    # ---- (2) sparse keys ----
    m = (args.sparse_rows // 64) * 64
    g = torch.Generator(device="cuda").manual_seed(3)
    ids = torch.randint(-2**62, 2**62, (m,), dtype=torch.int64, device="cuda", generator=g)
    vals = torch.randn(m, dtype=torch.float64, device="cuda", generator=g)
    for name, col in (("sparse Int64 ids", T.Column.int64(ids, None, length=m)),
                      ("Float64 values", T.Column.float64(vals, None, length=m))):
        for path, env in (("key lists", None), ("hash table", str(1 << 62))):
            if env is None:
                os.environ.pop("TGX_FP_LISTS_MIN_ROWS", None)
            else:
                os.environ["TGX_FP_LISTS_MIN_ROWS"] = env
            plan = T.Plan([spec(T.DISTINCT, 0)])
            st = T.State(plan)
            for it in range(args.steps + 2):
                if it == 2:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                st.reset()
                st.update([col])
                res = st.finalize()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps * 1e3
            print("%s, %d rows, %s: distinct=%d  wall %.2f ms/step = %.1f G rows/s" %
                  (name, m, path, res[0].distinct, dt, m / dt / 1e6))
            del st
    os.environ.pop("TGX_FP_LISTS_MIN_ROWS", None)
    # ---- (3) APPROX_DISTINCT: the HyperLogLog lane of the scan (2^14 registers), alone and next to min / max / mean ----
    for name, col in (("sparse Int64 ids", T.Column.int64(ids, None, length=m)),
                      ("Float64 values", T.Column.float64(vals, None, length=m))):
        for what, specs in (("lane alone", [spec(T.APPROX_DISTINCT, 0), spec(T.COUNT, 0)]),
                            ("lane + min/max/mean", [spec(T.APPROX_DISTINCT, 0), spec(T.NUMERIC_STATS, 0)])):
            st = T.State(T.Plan(specs))
            for it in range(args.steps + 2):
                if it == 2:
                    st.profile_enable(True)
                    st.profile_reset()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                st.reset()
                st.update([col])
                res = st.finalize()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps * 1e3
            k = st.profile_get("scan_hll")["total_ms"] / args.steps
            print("%s, %d rows, APPROX_DISTINCT (%s): estimate=%d  wall %.2f ms/step, kernels %.2f ms = %.2f TB/s" %
                  (name, m, what, res[0].distinct, dt, k, m * 8 / (k * 1e-3) / 1e12))
            del st


if __name__ == "__main__":
    main()
