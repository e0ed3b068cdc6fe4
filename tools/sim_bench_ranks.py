#!/usr/bin/env python3
"""bench.py's N-rank step at its REAL sizes on one GPU: N threads, each with the row shard of the 1 G-row table that
rank would hold (125 M rows x 16 columns at N = 8: 130 GB in all), the transport of tgx_allreduce replaced by the
thread-barrier stand-in of term_amd.distributed.thread_comm (device pointers, as RCCL gets them).  Checks the merged
results against the closed-form facts bench.py verifies.  Not a timing tool (the ranks share the GPU): it exercises
the exact buffer sizes and bitmap slices of the multi-GPU run (1 G-bit bitmaps, 16 MB slices) without an 8-GPU node.

    python tools/sim_bench_ranks.py [--ranks 8] [--rows 1000000000]"""
import argparse
import os
import sys
import threading

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd import synth
    from term_amd._lib import spec
    from term_amd.distributed import ThreadGroup, sharded_suite_step, thread_comm

    world = args.ranks
    layout, unique_cols = synth.COLUMNS_16, synth.UNIQUE_COLUMNS_16
    n_total = (args.rows // (64 * world)) * 64 * world
    n_local = n_total // world
    T.init(device_id=0, distinct_capacity_hint=n_local)
    specs = []
    for ci in range(len(layout)):
        specs += [spec(T.COUNT, ci), spec(T.NUMERIC_STATS, ci)]
    specs += [spec(T.DISTINCT, ci) for ci in unique_cols]
    plan = T.Plan(specs)
    group = ThreadGroup(world)
    results, errors = [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            table = synth.make_table(layout, rank * n_local, n_local, n_total, 0x7E570004, "cuda")
            cols = [(T.Column.float64 if k.startswith("f_") else T.Column.int64)(v, b, length=n_local)
                    for (k, _), (v, b) in zip(layout, table)]
            st = T.State(plan)
            comm = thread_comm(group, rank, device_buffers=True)
            for _ in range(args.steps):
                res = sharded_suite_step(plan, st, cols, comm)
            results[rank] = res
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            group.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise SystemExit("failed: %r" % errors)
    ok = True
    for rank in range(world):
        by_col = {}
        for s, r in zip(specs, results[rank]):
            by_col.setdefault(s.column, {})[s.kind] = r
        for ci, (kind, has_validity) in enumerate(layout):
            c, stt = by_col[ci][T.COUNT], by_col[ci][T.NUMERIC_STATS]
            ok &= c.total == n_total and stt.total == n_total and c.non_null == stt.non_null
            ok &= (c.non_null == n_total) if not has_validity else abs(c.non_null / n_total - (1 - synth.NULL_RATE)) < 1e-3
        d_id, d_k = by_col[0][T.DISTINCT], by_col[1][T.DISTINCT]
        ok &= d_id.distinct == n_total
        ok &= by_col[0][T.NUMERIC_STATS].min_i == 0 and by_col[0][T.NUMERIC_STATS].max_i == n_total - 1
        ok &= by_col[0][T.NUMERIC_STATS].sum_i == n_total * (n_total - 1) // 2
        ok &= 0 < d_k.distinct <= max(1, n_total // 10)
        ok &= [(r.total, r.non_null, r.distinct, r.min_i, r.max_i, r.sum_i) for r in results[rank]] == \
              [(r.total, r.non_null, r.distinct, r.min_i, r.max_i, r.sum_i) for r in results[0]]
    print({"ranks": world, "rows_total": n_total, "rows_per_rank": n_local, "distinct_id": results[0][-2].distinct,
           "distinct_k": results[0][-1].distinct, "verified": bool(ok)})
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
