#!/usr/bin/env python3
"""Spearman rank correlation of one Float64 pair (analyzer a13) on one MI355X:

    python tools/bench_spearman.py [--rows N] [--ranks W]

--ranks W > 1 also runs the pair row-sharded over W threaded ranks that share this GPU (tgx_allreduce: the distributed
ranking, transported by term_amd.distributed.thread_comm) and checks the five rank sums against the single state's --
the ranks take turns on the one device, so the time is the SUM of their work, not a multi-GPU time."""
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
    ap.add_argument("--ranks", type=int, default=1)
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
    torch.cuda.synchronize()  # (the state works on a stream of its own: the columns have to be complete)
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
    if args.ranks > 1:
        import threading
        from term_amd.distributed import ThreadGroup, shard_rows, sharded_suite_step, thread_comm

        want = st.finalize()[0]
        group, out, errors = ThreadGroup(args.ranks), [None] * args.ranks, []

        def worker(rank):
            try:
                torch.cuda.set_device(0)
                lo, hi = shard_rows(n, args.ranks, rank)
                shard = [c.sliced(lo, hi - lo) for c in cols]
                state = T.State(plan)
                comm = thread_comm(group, rank, device_buffers=True)
                sharded_suite_step(plan, state, shard, comm)
                group.barrier.wait()
                t1 = time.perf_counter()
                res = sharded_suite_step(plan, state, shard, comm)
                out[rank] = (res[0], time.perf_counter() - t1)
            except Exception:  # noqa: BLE001
                import traceback

                errors.append(traceback.format_exc())
                group.barrier.abort()

        threads = [threading.Thread(target=worker, args=(r,)) for r in range(args.ranks)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for r, _ in out:
            assert (r.total, r.non_null, r.sum_x, r.sum_y, r.sum_x2, r.sum_y2, r.sum_xy) == \
                (want.total, want.non_null, want.sum_x, want.sum_y, want.sum_x2, want.sum_y2, want.sum_xy)
        print(json.dumps({"workload": "Spearman, %d rows over %d threaded ranks on ONE GPU" % (n, args.ranks),
                          "ms_per_step_all_ranks_serialised": max(t for _, t in out) * 1e3, "equal_to_single_state": True}))


if __name__ == "__main__":
    main()
