#!/usr/bin/env python3
"""One table, one state per run (ValidationSuite::run's shape, core/suite.rs:399): create -> update -> finalize ->
destroy of the C2 suite (100 M rows x 8 columns) or, with --rows, of any size; prints per-state wall times beside the
warm step's (tgx_state_reset of a kept state).  tools/trace_cold_step.sh runs it under the profiler's HIP trace.

    python tools/cold_step.py [--rows 100000000] [--states 6]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--states", type=int, default=6)
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd import synth
    from term_amd._lib import spec

    n = args.rows // 64 * 64
    T.init(distinct_capacity_hint=n)
    layout = synth.COLUMNS_16[:4] + synth.COLUMNS_16[8:12]
    table = synth.make_table(layout, 0, n, n, 0x7E570004, "cuda")
    cols = [(T.Column.float64 if k.startswith("f_") else T.Column.int64)(v, b, length=n) for (k, _), (v, b) in zip(layout, table)]
    specs = []
    for ci in range(len(layout)):
        specs += [spec(T.COUNT, ci), spec(T.NUMERIC_STATS, ci)]
    specs += [spec(T.DISTINCT, 0), spec(T.DISTINCT, 1)]
    plan = T.Plan(specs)
    st = T.State(plan)
    warm = []
    for it in range(8):
        t0 = time.perf_counter()
        st.reset()
        st.update(cols)
        res = st.finalize()
        warm.append((time.perf_counter() - t0) * 1e3)
    st.close()
    cold = []
    for it in range(args.states):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = T.State(plan)
        t1 = time.perf_counter()
        st.update(cols)
        t2 = time.perf_counter()
        res = st.finalize()
        t3 = time.perf_counter()
        st.close()
        t4 = time.perf_counter()
        cold.append([(b - a) * 1e3 for a, b in ((t0, t4), (t0, t1), (t1, t2), (t2, t3), (t3, t4))])
    assert res[-2].distinct == n
    print("warm step (median of 8): %.3f ms" % sorted(warm)[4])
    for c in cold:
        print("cold state: %.3f ms = create %.3f + update %.3f + finalize %.3f + destroy %.3f" % tuple(c))


if __name__ == "__main__":
    main()
