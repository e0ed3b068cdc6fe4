#!/usr/bin/env python3
"""KLL sketch update alone: one Float64 column (5 % NULL) of --rows rows, k = 200.
    python tools/bench_kll.py [--rows 1000000000] [--steps 3]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd import synth
    from term_amd._lib import spec

    n = args.rows // 64 * 64
    T.init()
    (vals, validity), = synth.make_table([("f_uniform", True)], 0, n, n, 0x7E570004, "cuda")
    col = T.Column.float64(vals, validity, length=n)
    plan = T.Plan([spec(T.KLL, 0, kll_k=200), spec(T.NUMERIC_STATS, 0)])
    st = T.State(plan)
    for it in range(args.steps + 1):
        if it == 1:
            st.profile_enable(True)
            st.profile_reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        st.reset()
        st.update([col])
        res = st.finalize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    qs = [st.kll_quantile(0, q) for q in (0.01, 0.25, 0.5, 0.75, 0.95, 0.99)]
    lo, hi = res[1].min_f, res[1].max_f
    err = max(abs((v - lo) / (hi - lo) - q) for v, q in zip(qs, (0.01, 0.25, 0.5, 0.75, 0.95, 0.99)))
    print("rows %d: %.2f ms/step (kll kernels %.2f ms), kll_n %d == non_null %d: %s, max rank error on uniform data %.5f"
          % (n, dt * 1e3, st.profile_get("kll")["total_ms"] / args.steps, res[0].kll_n, res[1].non_null,
             res[0].kll_n == res[1].non_null, err))


if __name__ == "__main__":
    main()
