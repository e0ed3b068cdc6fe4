#!/usr/bin/env python3
"""BASELINE.json configs C2 and C4 on ONE MI355X (the headline bench.py covers the 1 G x 16 null+range+unique suite).

  C2  null+range+unique suite, 100 M rows x 8 int64/f64 columns
  C4  full suite + KLL p50/p95/p99 + correlation, 1 G rows x 16 columns (single-GPU leg: the 8-GPU run is
      bench.py's row-shard path with the same plan)

One JSON line per config: rows/s of the whole step (reset -> update -> finalize, device-resident batch), the
per-kernel-family times from tgx_profile_get, and closed-form / cross checks of the results.
    python tools/bench_configs.py [--steps 3] [--only C2|C4]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(name, T, torch, synth, spec, layout, unique_cols, n, steps, extra):
    table = synth.make_table(layout, 0, n, n, 0x7E570004, "cuda")
    columns = []
    for (kind, _), (vals, validity) in zip(layout, table):
        ctor = T.Column.float64 if kind.startswith("f_") else T.Column.int64
        columns.append(ctor(vals, validity, length=n))
    specs = []
    for ci in range(len(layout)):
        specs += [spec(T.COUNT, ci), spec(T.NUMERIC_STATS, ci)]
    specs += [spec(T.DISTINCT, ci) for ci in unique_cols]
    f_cols = [ci for ci, (k, _) in enumerate(layout) if k.startswith("f_")]
    if extra:
        specs += [spec(T.KLL, ci, kll_k=200) for ci in f_cols]
        specs += [spec(T.COMOMENTS, a, column2=b) for a, b in zip(f_cols[0::2], f_cols[1::2])]
    plan = T.Plan(specs)
    st = T.State(plan)
    torch.cuda.synchronize()
    res = None
    for it in range(steps + 2):
        if it == 2:
            st.profile_enable(True)
            st.profile_reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        st.reset()
        st.update(columns)
        res = st.finalize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = {k: st.profile_get(k)["total_ms"] / steps for k in ("scan", "distinct", "kll", "comoments")}
    ok = True
    by = {}
    for s, r in zip(specs, res):
        by.setdefault((s.kind, s.column), r)
    ok &= by[(T.DISTINCT, unique_cols[0])].distinct == n
    ok &= by[(T.NUMERIC_STATS, 0)].sum_i == n * (n - 1) // 2
    quant = {}
    if extra:
        for si, s in enumerate(specs):
            if s.kind == T.KLL:
                r = res[si]
                ok &= r.kll_n == by[(T.COUNT, s.column)].non_null  # total weight == non-null rows exactly
                qs = [st.kll_quantile(si, q) for q in (0.5, 0.95, 0.99)]
                ok &= qs[0] <= qs[1] <= qs[2]
                quant[s.column] = qs
        # a uniform(0,1) column: p50/p95/p99 must sit within the stated rank error of 0.5/0.95/0.99
        ucol = next(ci for ci in f_cols if layout[ci][0] == "f_uniform")
        lo, hi = by[(T.NUMERIC_STATS, ucol)].min_f, by[(T.NUMERIC_STATS, ucol)].max_f
        eps = 1.65 / 200 ** 0.5
        for q, v in zip((0.5, 0.95, 0.99), quant[ucol]):
            ok &= abs((v - lo) / (hi - lo) - q) < eps
    alg = synth.algorithmic_bytes(layout, n)
    print(json.dumps({"config": name, "rows": n, "cols": len(layout), "checks": len(specs), "ms_per_step": dt * 1e3,
                      "rows_per_s": n / dt, "suite_algorithmic_GBs": alg / dt / 1e9, "frac_of_8TBs": alg / dt / 8e12,
                      "kernel_ms": prof, "verified": bool(ok),
                      "quantiles_first_uniform_col": quant.get(next(iter(quant), None)) if quant else None}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd import synth
    from term_amd._lib import spec

    T.init(distinct_capacity_hint=1 << 20)
    if args.only in ("", "C2"):
        layout8 = synth.COLUMNS_16[:4] + synth.COLUMNS_16[8:12]
        run("C2 null+range+unique, 100M x 8", T, torch, synth, spec, layout8, [0, 1], 100_000_000 // 64 * 64, args.steps,
            extra=False)
    if args.only in ("", "C4"):
        run("C4 full suite + KLL + correlation, 1G x 16 (1 GPU)", T, torch, synth, spec, synth.COLUMNS_16,
            synth.UNIQUE_COLUMNS_16, 1_000_000_000 // 64 * 64, args.steps, extra=True)


if __name__ == "__main__":
    main()
