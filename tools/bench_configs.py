#!/usr/bin/env python3
"""BASELINE.json configs C2, C4 and C5 on ONE MI355X (the headline bench.py covers the 1 G x 16 null+range+unique
suite); shapes as in SURVEY.md section 8d.

  C2  null+range+unique suite, 100 M rows x 8 int64/f64 columns
  C4  full suite + KLL(k=200) p50/p95/p99 on 4 f64 columns + Pearson on 2 pairs, 1 G rows x 16 columns; Spearman on
      1 pair timed separately (single-GPU leg: the 8-GPU run is bench.py's row-shard path with the same plan)
  C5  64 columns = 24 Int64 + 24 Float64 + 16 Dictionary<Int32, Utf8> (cardinality 1 k .. 1 M), all nullable,
      250 M rows, every applicable check in ONE fused plan: completeness x64, min/max/mean x48, and on the
      dictionary columns COUNT(DISTINCT) + the e-mail format pattern

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
        specs += [spec(T.KLL, ci, kll_k=200) for ci in f_cols[:4]]
        specs += [spec(T.COMOMENTS, a, column2=b) for a, b in zip(f_cols[0:4:2], f_cols[1:4:2])]
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
    spearman_ms = None
    if extra:
        # Spearman on one pair, timed separately (two device radix sorts + rank sums)
        plan_s = T.Plan([spec(T.SPEARMAN, f_cols[0], column2=f_cols[1])])
        st_s = T.State(plan_s)
        for it in range(2):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            st_s.reset()
            st_s.update(columns)
            rs = st_s.finalize()
            torch.cuda.synchronize()
            spearman_ms = (time.perf_counter() - t1) * 1e3
        ok &= rs[0].non_null > 0
        del st_s
    alg = synth.algorithmic_bytes(layout, n)
    print(json.dumps({"config": name, "spearman_ms_one_pair": spearman_ms, "rows": n, "cols": len(layout), "checks": len(specs), "ms_per_step": dt * 1e3,
                      "rows_per_s": n / dt, "suite_algorithmic_GBs": alg / dt / 1e9, "frac_of_8TBs": alg / dt / 8e12,
                      "kernel_ms": prof, "verified": bool(ok),
                      "quantiles_first_uniform_col": quant.get(next(iter(quant), None)) if quant else None}), flush=True)


def run_c5(T, torch, synth, spec, n, steps):
    from term_amd.csrc_patterns import EMAIL

    seed = 0x7E570005
    layout = [("i_wide", True)] * 24 + [("f_uniform", True)] * 12 + [("f_normal", True)] * 12
    table = synth.make_table(layout, 0, n, n, seed, "cuda")
    columns = []
    for (kind, _), (vals, validity) in zip(layout, table):
        ctor = T.Column.float64 if kind.startswith("f_") else T.Column.int64
        columns.append(ctor(vals, validity, length=n))
    cards = [1000, 1000, 4096, 10_000, 10_000, 50_000, 100_000, 100_000, 250_000, 250_000, 500_000, 500_000,
             1_000_000, 1_000_000, 1_000_000, 1_000_000]
    dict_cols, expect = [], []
    import numpy as np
    for k, card in enumerate(cards):
        ci = len(layout) + k
        # dictionary: `card` strings, 1 in 16 of them not an e-mail address
        entries = [("user%07d@example%03d.com" % (e, e % 997)) if e % 16 else ("not-an-email-%d" % e) for e in range(card)]
        data = np.frombuffer("".join(entries).encode(), dtype=np.uint8)
        offs = np.zeros(card + 1, dtype=np.int32)
        offs[1:] = np.cumsum([len(e) for e in entries])
        dcol = T.Column.utf8(torch.from_numpy(offs).cuda(), torch.cat([torch.from_numpy(data.copy()), torch.zeros(64, dtype=torch.uint8)]).cuda())
        idx = (synth.gen_column("k_mod10", ci, 0, n, 10 * card, seed, "cuda")).to(torch.int32)  # uniform in [0, card)
        validity = synth.gen_validity(ci, 0, n, seed, "cuda")
        dict_cols.append(T.Column.dict32_utf8(idx, dcol, validity=validity, length=n))
        expect.append((idx, validity, card))
    columns += dict_cols
    specs = []
    for ci in range(len(columns)):
        specs.append(spec(T.COUNT, ci))
    for ci in range(len(layout)):
        specs.append(spec(T.NUMERIC_STATS, ci))
    for k in range(len(cards)):
        ci = len(layout) + k
        specs += [spec(T.DISTINCT, ci), spec(T.REGEX_MATCH, ci, pattern=EMAIL, flags=T.FLAG_NULL_IS_VALID)]
    plan = T.Plan(specs)
    st = T.State(plan)
    torch.cuda.synchronize()
    for it in range(steps + 1):
        if it == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        st.reset()
        st.update(columns)
        res = st.finalize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # cross-check the dictionary columns with torch: distinct = referenced entries, matches = valid rows whose
    # entry is an address (e % 16 != 0) + NULL rows (null_is_valid is the format default)
    ok = True
    by = {}
    for s_, r in zip(specs, res):
        by[(s_.kind, s_.column)] = r
    for k, (idx, validity, card) in enumerate(expect[:4] + expect[-1:]):
        ci = len(layout) + (k if k < 4 else len(cards) - 1)
        bits = validity[: (n + 7) // 8]
        valid = ((bits.view(-1, 1) >> torch.arange(8, device="cuda", dtype=torch.uint8)) & 1).view(-1)[:n].bool()
        used = torch.unique(idx[valid])
        got = (by[(T.DISTINCT, ci)].distinct, by[(T.COUNT, ci)].non_null, by[(T.REGEX_MATCH, ci)].matches)
        want = (used.numel(), int(valid.sum()), int(((idx % 16 != 0) & valid).sum()) + int((~valid).sum()))
        if got != want:
            print("C5 mismatch on dictionary column %d (card %d): got %s want %s" % (ci, card, got, want), file=sys.stderr)
        ok &= got == want
        del valid, used
    alg = synth.algorithmic_bytes(layout, n) + len(cards) * (4 * n + (n + 7) // 8)
    print(json.dumps({"config": "C5 64-col mixed + dictionary strings, fused (1 GPU)", "rows": n, "cols": len(columns),
                      "checks": len(specs), "ms_per_step": dt * 1e3, "rows_per_s": n / dt,
                      "suite_algorithmic_GBs": alg / dt / 1e9, "frac_of_8TBs": alg / dt / 8e12, "verified": bool(ok)}),
          flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--only", default="")
    ap.add_argument("--c5-rows", type=int, default=250_000_000)
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
    if args.only in ("", "C5"):
        run_c5(T, torch, synth, spec, args.c5_rows // 64 * 64, args.steps)


if __name__ == "__main__":
    main()
