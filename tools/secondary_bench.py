"""The other BASELINE.json configs on ONE MI355X, measured with bench.py's protocol and appended to its line as
`secondary` (round-4 verdict: only the headline was driver-observed).

  C2        null+range+unique suite, 100 M rows x 8 int64 / f64 columns
  C3_one    `FormatType::Email` (format.rs:237) on a 100 M-row Utf8 column of 28-byte addresses
  C3_three  `@` + `^[^@]+@[^@]+\\.[^@]+$` + `FormatType::Email` on the same column: one walk of the product automaton
  C4_1gpu   full suite + KLL(k = 200) on 4 f64 columns + Pearson on 2 pairs, 1 G rows x 16 columns (the headline's
            table; the 8-GPU form of C4 is `bench.py --gpus 8` with this plan's row-shard path); the Spearman pair of
            SURVEY 8d is timed separately (`spearman_ms_one_pair`)
  C5        64 columns = 24 Int64 + 24 Float64 + 16 Dictionary<Int32, Utf8> (1 k .. 1 M entries), all nullable,
            250 M rows, completeness x64 + min/max/mean x48 + COUNT(DISTINCT) and the e-mail format on the dictionary
            columns, ONE fused plan
  cold      `ValidationSuite::run` is one state per table (core/suite.rs:399): tgx_state_create -> tgx_update ->
            tgx_finalize -> tgx_state_destroy on a fresh state, for the headline table and for C2, beside the warm step
            (tgx_state_reset of a kept state)

A step = tgx_state_reset -> tgx_update -> tgx_finalize on device-resident columns; W warm-up steps, K timed, each timed
on the host clock around the call (tgx_finalize returns when the device is through); `ms_per_step` = the median.
`algorithmic_bytes` per SURVEY 8d (every column counted once); `frac_of_8TBs` = algorithmic bytes / median / 8 TB/s.
`verified` = closed-form facts of the synthetic data (and torch cross-checks of the dictionary columns).
"""
import os
import sys
import time

HBM_PEAK = 8.0e12


def _median(xs):
    return sorted(xs)[len(xs) // 2]


def timed_steps(torch, st, columns, steps, warmup):
    res = None
    for _ in range(warmup):
        st.reset()
        st.update(columns)
        res = st.finalize()
    torch.cuda.synchronize()
    ms = []
    for _ in range(steps):
        t0 = time.perf_counter()
        st.reset()
        st.update(columns)
        res = st.finalize()
        ms.append((time.perf_counter() - t0) * 1e3)
    return ms, res


def cold_steps(torch, T, plan, columns, n_states):
    """create -> update -> finalize -> destroy, `n_states` times; the first of them is reported apart (it may have to
    allocate; the later ones get their blocks from the library's cache)"""
    ms, res = [], None
    torch.cuda.synchronize()
    for _ in range(n_states):
        t0 = time.perf_counter()
        st = T.State(plan)
        st.update(columns)
        res = st.finalize()
        st.close()
        ms.append((time.perf_counter() - t0) * 1e3)
    return ms, res


def entry(ms, alg_bytes, rows, verified, **extra):
    med = _median(ms)
    out = {"ms_per_step": med, "ms_min": min(ms), "steps": len(ms), "rows": rows, "rows_per_s": rows / (med * 1e-3),
           "algorithmic_bytes": alg_bytes, "frac_of_8TBs": alg_bytes / (med * 1e-3) / HBM_PEAK, "verified": bool(verified)}
    out.update(extra)
    return out


def numeric_columns(T, layout, table, n):
    cols = []
    for (kind, _), (vals, validity) in zip(layout, table):
        ctor = T.Column.float64 if kind.startswith("f_") else T.Column.int64
        cols.append(ctor(vals, validity, length=n))
    return cols


def suite_specs(T, spec, layout, unique_cols):
    specs = []
    for ci in range(len(layout)):
        specs += [spec(T.COUNT, ci), spec(T.NUMERIC_STATS, ci)]
    specs += [spec(T.DISTINCT, ci) for ci in unique_cols]
    return specs


def verify_suite(T, synth, specs, res, layout, n):
    by = {}
    for s, r in zip(specs, res):
        by.setdefault((s.kind, s.column), r)
    ok = by[(T.DISTINCT, 0)].distinct == n and by[(T.NUMERIC_STATS, 0)].sum_i == n * (n - 1) // 2
    ok &= by[(T.NUMERIC_STATS, 0)].min_i == 0 and by[(T.NUMERIC_STATS, 0)].max_i == n - 1
    ok &= 0 < by[(T.DISTINCT, 1)].distinct <= max(1, n // 10)
    for ci, (_, has_validity) in enumerate(layout):
        c = by[(T.COUNT, ci)]
        ok &= c.total == n and (c.non_null == n if not has_validity else abs(c.non_null / n - (1 - synth.NULL_RATE)) < 1e-3)
    return bool(ok), by


def run_c2(T, torch, synth, spec, steps, warmup, seed):
    n = 100_000_000 // 64 * 64
    layout = synth.COLUMNS_16[:4] + synth.COLUMNS_16[8:12]
    table = synth.make_table(layout, 0, n, n, seed, "cuda")
    columns = numeric_columns(T, layout, table, n)
    specs = suite_specs(T, spec, layout, [0, 1])
    plan = T.Plan(specs)
    st = T.State(plan)
    ms, res = timed_steps(torch, st, columns, steps, warmup)
    ok, _ = verify_suite(T, synth, specs, res, layout, n)
    st.close()
    cold, cres = cold_steps(torch, T, plan, columns, 5)
    okc, _ = verify_suite(T, synth, specs, cres, layout, n)
    e = entry(ms, synth.algorithmic_bytes(layout, n), n, ok and okc, cols=len(layout), checks=len(specs),
              cold_step_ms=_median(cold[1:]), cold_first_state_ms=cold[0])
    del table, columns
    return e


def make_email_column(torch, n, device="cuda"):
    """'user%09d@example%03d.com' (28 bytes), 4 % without '@', 1 % NULL; LargeUtf8 offsets (2.8 GB of values)"""
    tmpl = torch.tensor(list(b"user000000000@example000.com"), dtype=torch.uint8, device=device)
    L = tmpl.numel()
    data = tmpl.repeat(n).view(n, L)
    rows = torch.arange(n, dtype=torch.int64, device=device)
    v = rows.clone()
    for pos in range(12, 3, -1):
        data[:, pos] = (48 + v % 10).to(torch.uint8)
        v //= 10
    d = rows % 1000
    for pos in range(23, 20, -1):
        data[:, pos] = (48 + d % 10).to(torch.uint8)
        d //= 10
    h = (rows * 2654435761) % 100
    data[h < 4, 13] = ord("#")
    valid = h != 99
    offsets = torch.arange(n + 1, dtype=torch.int64, device=device) * L
    pad = (-n) % 8
    bits = torch.cat([valid, torch.zeros(pad, dtype=torch.bool, device=device)]).view(-1, 8).to(torch.int32)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=device)
    validity = torch.cat([(bits * w).sum(dim=1).to(torch.uint8), torch.zeros(64, dtype=torch.uint8, device=device)])
    flat = torch.cat([data.view(-1), torch.zeros(64, dtype=torch.uint8, device=device)])
    expect = dict(n=n, nulls=int((~valid).sum()), with_at=int(((h >= 4) & valid).sum()))
    torch.cuda.synchronize()
    return offsets, flat, validity, L, expect


def run_c3(T, torch, spec, steps, warmup):
    from term_amd.csrc_patterns import EMAIL

    n = 100_000_000
    offsets, data, validity, L, expect = make_email_column(torch, n)
    col = T.Column(T.LARGE_UTF8, n, offsets=offsets, data=data, validity=validity)
    alg = n * (8 + L) + n // 8
    out = {}
    for name, pats in (("C3_one", [EMAIL]), ("C3_three", [r"@", r"^[^@]+@[^@]+\.[^@]+$", EMAIL])):
        plan = T.Plan([spec(T.REGEX_MATCH, 0, pattern=p) for p in pats])
        st = T.State(plan)
        ms, res = timed_steps(torch, st, [col], steps, warmup)
        ok = all(r.total == expect["n"] and r.matches == expect["with_at"] for r in res)
        out[name] = entry(ms, alg, n, ok, patterns=len(pats), value_bytes=L)
        st.close()
    # uniqueness of the same string column (COUNT(DISTINCT), constraints/uniqueness.rs:612-617): by keyed 128-bit
    # fingerprint, and as an EXACT key set (TGX_FLAG_EXACT_KEYS: equal fingerprints confirmed on the bytes) -- what the
    # host layer and the shim ask for by default; a name of <= 64 word characters (`^[\w.@+-]{1,64}$`: the automaton of
    # `^[\w.@+-]*$` and a character count) rides along as the pattern round 5 could not take
    for name, flags in (("strings_distinct_fingerprint", 0), ("strings_distinct_exact", T.FLAG_EXACT_KEYS)):
        plan = T.Plan([spec(T.DISTINCT, 0, flags=flags)])
        st = T.State(plan)
        ms, res = timed_steps(torch, st, [col], steps, warmup)
        out[name] = entry(ms, alg, n, res[0].distinct == expect["n"] - expect["nulls"], value_bytes=L)
        st.close()
    plan = T.Plan([spec(T.REGEX_MATCH, 0, pattern=r"^[\w.@+-]{1,64}$")])
    st = T.State(plan)
    ms, res = timed_steps(torch, st, [col], steps, warmup)
    out["C3_counted_class"] = entry(ms, alg, n, res[0].total == expect["n"] and res[0].matches == expect["with_at"],
                                    pattern=r"^[\w.@+-]{1,64}$", value_bytes=L)
    st.close()
    del col, offsets, data, validity
    return out


def run_c4(T, torch, synth, spec, layout, unique_cols, columns, n, steps, warmup):
    specs = suite_specs(T, spec, layout, unique_cols)
    f_cols = [ci for ci, (k, _) in enumerate(layout) if k.startswith("f_")]
    specs += [spec(T.KLL, ci, kll_k=200) for ci in f_cols[:4]]
    specs += [spec(T.COMOMENTS, a, column2=b) for a, b in zip(f_cols[0:4:2], f_cols[1:4:2])]
    plan = T.Plan(specs)
    st = T.State(plan)
    ms, res = timed_steps(torch, st, columns, steps, warmup)
    ok, by = verify_suite(T, synth, specs, res, layout, n)
    eps = 1.65 / 200 ** 0.5  # kll_sketch.rs:397-399
    ucol = next(ci for ci in f_cols[:4] if layout[ci][0] == "f_uniform")
    for si, s in enumerate(specs):
        if s.kind == T.KLL:
            ok &= res[si].kll_n == by[(T.COUNT, s.column)].non_null  # total weight == non-null rows exactly
            qs = [st.kll_quantile(si, q) for q in (0.5, 0.95, 0.99)]
            ok &= qs[0] <= qs[1] <= qs[2]
            if s.column == ucol:  # uniform on [0, 1000): the quantiles sit within the stated rank error
                lo, hi = by[(T.NUMERIC_STATS, ucol)].min_f, by[(T.NUMERIC_STATS, ucol)].max_f
                ok &= all(abs((v - lo) / (hi - lo) - q) < eps for q, v in zip((0.5, 0.95, 0.99), qs))
        if s.kind == T.COMOMENTS:
            ok &= res[si].non_null > 0
    st.close()
    # Spearman on one pair, timed separately: two rankings by the library's sample sort
    plan_s = T.Plan([spec(T.SPEARMAN, f_cols[0], column2=f_cols[1])])
    st_s = T.State(plan_s)
    sp_ms, rs = timed_steps(torch, st_s, columns, 2, 1)
    ok &= rs[0].non_null > 0
    st_s.close()
    return entry(ms, synth.algorithmic_bytes(layout, n), n, ok, cols=len(layout), checks=len(specs),
                 spearman_ms_one_pair=min(sp_ms))


def run_c5(T, torch, synth, spec, steps, warmup, n=250_000_000 // 64 * 64):
    import numpy as np
    from term_amd.csrc_patterns import EMAIL

    seed = 0x7E570005
    layout = [("i_wide", True)] * 24 + [("f_uniform", True)] * 12 + [("f_normal", True)] * 12
    table = synth.make_table(layout, 0, n, n, seed, "cuda")
    columns = numeric_columns(T, layout, table, n)
    cards = [1000, 1000, 4096, 10_000, 10_000, 50_000, 100_000, 100_000, 250_000, 250_000, 500_000, 500_000,
             1_000_000, 1_000_000, 1_000_000, 1_000_000]
    # entry e of every dictionary: an address, 1 in 16 of them not one (a dictionary of `card` entries is a prefix)
    entries = [("user%07d@example%03d.com" % (e, e % 997)) if e % 16 else ("not-an-email-%d" % e) for e in range(max(cards))]
    lens = np.fromiter((len(e) for e in entries), dtype=np.int64, count=len(entries))
    all_offs = np.zeros(len(entries) + 1, dtype=np.int64)
    all_offs[1:] = np.cumsum(lens)
    all_data = np.frombuffer("".join(entries).encode(), dtype=np.uint8)
    expect = []
    for k, card in enumerate(cards):
        ci = len(layout) + k
        offs = all_offs[: card + 1].astype(np.int32)
        data = all_data[: int(all_offs[card])]
        dcol = T.Column.utf8(torch.from_numpy(offs).cuda(),
                             torch.cat([torch.from_numpy(data.copy()), torch.zeros(64, dtype=torch.uint8)]).cuda())
        idx = (synth.gen_column("k_mod10", ci, 0, n, 10 * card, seed, "cuda")).to(torch.int32)  # uniform in [0, card)
        validity = synth.gen_validity(ci, 0, n, seed, "cuda")
        columns.append(T.Column.dict32_utf8(idx, dcol, validity=validity, length=n))
        expect.append((idx, validity, card))
    torch.cuda.synchronize()
    specs = [spec(T.COUNT, ci) for ci in range(len(columns))]
    specs += [spec(T.NUMERIC_STATS, ci) for ci in range(len(layout))]
    for k in range(len(cards)):
        ci = len(layout) + k
        specs += [spec(T.DISTINCT, ci), spec(T.REGEX_MATCH, ci, pattern=EMAIL, flags=T.FLAG_NULL_IS_VALID)]
    plan = T.Plan(specs)
    st = T.State(plan)
    ms, res = timed_steps(torch, st, columns, steps, warmup)
    by = {(s.kind, s.column): r for s, r in zip(specs, res)}
    ok = all(by[(T.COUNT, ci)].total == n for ci in range(len(columns)))
    # the dictionary columns against torch: distinct = referenced entries, matches = valid rows whose entry is an
    # address (e % 16 != 0) + NULL rows (null_is_valid is the format default, format.rs:376-384)
    for k in (0, 3, len(cards) - 1):
        idx, validity, card = expect[k]
        ci = len(layout) + k
        bits = validity[: (n + 7) // 8]
        valid = ((bits.view(-1, 1) >> torch.arange(8, device="cuda", dtype=torch.uint8)) & 1).view(-1)[:n].bool()
        seen = torch.zeros(card, dtype=torch.bool, device="cuda")
        seen[idx[valid].long()] = True
        got = (by[(T.DISTINCT, ci)].distinct, by[(T.COUNT, ci)].non_null, by[(T.REGEX_MATCH, ci)].matches)
        want = (int(seen.sum()), int(valid.sum()), int(((idx % 16 != 0) & valid).sum()) + int((~valid).sum()))
        if got != want:
            print("C5 mismatch on dictionary column %d (card %d): got %s want %s" % (ci, card, got, want), file=sys.stderr)
        ok &= got == want
        del valid, seen
    alg = synth.algorithmic_bytes(layout, n) + len(cards) * (4 * n + (n + 7) // 8) + 16 * 0
    alg += sum(int(all_offs[c]) + 4 * (c + 1) for c in cards)  # the dictionaries, once
    st.close()
    e = entry(ms, alg, n, ok, cols=len(columns), checks=len(specs))
    del table, columns, expect
    return e


def run_ingest(root):
    """8192-row batches through the C ABI from plain C (build/feed_batches, build/feed_strings: child processes): HOST and
    DEVICE buffers, numeric and string columns, beside the same rows as ONE batch.  rows/s; `of_one_batch` = the ratio."""
    import json
    import subprocess

    out = {}

    def lines(binary):
        path = os.path.join(root, "build", binary)
        if not os.path.exists(path):
            return []
        p = subprocess.run([path], capture_output=True, text=True, timeout=120)
        got = []
        for line in p.stdout.splitlines():
            try:
                got.append(json.loads(line))
            except ValueError:
                pass
        return got

    def fold(rows, key_of):
        one = {}
        for d in rows:  # (the one-batch line of a leg comes first)
            k = key_of(d)
            if d["updates"] == 1:
                one[k.replace(" (kept until finalize)", "").replace(", kept until finalize", "")] = d["rows_per_s"]
        for d in rows:
            if d["batch_rows"] != 8192:
                continue
            k = key_of(d)
            base = one.get(k.replace(" (kept until finalize)", "").replace(", kept until finalize", ""))
            out[k] = {"rows_per_s_8192_row_batches": d["rows_per_s"], "rows_per_s_one_batch": base,
                      "of_one_batch": (d["rows_per_s"] / base) if base else None, "verified": d.get("verified")}

    fold(lines("feed_batches"), lambda d: "%s, %s buffers" % (d["suite"], d["buffers"]))
    import re

    fold(lines("feed_strings"), lambda d: re.sub(r", \d+ rows.*$", "", d["workload"]))
    return out


def run_shard8_world1(root):
    """One rank's step of the 8-way shard of the headline table -- 125 M rows x 16 columns -- through the N > 1 code path
    (`bench.py --force-distributed`: facts round, bitmap-slice exchange over the library's RCCL communicator with the
    rank as its own peer, state all-gather, rank-ordered merge) in a child process.  What a rank of `--gpus 8` runs,
    minus what only exists with peers: the xGMI transfer and the other ranks' latencies.  8 x this step is the budget of
    the 8-GPU run: 23.7 / 6 = 3.95 ms for >= 6x."""
    import json
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-distributed", "--rows",
                        "125000000", "--steps", "20", "--warmup", "5", "--no-secondary", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": "rc %d: %s" % (p.returncode, (p.stderr or p.stdout)[-300:])}
    d = json.loads(lines[-1])
    kernels = d["roofline"]["launch_ms"] + d["config"]["distinct_ms_per_step"]
    return {"ms_per_step": d["ms_per_step"], "ms_min": d["ms_min"], "steps": d["steps"], "rows": d["config"]["rows_total"],
            "kernels_ms": kernels, "tail_ms": d["ms_per_step"] - kernels, "verified": d["config"]["verified"],
            "rank_breakdown": d["config"].get("rank_breakdown"),
            "how": "child process: bench.py --gpus 1 --force-distributed --rows 125000000 --steps 20 --warmup 5"}


def measure(T, torch, synth, spec, layout, unique_cols, headline_plan, table, columns, n, seed, steps=5, warmup=2,
            warm_headline_ms=None, log=None):
    """Everything above, in an order that fits one GPU's memory: the headline table's legs first, then the table is
    dropped and the smaller configs are generated one after the other.  `table` / `columns` are emptied in place."""
    out = {"protocol": "per config: %d warm-up + %d timed steps of reset -> update -> finalize on device-resident "
                       "columns, host clock per step, median; cold = create -> update -> finalize -> destroy" % (warmup, steps)}
    t_start = time.perf_counter()

    def note(name):
        if log:
            log("secondary: %s done at %.1f s" % (name, time.perf_counter() - t_start))

    cold, cres = cold_steps(torch, T, headline_plan, columns, 4)
    ok = cres[-2].distinct == n  # (the id column's DISTINCT is the second to last spec of build_suite)
    out["cold"] = {"headline_cold_step_ms": _median(cold[1:]), "headline_cold_first_state_ms": cold[0],
                   "headline_warm_step_ms": warm_headline_ms,
                   "ratio_to_warm": (_median(cold[1:]) / warm_headline_ms) if warm_headline_ms else None,
                   "verified": bool(ok), "cache": None}
    note("cold")
    out["C4_1gpu"] = run_c4(T, torch, synth, spec, layout, unique_cols, columns, n, steps, warmup)
    note("C4")
    del columns[:]
    del table[:]
    torch.cuda.empty_cache()
    T.init(distinct_capacity_hint=100_000_000)  # (the hint is per process: each config states its own rows)
    out["C2"] = run_c2(T, torch, synth, spec, max(steps, 10), warmup, seed)
    out["cold"]["C2_cold_step_ms"] = out["C2"].pop("cold_step_ms")
    out["cold"]["C2_cold_first_state_ms"] = out["C2"].pop("cold_first_state_ms")
    out["cold"]["C2_warm_step_ms"] = out["C2"]["ms_per_step"]
    out["cold"]["C2_ratio_to_warm"] = out["cold"]["C2_cold_step_ms"] / out["C2"]["ms_per_step"]
    note("C2")
    torch.cuda.empty_cache()
    out.update(run_c3(T, torch, spec, max(steps, 10), warmup))
    note("C3")
    torch.cuda.empty_cache()
    T.init(distinct_capacity_hint=1 << 20)  # (dictionaries of at most 1 M entries)
    T.trim()
    out["C5"] = run_c5(T, torch, synth, spec, steps, warmup)
    note("C5")
    torch.cuda.empty_cache()
    cs = T.cache_stats()
    out["cold"]["cache"] = {"device_hits": cs.device_hits, "device_misses": cs.device_misses,
                            "device_cached_bytes": cs.device_cached_bytes, "pinned_hits": cs.pinned_hits,
                            "pinned_misses": cs.pinned_misses}
    T.trim()
    try:
        out["shard8_world1"] = run_shard8_world1(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    except Exception as e:
        out["shard8_world1"] = {"error": str(e)}
    note("shard8_world1")
    try:
        out["ingest"] = run_ingest(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    except Exception as e:  # (a feeder that is not built, or stuck: the rest of the line stands)
        out["ingest"] = {"error": str(e)}
    note("ingest")
    out["seconds"] = time.perf_counter() - t_start
    return out


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import json

    import torch
    import term_amd as T
    from term_amd import synth
    from term_amd._lib import spec

    T.init(distinct_capacity_hint=1 << 20)
    print(json.dumps({"C2": run_c2(T, torch, synth, spec, 10, 2, 0x7E570004)}))
    print(json.dumps(run_c3(T, torch, spec, 10, 2)))
