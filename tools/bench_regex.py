#!/usr/bin/env python3
"""C3 of BASELINE.json: has_pattern / contains_email on a 100 M-row Utf8 column, 1 MI355X.
Prints one JSON line per pattern set: rows/s and algorithmic GB/s (8 B LargeUtf8 offset + value bytes + 1 bit per row).
    python tools/bench_regex.py [--rows 100000000] [--steps 5]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_column(torch, n, device="cuda"):
    """'user%09d@example%03d.com' (28 bytes), 4 % without '@', 1 % NULL"""
    tmpl = torch.tensor(list(b"user000000000@example000.com"), dtype=torch.uint8, device=device)
    L = tmpl.numel()
    data = tmpl.repeat(n).view(n, L)
    rows = torch.arange(n, dtype=torch.int64, device=device)
    v = rows.clone()
    for pos in range(12, 3, -1):  # 9 digits of the row id
        data[:, pos] = (48 + v % 10).to(torch.uint8)
        v //= 10
    d = rows % 1000
    for pos in range(23, 20, -1):
        data[:, pos] = (48 + d % 10).to(torch.uint8)
        d //= 10
    h = (rows * 2654435761) % 100
    data[h < 4, 13] = ord("#")
    valid = h != 99
    offsets = torch.arange(n + 1, dtype=torch.int64, device=device) * L  # 2.8 GB of values: LargeUtf8 offsets
    pad = (-n) % 8
    bits = torch.cat([valid, torch.zeros(pad, dtype=torch.bool, device=device)]).view(-1, 8).to(torch.int32)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=device)
    validity = torch.cat([(bits * w).sum(dim=1).to(torch.uint8), torch.zeros(64, dtype=torch.uint8, device=device)])
    flat = torch.cat([data.view(-1), torch.zeros(64, dtype=torch.uint8, device=device)])
    expect = dict(n=n, nulls=int((~valid).sum()), with_at=int(((h >= 4) & valid).sum()))
    return offsets, flat, validity, L, expect


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--view", action="store_true", help="hold the column as Utf8View (16-byte views + one data buffer)")
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd._lib import spec
    from term_amd.csrc_patterns import EMAIL

    T.init()
    offsets, data, validity, L, expect = make_column(torch, args.rows)
    col = T.Column(T.LARGE_UTF8, args.rows, offsets=offsets, data=data, validity=validity)
    alg_bytes = args.rows * (8 + L) + args.rows // 8
    if args.view:
        # Utf8View (DataFusion's default string layout since 43): {len, 4-byte prefix, buffer index, offset} per row
        n = args.rows
        views = torch.zeros(n, 4, dtype=torch.int32, device="cuda")
        views[:, 0] = L
        views[:, 1] = data[: n * L].view(n, L)[:, :4].contiguous().view(torch.int32).view(n)
        views[:, 3] = (torch.arange(n, dtype=torch.int64, device="cuda") * L).to(torch.int32)
        col = T.Column.utf8_view(views.view(torch.uint8).view(-1), [data], validity=validity, length=n)
        alg_bytes = args.rows * (16 + L) + args.rows // 8
    sets = {"contains '@'": [r"@"], "simple e-mail": [r"^[^@]+@[^@]+\.[^@]+$"], "FormatType::Email": [EMAIL],
            "all three": [r"@", r"^[^@]+@[^@]+\.[^@]+$", EMAIL]}
    for name, pats in sets.items():
        plan = T.Plan([spec(T.REGEX_MATCH, 0, pattern=p) for p in pats])
        st = T.State(plan)
        st.update([col])
        res = st.finalize()
        assert all(r.total == expect["n"] and r.matches == expect["with_at"] for r in res), [(r.total, r.matches) for r in res]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st.reset()
            st.update([col])
            st.finalize()
        dt = (time.perf_counter() - t0) / args.steps
        print(json.dumps({"workload": "C3 regex, %d rows x %d B" % (args.rows, L), "patterns": name,
                          "ms_per_step": dt * 1e3, "rows_per_s": args.rows / dt,
                          "algorithmic_GBs": alg_bytes / dt / 1e9, "frac_of_8TBs": alg_bytes / dt / 8e12}))


if __name__ == "__main__":
    main()
