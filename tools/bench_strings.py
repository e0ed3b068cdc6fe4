#!/usr/bin/env python3
"""String checks other than patterns on the C3 column (100 M e-mail rows x 28 B, LargeUtf8, 1 % NULL), 1 MI355X:
exact COUNT(DISTINCT) (uniqueness of a string key), LENGTH bounds, and all of them with a pattern in one plan.

    python tools/bench_strings.py [--rows N] [--steps K]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd._lib import spec
    from bench_regex import make_column

    T.init(distinct_capacity_hint=args.rows)
    offsets, data, validity, L, expect = make_column(torch, args.rows)
    col = T.Column(T.LARGE_UTF8, args.rows, offsets=offsets, data=data, validity=validity)
    alg_bytes = args.rows * (8 + L) + args.rows // 8
    # the same values as a Utf8View column (16-byte views + one data buffer; 28-byte values are out of line)
    views = torch.zeros(args.rows, 4, dtype=torch.int32, device="cuda")
    views[:, 0] = L
    views[:, 1] = data[: args.rows * L].view(args.rows, L)[:, :4].contiguous().view(torch.int32).view(args.rows)
    views[:, 3] = (torch.arange(args.rows, dtype=torch.int64, device="cuda") * L).to(torch.int32)
    view_col = T.Column.utf8_view(views.view(torch.uint8).view(-1), [data], validity=validity, length=args.rows)
    sets = {"COUNT(DISTINCT)": [spec(T.DISTINCT, 0)],
            "COUNT(DISTINCT), exact key set (TGX_FLAG_EXACT_KEYS)": [spec(T.DISTINCT, 0, flags=T.FLAG_EXACT_KEYS)],
            "COUNT(DISTINCT), column held as Utf8View": [spec(T.DISTINCT, 0)],
            "COUNT(DISTINCT), exact key set, column held as Utf8View": [spec(T.DISTINCT, 0, flags=T.FLAG_EXACT_KEYS)],
            "LENGTH between 5 and 40": [spec(T.LENGTH, 0, length_min=5, length_max=40)],
            "DISTINCT + LENGTH + '@'": [spec(T.DISTINCT, 0), spec(T.LENGTH, 0, length_min=5, length_max=40),
                                        spec(T.REGEX_MATCH, 0, pattern="@")]}
    # the same through the TABLE (what small batches, HOST batches and every batch after the first take): the list path
    # switched off for these rows
    sets["COUNT(DISTINCT) through the table"] = [spec(T.DISTINCT, 0)]
    sets["COUNT(DISTINCT) through the table, exact key set (key store: 16 B + the key per distinct key)"] = \
        [spec(T.DISTINCT, 0, flags=T.FLAG_EXACT_KEYS)]
    plain = col
    for name, specs in sets.items():
        os.environ.pop("TGX_FP_LISTS_MIN_ROWS", None)
        if "through the table" in name:
            os.environ["TGX_FP_LISTS_MIN_ROWS"] = str(10**12)
        col = view_col if "Utf8View" in name else plain
        plan = T.Plan(specs)
        st = T.State(plan)
        st.update([col])
        res = st.finalize()
        if specs[0].kind == T.DISTINCT:
            # every non-NULL row is distinct except the 4 % whose '@' was replaced: still distinct strings
            assert res[0].distinct == expect["n"] - expect["nulls"], (res[0].distinct, expect)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st.reset()
            st.update([col])
            st.finalize()
        dt = (time.perf_counter() - t0) / args.steps
        print(json.dumps({"workload": "%d rows x %d B" % (args.rows, L), "checks": name, "ms_per_step": dt * 1e3,
                          "rows_per_s": args.rows / dt, "algorithmic_GBs": alg_bytes / dt / 1e9}))


if __name__ == "__main__":
    main()
