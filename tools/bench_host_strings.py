#!/usr/bin/env python3
"""A HOST string column streamed as DataFusion hands it out (8192-row batches, TG/core/context.rs:31) through a format
check: completeness + FormatType::Email + LENGTH bounds on 8 Mi e-mail addresses (~28 B each), held as Utf8, as
Utf8View (what DataFusion reads Parquet strings as: one data buffer per 64 Ki rows, shared by the batches cut from it) and
as Dictionary<Int32, Utf8> (131 072 entries, one dictionary for all batches).  Prints rows/s of the stream and of the
same column as ONE batch (the PCIe copy of the column's bytes is the floor of both).

    python tools/bench_host_strings.py [--layout utf8|view|dict|all]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--layout", default="all")
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd._lib import spec
    from term_amd.csrc_patterns import EMAIL as EMAIL_PATTERN

    n = 8192 * 1024
    rng = np.random.default_rng(5)
    # "user<k>@example<k%1000>.com": built with numpy (offsets int32, data uint8)
    users = rng.integers(0, 10**9, size=n)
    parts = [("user%d@example%d.com" % (int(u), int(u) % 1000)).encode() for u in users[:131_072]]
    reps = n // len(parts)
    lens = np.array([len(p) for p in parts], dtype=np.int64)
    data = np.frombuffer(b"".join(parts) * reps, dtype=np.uint8).copy()
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.tile(lens, reps), out=offsets[1:])
    offsets = offsets.astype(np.int32)
    data = np.concatenate([data, np.zeros(64, np.uint8)])
    T.init()
    plan = T.Plan([spec(T.COUNT, 0), spec(T.REGEX_MATCH, 0, pattern=EMAIL_PATTERN),
                   spec(T.LENGTH, 0, length_min=5, length_max=64)])
    # the same values as Utf8View: every value is long (> 12 bytes): {length, 4-byte prefix, buffer index, offset}
    rows_per_buf = 65536
    lens_all = np.tile(lens, reps).astype(np.int32)
    starts = offsets[:-1].astype(np.int64)
    views = np.zeros((n, 4), dtype=np.int32)
    views[:, 0] = lens_all
    pref = np.zeros((n, 4), dtype=np.uint8)
    for k in range(4):
        pref[:, k] = data[starts + k]
    views[:, 1] = pref.view(np.int32)[:, 0]
    buf_of_row = np.arange(n) // rows_per_buf
    views[:, 2] = buf_of_row.astype(np.int32)
    buf_start = starts[::rows_per_buf]
    views[:, 3] = (starts - buf_start[buf_of_row]).astype(np.int32)
    bufs = [np.concatenate([data[int(buf_start[b]): int(buf_start[b + 1]) if b + 1 < len(buf_start) else int(offsets[-1])],
                            np.zeros(16, np.uint8)]) for b in range(len(buf_start))]
    views_u8 = views.view(np.uint8).reshape(-1)
    # ... and as a dictionary column: the 131 072 distinct values, int32 indices
    d_off = np.zeros(len(parts) + 1, dtype=np.int32)
    np.cumsum(lens, out=d_off[1:])
    d_data = np.concatenate([np.frombuffer(b"".join(parts), dtype=np.uint8), np.zeros(64, np.uint8)])
    dictionary = T.Column(T.UTF8, len(parts), offsets=d_off, data=d_data, validity=None)
    indices = np.tile(np.arange(len(parts), dtype=np.int32), reps)

    def column(layout, lo, rows):
        if layout == "utf8":
            return T.Column(T.UTF8, rows, offsets=offsets, data=data, validity=None, offset=lo)
        if layout == "view":
            return T.Column.utf8_view(views_u8, bufs, validity=None, length=rows, offset=lo)
        return T.Column.dict32_utf8(indices, dictionary, validity=None, length=rows, offset=lo)

    row_bytes = {"utf8": (data.size + 4 * n) / n, "view": (data.size + 16 * n) / n, "dict": 4.0 + (d_data.size + d_off.size * 4) / n}
    names = {"utf8": "Utf8", "view": "Utf8View", "dict": "Dictionary<Int32, Utf8>"}
    for layout in (["utf8", "view", "dict"] if args.layout == "all" else [args.layout]):
        st = T.State(plan)
        bytes_per_row = row_bytes[layout]
        for batch_rows in (n, 65536, 8192):
            batches = [[column(layout, lo, min(batch_rows, n - lo))] for lo in range(0, n, batch_rows)]
            best = 1e9
            for rep in range(3):
                st.reset()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for cols in batches:
                    st.update(cols)
                res = st.finalize()
                best = min(best, time.perf_counter() - t0)
            print(json.dumps({"workload": "HOST %s column, %d rows x %.0f B, completeness + e-mail format + length" % (names[layout], n, bytes_per_row),
                              "batch_rows": batch_rows, "updates": len(batches), "total_ms": best * 1e3,
                              "us_per_update": best * 1e6 / len(batches), "rows_per_s": n / best,
                              "host_to_device_GBs": n * bytes_per_row / best / 1e9,
                              "coalesce_flushes": st.profile_get("coalesce")["launches"],
                              "matches": int(res[1].matches), "verified": int(res[1].matches) == n and int(res[2].matches) == n}))
        del st


if __name__ == "__main__":
    main()
