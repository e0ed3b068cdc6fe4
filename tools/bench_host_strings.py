#!/usr/bin/env python3
"""A HOST Utf8 column streamed as DataFusion hands it out (8192-row batches, TG/core/context.rs:31) through a format
check: completeness + FormatType::Email + LENGTH bounds on 8 Mi e-mail addresses (~28 B each).  Prints rows/s of the
stream and of the same column as ONE batch (the PCIe copy of offsets + bytes is the floor of both)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
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
    st = T.State(plan)
    bytes_per_row = (data.size + 4 * n) / n
    for batch_rows in (n, 65536, 8192):
        batches = [[T.Column(T.UTF8, min(batch_rows, n - lo), offsets=offsets, data=data, validity=None, offset=lo)]
                   for lo in range(0, n, batch_rows)]
        best = 1e9
        for rep in range(3):
            st.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for cols in batches:
                st.update(cols)
            res = st.finalize()
            best = min(best, time.perf_counter() - t0)
        print(json.dumps({"workload": "HOST Utf8 column, %d rows x %.0f B, completeness + e-mail format + length" % (n, bytes_per_row),
                          "batch_rows": batch_rows, "updates": len(batches), "total_ms": best * 1e3,
                          "us_per_update": best * 1e6 / len(batches), "rows_per_s": n / best,
                          "host_to_device_GBs": n * bytes_per_row / best / 1e9,
                          "matches": int(res[1].matches), "verified": int(res[1].matches) == n and int(res[2].matches) == n}))


if __name__ == "__main__":
    main()
