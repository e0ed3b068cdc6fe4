#!/usr/bin/env python3
"""Per-batch overhead: the same table fed as DataFusion-sized RecordBatches (8192 rows, TG/core/context.rs:31)
vs one batch.  Prints microseconds per tgx_update call."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import term_amd as T
    from term_amd import synth
    from term_amd._lib import spec

    n = 8192 * 1024
    T.init(distinct_capacity_hint=n)
    layout = synth.COLUMNS_16[:8]
    table = synth.make_table(layout, 0, n, n, 1, "cuda")
    specs = []
    for ci in range(len(layout)):
        specs += [spec(T.COUNT, ci), spec(T.NUMERIC_STATS, ci)]
    specs_d = specs + [spec(T.DISTINCT, 0), spec(T.DISTINCT, 1)]
    host_table = [(v.cpu().numpy(), None if b is None else b.cpu().numpy()) for v, b in table]
    for name, sp, where in (("null+range x8", specs, "device"), ("null+range x8 + unique x2", specs_d, "device"),
                            ("null+range x8", specs, "host")):
        plan = T.Plan(sp)
        st = T.State(plan)
        src = table if where == "device" else host_table
        for batch_rows in (n, 65536, 8192):
            cols_per_batch = []
            for lo in range(0, n, batch_rows):
                cols = []
                for (kind, _), (vals, validity) in zip(layout, src):
                    ctor = T.Column.float64 if kind.startswith("f_") else T.Column.int64
                    cols.append(ctor(vals, validity, length=batch_rows, offset=lo))
                cols_per_batch.append(cols)
            st.reset()
            for cols in cols_per_batch[:4]:
                st.update(cols)
            st.finalize()
            # the first pass over a batch size may still grow the state's pinned arenas (a one-time cost per process:
            # round 4's "64 Ki cliff" from Python was this pass, timed); the steady figure is the best of the next three
            times = []
            for rep in range(4):
                st.reset()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for cols in cols_per_batch:
                    st.update(cols)
                res = st.finalize()
                times.append(time.perf_counter() - t0)
            dt = min(times[1:])
            print(json.dumps({"suite": name, "buffers": where, "rows": n, "batch_rows": batch_rows, "updates": len(cols_per_batch),
                              "total_ms": dt * 1e3, "first_pass_ms": times[0] * 1e3, "us_per_update": dt * 1e6 / len(cols_per_batch),
                              "rows_per_s": n / dt, "distinct0": res[-2].distinct if len(sp) > 16 else None}))


if __name__ == "__main__":
    main()
