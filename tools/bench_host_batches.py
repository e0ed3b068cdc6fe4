#!/usr/bin/env python3
"""PCIe-inclusive rate: the 16-column null+range+unique suite fed from HOST Arrow buffers (TGX_MEM_HOST columns,
staged by tgx_update with hipMemcpyAsync), 4 M-row batches; pageable and pinned host memory.
    python tools/bench_host_batches.py [--rows 64000000]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=64_000_000)
    ap.add_argument("--batch", type=int, default=4_000_000)
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd import synth
    from term_amd._lib import spec

    n = args.rows // 64 * 64
    T.init(distinct_capacity_hint=n)
    layout, unique = synth.COLUMNS_16, synth.UNIQUE_COLUMNS_16
    table = synth.make_table(layout, 0, n, n, 0x7E570004, "cuda")
    specs = []
    for ci in range(len(layout)):
        specs += [spec(T.COUNT, ci), spec(T.NUMERIC_STATS, ci)]
    specs += [spec(T.DISTINCT, ci) for ci in unique]
    plan = T.Plan(specs)
    # device-resident reference result
    st = T.State(plan)
    st.update([(T.Column.float64 if k.startswith("f_") else T.Column.int64)(v, b, length=n) for (k, _), (v, b) in zip(layout, table)])
    want = [(r.total, r.non_null, r.distinct, r.sum_i) for r in st.finalize()]
    # pinned first: pinned buffers allocated AFTER the pageable copy of the table had been made and used measured
    # 11.7 GB/s; allocated first they give 51.5 GB/s, as the pageable pass does in either position, and single-column
    # probes give 52-55 GB/s for both kinds of memory -- where the pinned pages end up, not the staging path
    for pinned in (True, False):
        host = []
        for vals, validity in table:
            hv = vals.cpu()
            hb = None if validity is None else validity.cpu()
            if pinned:
                hv = hv.pin_memory()
                hb = None if hb is None else hb.pin_memory()
            host.append((hv.numpy(), None if hb is None else hb.numpy()))
        st = T.State(plan)
        bs = args.batch // 64 * 64
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r0 in range(0, n, bs):
            m = min(bs, n - r0)
            cols = []
            for (k, _), (hv, hb) in zip(layout, host):
                ctor = T.Column.float64 if k.startswith("f_") else T.Column.int64
                cols.append(ctor(hv, hb, length=m, offset=r0))  # a sliced Arrow array: same buffers, offset = r0
            st.update(cols)
        res = st.finalize()
        dt = time.perf_counter() - t0
        got = [(r.total, r.non_null, r.distinct, r.sum_i) for r in res]
        bytes_moved = synth.algorithmic_bytes(layout, n)
        print(json.dumps({"host_memory": "pinned" if pinned else "pageable", "rows": n, "batch_rows": bs,
                          "rows_per_s": n / dt, "host_to_device_GBs": bytes_moved / dt / 1e9,
                          "same_as_device_resident": got == want}), flush=True)
        del st


if __name__ == "__main__":
    main()
