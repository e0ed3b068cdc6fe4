#!/usr/bin/env python3
"""Experiment: does the headline step get shorter when the numeric scan (14 columns, pure streaming reads) and the
uniqueness pass (2 key columns, LDS-bound phases) run SIDE BY SIDE on disjoint sets of CUs
(hipExtStreamCreateWithCUMask) instead of one after the other on one stream?

    python tools/exp_cu_mask.py [--rows N] [--steps K] [--scan-cus 96,128,160,192]

Two plans, two states, two streams; prints the sequential time (both on full-chip streams, back to back) and the
concurrent time per split."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def masked_stream(hip, cus, n_cu=256):
    """stream restricted to the CUs in `cus` (bit i of the mask = CU i in the runtime's numbering)"""
    words = (n_cu + 31) // 32
    mask = (C.c_uint32 * words)()
    for cu in cus:
        mask[cu // 32] |= 1 << (cu % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, mask)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--scan-cus", default="96,128,160,192")
    ap.add_argument("--interleave", type=int, default=1, help="1: the scan takes every k-th CU; 0: the first CUs")
    args = ap.parse_args()
    import torch
    import term_amd as T
    from term_amd import synth
    from term_amd._lib import spec

    n = (args.rows // 64) * 64
    T.init(distinct_capacity_hint=n)
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=C.RTLD_GLOBAL)
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    layout, unique_cols = synth.COLUMNS_16, synth.UNIQUE_COLUMNS_16
    scan_specs, key_specs = [], []
    for ci in range(len(layout)):
        into = key_specs if ci in unique_cols else scan_specs
        into.append(spec(T.COUNT, ci))
        into.append(spec(T.NUMERIC_STATS, ci))
    for ci in unique_cols:
        key_specs.append(spec(T.DISTINCT, ci))
    table = synth.make_table(layout, 0, n, n, 1234, "cuda")
    columns = []
    for (kind, _), (vals, validity) in zip(layout, table):
        ctor = T.Column.float64 if kind.startswith("f_") else T.Column.int64
        columns.append(ctor(vals, validity, length=n))
    torch.cuda.synchronize()
    plan_scan, plan_keys = T.Plan(scan_specs), T.Plan(key_specs)

    def run(stream_scan, stream_keys, concurrent):
        st_scan = T.State(plan_scan, stream=stream_scan)
        st_keys = T.State(plan_keys, stream=stream_keys)

        def step():
            st_scan.reset()
            st_keys.reset()
            if concurrent:
                st_keys.update(columns)   # (waits for its sample first, then queues the pass)
                st_scan.update(columns)
                a = st_scan.finalize()
                b = st_keys.finalize()
            else:
                st_scan.update(columns)
                a = st_scan.finalize()
                st_keys.update(columns)
                b = st_keys.finalize()
            return a, b

        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            a, b = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        assert b[-2].distinct == n, b[-2].distinct  # the bijective id column
        return dt * 1e3

    full_a, full_b = torch.cuda.Stream(), torch.cuda.Stream()
    print(json.dumps({"mode": "sequential, full chip", "ms_per_step": run(full_a.cuda_stream, full_b.cuda_stream, False)}))
    print(json.dumps({"mode": "concurrent, unmasked streams", "ms_per_step": run(full_a.cuda_stream, full_b.cuda_stream, True)}))
    for k in [int(x) for x in args.scan_cus.split(",")]:
        if args.interleave:
            # spread both sets over all XCDs: CU i belongs to the scan when (i * k) // 256 advances
            scan = [i for i in range(256) if ((i + 1) * k) // 256 != (i * k) // 256]
        else:
            scan = list(range(k))
        keys = [i for i in range(256) if i not in set(scan)]
        s_scan, s_keys = masked_stream(hip, scan), masked_stream(hip, keys)
        ms = run(s_scan.value, s_keys.value, True)
        print(json.dumps({"mode": "concurrent, CU masks", "scan_cus": len(scan), "key_cus": len(keys), "ms_per_step": ms}))


if __name__ == "__main__":
    main()
