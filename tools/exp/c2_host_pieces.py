#!/usr/bin/env python3
"""Where the host's share of a C2 step goes: reset / update (Python marshalling, the library call) / finalize."""
import ctypes as C
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import term_amd as T
from term_amd import synth
from term_amd import _lib as L
from term_amd._lib import spec
from secondary_bench import numeric_columns, suite_specs

T.init(distinct_capacity_hint=1 << 20)
n = 100_000_000 // 64 * 64
layout = synth.COLUMNS_16[:4] + synth.COLUMNS_16[8:12]
table = synth.make_table(layout, 0, n, n, 0x7E570004, "cuda")
columns = numeric_columns(T, layout, table, n)
plan = T.Plan(suite_specs(T, spec, layout, [0, 1]))
st = T.State(plan)
for _ in range(5):
    st.reset(); st.update(columns); st.finalize()
arr = (L._Column * len(columns))(*[c.c for c in columns])
err = L._Error()
t = {k: [] for k in ("reset", "marshal", "update_call", "finalize", "step")}
for _ in range(40):
    t0 = time.perf_counter()
    st.reset()
    t1 = time.perf_counter()
    arr2 = (L._Column * len(columns))(*[c.c for c in columns])
    held = [c for c in columns if c is not None and (c.c.mem != 0 or (c.c.dictionary and c.c.dictionary.contents.mem != 0))]
    t2 = time.perf_counter()
    L.lib().tgx_update(plan.h, st.h, arr, len(columns), C.byref(err))
    t3 = time.perf_counter()
    st.finalize()
    t4 = time.perf_counter()
    for k, v in zip(("reset", "marshal", "update_call", "finalize", "step"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)):
        t[k].append(v * 1e6)
print({k: round(statistics.median(v), 1) for k, v in t.items()}, "us (medians of 40)")
