#!/usr/bin/env python3
"""Device memory must be flat across reset/update/finalize cycles (buffers are reused, nothing is reallocated)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import term_amd as T
from term_amd import synth
from term_amd._lib import spec

n = 8_000_000
T.init(distinct_capacity_hint=n)
layout = synth.COLUMNS_16[:4] + synth.COLUMNS_16[8:10]
table = synth.make_table(layout, 0, n, n, 7, "cuda")
cols = [(T.Column.float64 if k.startswith("f_") else T.Column.int64)(v, b, length=n) for (k, _), (v, b) in zip(layout, table)]
specs = []
for ci in range(len(layout)):
    specs += [spec(T.COUNT, ci), spec(T.NUMERIC_STATS, ci, flags=T.FLAG_VARIANCE)]
specs += [spec(T.DISTINCT, 0), spec(T.DISTINCT, 1, flags=T.FLAG_MULTIPLICITY), spec(T.DISTINCT, 4), spec(T.KLL, 4, kll_k=200),
          spec(T.COMOMENTS, 4, column2=5), spec(T.DISTINCT, 0, columns=[1, 2])]
plan = T.Plan(specs)
st = T.State(plan)
used = []
for it in range(40):
    st.reset()
    st.update(cols)
    res = st.finalize()
    blob = st.serialize()
    other = T.State.deserialize(plan, blob)
    other.merge([st])
    other.finalize()
    del other
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    used.append(total - free)
print("device memory in use after iteration 5 / 20 / 40: %.1f / %.1f / %.1f MiB" % (used[4] / 2**20, used[19] / 2**20, used[39] / 2**20))
assert used[39] - used[4] < 64 << 20, "device memory grows across steps"
print("ok")
