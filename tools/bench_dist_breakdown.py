#!/usr/bin/env python3
"""Stage-by-stage wall time of bench.py's distributed step at world size 1 (python tools/bench_dist_breakdown.py).
Every stage is fenced with a device synchronize, so the sum exceeds the pipelined step time."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

import term_amd as T
from term_amd import synth
from term_amd._lib import spec
from term_amd.distributed import agree_on_ranges, allgather_many, exchange_distinct_auto, merge_blobs

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29578")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
layout, unique_cols = synth.COLUMNS_16, synth.UNIQUE_COLUMNS_16
n = (rows // 64) * 64
T.init(device_id=0, distinct_capacity_hint=n)
specs = []
for ci in range(len(layout)):
    specs += [spec(T.COUNT, ci), spec(T.NUMERIC_STATS, ci)]
n_stats = len(specs)
specs += [spec(T.DISTINCT, ci) for ci in unique_cols]
plan, plan_d = T.Plan(specs[:n_stats]), T.Plan(specs[n_stats:])
stream = torch.cuda.Stream()
st, st_d = T.State(plan, stream=stream.cuda_stream), T.State(plan_d, stream=stream.cuda_stream)
table = synth.make_table(layout, 0, n, n, 0x7E570004, "cuda")
columns = [(T.Column.float64 if k.startswith("f_") else T.Column.int64)(v, b, length=n) for (k, _), (v, b) in zip(layout, table)]
torch.cuda.synchronize()
acc = {}


def stage(name, fn):
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return r


def step():
    stage("reset", lambda: (st.reset(), st_d.reset()))
    stage("scan update", lambda: st.update(columns))
    local = stage("local finalize", st.finalize)
    minmax = []
    for s in specs[n_stats:]:
        r = next(x for sp, x in zip(specs[:n_stats], local) if sp.kind == T.NUMERIC_STATS and sp.column == s.column)
        minmax.append((bool(r.has_value) and not r.is_float, r.min_i, r.max_i))
    rngs = stage("agree_on_ranges", lambda: agree_on_ranges(minmax, dist, 1))
    for j, rng in enumerate(rngs):
        if rng is not None:
            st_d.distinct_range_hint(j, rng[0], rng[1])
    stage("distinct update", lambda: st_d.update(columns))
    stage("exchange bitmaps", lambda: exchange_distinct_auto(st_d, list(range(len(specs) - n_stats)), dist, 1, 0))
    blobs = stage("serialize", lambda: [st.serialize(), st_d.serialize()])
    per_rank = stage("allgather", lambda: allgather_many(blobs, dist, 1, device="cuda", cache_key="bench"))
    merged = stage("merge", lambda: (merge_blobs(plan, [p[0] for p in per_rank]), merge_blobs(plan_d, [p[1] for p in per_rank])))
    return stage("finalize", lambda: merged[0].finalize() + merged[1].finalize())


for _ in range(3):
    step()
acc.clear()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    step()
tot = time.perf_counter() - t0
for k, v in acc.items():
    print("%-18s %7.3f ms" % (k, v / K * 1e3))
print("%-18s %7.3f ms" % ("sum (fenced)", tot / K * 1e3))
dist.destroy_process_group()
