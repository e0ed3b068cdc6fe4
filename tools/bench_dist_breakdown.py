#!/usr/bin/env python3
"""Stage-by-stage wall time of bench.py's distributed step at world size 1 over the library's own RCCL communicator
(python tools/bench_dist_breakdown.py [rows]).  Every stage is fenced with a device synchronize, so the sum exceeds
the pipelined step time; `allreduce` is the whole cross-rank step (tgx_allreduce: facts gather, bitmap re-base +
all-to-all + adoption, state gather, rank-ordered merge)."""
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
from term_amd.distributed import rccl_comm

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
specs += [spec(T.DISTINCT, ci) for ci in unique_cols]
plan = T.Plan(specs)
stream = torch.cuda.Stream()
st = T.State(plan, stream=stream.cuda_stream)
comm = rccl_comm(dist, 0, 1)
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
    stage("reset", st.reset)
    stage("update (scan + distinct)", lambda: st.update(columns))
    stage("allreduce", lambda: st.allreduce(comm))
    return stage("finalize", st.finalize)


for _ in range(3):
    step()
acc.clear()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    step()
tot = time.perf_counter() - t0
for k, v in acc.items():
    print("%-26s %7.3f ms" % (k, v / K * 1e3))
print("%-26s %7.3f ms" % ("sum (fenced)", tot / K * 1e3))
del comm
dist.destroy_process_group()
