#!/usr/bin/env python3
"""Child process of tests/test_gpu_rccl.py: ONE rank over the library's RCCL transport -- `tgx_comm_create_rccl`
(dlopen of librccl, ncclCommInitRank), then `sharded_suite_step` = reset -> update -> `tgx_allreduce` (facts all-gather,
bitmap-slice all-to-all as grouped ncclSend / ncclRecv, state all-gather, rank-ordered merge) -> finalize: what
`bench.py --force-distributed` and every rank of `bench.py --gpus N` run.  Prints one JSON line with every result of the
sharded step beside the plain state's (reset -> update -> finalize, no communicator) on the same columns.

    python tests/rccl_world1_child.py --rows 8000000 [--suite headline|full]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def result_tuple(T, s, r):
    k = s.kind
    if k == T.COUNT:
        return ["count", r.total, r.non_null]
    if k == T.NUMERIC_STATS:
        return ["stats", r.total, r.non_null, r.min_i, r.max_i, r.sum_i, r.min_f.hex(), r.max_f.hex(), r.sum_f.hex(),
                r.mean.hex()]
    if k == T.DISTINCT:
        return ["distinct", r.total, r.non_null, r.distinct, r.groups_once]
    if k == T.COMOMENTS:
        return ["como", r.non_null, r.sum_x.hex(), r.sum_y.hex(), r.sum_x2.hex(), r.sum_y2.hex(), r.sum_xy.hex()]
    if k == T.KLL:
        return ["kll", r.kll_n]
    if k == T.REGEX_MATCH:
        return ["regex", r.total, r.matches]
    return ["other", r.total, r.non_null]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8_000_000)
    ap.add_argument("--suite", default="headline")
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")

    import torch
    import torch.distributed as dist

    import term_amd as T
    from term_amd import synth
    from term_amd._lib import spec
    from term_amd.distributed import rccl_comm, sharded_suite_step

    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    n = args.rows // 64 * 64
    T.init(device_id=0, distinct_capacity_hint=n)
    layout, unique_cols = synth.COLUMNS_16, synth.UNIQUE_COLUMNS_16
    specs = []
    for ci in range(len(layout)):
        specs += [spec(T.COUNT, ci), spec(T.NUMERIC_STATS, ci)]
    specs += [spec(T.DISTINCT, ci) for ci in unique_cols]
    if args.suite == "full":
        f_cols = [ci for ci, (k, _) in enumerate(layout) if k.startswith("f_")]
        specs += [spec(T.KLL, ci, kll_k=200) for ci in f_cols[:2]]
        specs += [spec(T.COMOMENTS, f_cols[0], column2=f_cols[1])]
        specs += [spec(T.DISTINCT, 2, flags=T.FLAG_MULTIPLICITY), spec(T.DISTINCT, 9)]  # a wide-range Int64 and a Float64 key set
    plan = T.Plan(specs)
    table = synth.make_table(layout, 0, n, n, 0x7E570004, "cuda")
    columns = []
    for (kind, _), (vals, validity) in zip(layout, table):
        ctor = T.Column.float64 if kind.startswith("f_") else T.Column.int64
        columns.append(ctor(vals, validity, length=n))
    torch.cuda.synchronize()

    plain = T.State(plan)
    plain.update(columns)
    want = plain.finalize()

    stream = torch.cuda.Stream()
    st = T.State(plan, stream=stream.cuda_stream)
    comm = rccl_comm(dist, 0, 1)
    got = None
    for _ in range(args.steps):   # (the second step runs on remembered ranges and settled blob capacities)
        got = sharded_suite_step(plan, st, columns, comm)
    torch.cuda.synchronize()
    out = {"rows": n, "specs": len(specs), "steps": args.steps,
           "sharded": [result_tuple(T, s, r) for s, r in zip(specs, got)],
           "plain": [result_tuple(T, s, r) for s, r in zip(specs, want)]}
    if args.suite == "full":
        out["kll_sharded"] = [st.kll_quantile(i, q) for i, s in enumerate(specs) if s.kind == T.KLL for q in (0.5, 0.95)]
        out["kll_plain"] = [plain.kll_quantile(i, q) for i, s in enumerate(specs) if s.kind == T.KLL for q in (0.5, 0.95)]
    del st, plain
    dist.barrier()
    del comm
    dist.destroy_process_group()
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
