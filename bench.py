#!/usr/bin/env python3
"""bench.py -- validated rows/s of the fused check suite on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (tgx_state_reset -> tgx_update -> [tgx_allreduce] -> tgx_finalize)
over one synthetic, device-resident batch of the 16-column "null + range + unique" table
(term_amd/synth.py):
    Completeness x16 + Min/Max/Mean x16 + FullUniqueness on 2 columns.
Every column crosses HBM once: 14 columns in scan_kernel (the `roofline` kernel), the two key columns in their
uniqueness pass, which takes their COUNT / MIN / MAX / SUM along (partition_kernel + bucket_apply_kernel).
Rows are sharded by row range across ranks (strong scaling: the table size is fixed); every rank runs the SAME fused
plan on its shard, then tgx_allreduce (C ABI, RCCL over xGMI): the ranks agree on the unique columns' value ranges,
exact distinct swaps slices of the range bitmaps (re-based on the agreed range on the fly) with one all-to-all --
16-byte key records by hash owner where a key set is not a bitmap -- and the packed partial states are all-gathered
and merged in rank order on every rank.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (bare: starts the N ranks itself, as children, and relays rank 0's line;
                                         refuses -- non-zero exit -- when the machine has fewer than N GPUs)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming ceiling)


def build_suite(T, spec, layout, unique_cols):
    specs = []
    for ci in range(len(layout)):
        specs.append(spec(T.COUNT, ci))           # completeness
        specs.append(spec(T.NUMERIC_STATS, ci))   # has_min / has_max / has_mean
    for ci in unique_cols:
        specs.append(spec(T.DISTINCT, ci))        # validates_uniqueness (FullUniqueness)
    return specs


def reference_cpu_baseline(sample_rows, seed):
    """term-guard itself on the host cores: baselines/term_guard_cpu is a Rust main that puts the
    same synthetic table into a DataFusion MemTable (target_partitions = nproc) and runs the same suite through
    ValidationSuite::run.  It needs cargo, the reference's sources and an OFFLINE crate registry (datafusion 50.3,
    arrow 56.2, ...): probed here, absent in the builder's image and on its GPU boxes -- then (None, why) comes back
    and the caller reports the oracle's restatement instead (`kind: "port"`)."""
    import shutil
    import subprocess

    crate = os.path.join(ROOT, "baselines", "term_guard_cpu")
    cargo = shutil.which("cargo")
    if not cargo:
        return None, "no cargo on PATH"
    ref = os.environ.get("TGX_TERM_GUARD_PATH", "/root/reference/term-guard")
    if not os.path.exists(os.path.join(ref, "Cargo.toml")):
        return None, "term-guard sources not found (TGX_TERM_GUARD_PATH)"
    home = os.environ.get("TGX_CARGO_HOME", os.environ.get("CARGO_HOME", os.path.expanduser("~/.cargo")))
    if not (os.path.isdir(os.path.join(crate, "vendor")) or os.path.isdir(os.path.join(home, "registry", "cache"))):
        return None, "no offline crate registry (vendor/ or $CARGO_HOME/registry)"
    env = dict(os.environ, CARGO_HOME=home)
    cmd = [cargo, "build", "--release", "--offline", "--manifest-path", os.path.join(crate, "Cargo.toml")]
    if ref != "/root/reference/term-guard":
        cmd += ["--config", 'patch.crates-io.term-guard.path="%s"' % ref]
    try:
        b = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800)
        if b.returncode != 0:
            return None, "cargo build failed: " + b.stderr.strip().splitlines()[-1][:200]
        exe = os.path.join(crate, "target", "release", "term_guard_cpu")
        r = subprocess.run([exe, str(sample_rows), str(seed)], capture_output=True, text=True, timeout=1800)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        got = json.loads(line)
    except (OSError, subprocess.TimeoutExpired, IndexError, ValueError) as e:
        return None, "term_guard_cpu did not run: %r" % (e,)
    return {"value": got["rows_per_s"], "unit": "rows/s", "cores": got["cores"], "kind": "reference",
            "verified": bool(got.get("total_checks")),
            "sample": "term-guard's ValidationSuite::run over a MemTable of the first %d rows of the same table "
                      "(%d checks, %.2f s), baselines/term_guard_cpu" % (got["rows"], got["total_checks"], got["seconds"])}, None


def cpu_baseline(torch, layout, unique_cols, table, sample_rows):
    """The proxy CPU baseline of SURVEY.md section 8d: the oracle's restatement of the reference semantics (NOT
    term-guard itself, which cannot be built here: no cargo / crates) on ALL host cores -- row-range partitions,
    one pass per constraint, hash-set COUNT(DISTINCT) re-partitioned by owner, merge in partition order
    (oracle/suite_mt.c) -- over the first `sample_rows` rows of the same table; the single-thread figure of the same
    code is reported beside it (on a quarter of the sample)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_binding as orc

    host = []
    for vals, validity in table:
        v = vals[:sample_rows].cpu().numpy()
        b = None if validity is None else validity[: (sample_rows + 7) // 8 + 8].cpu().numpy()
        host.append((np.ascontiguousarray(v), b))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    best, runs = None, 0
    t_all = time.perf_counter()
    while runs < 5 and (runs == 0 or time.perf_counter() - t_all < 8.0):  # a many-core host finishes in < 1 s: repeat
        t0 = time.perf_counter()
        counts, stats, dist = orc.suite_mt(host, list(unique_cols), sample_rows, cores)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        runs += 1
    ok = all(c.total == sample_rows for c in counts) and dist[0].distinct == dist[0].non_null  # ids are unique
    one_rows = max(1 << 20, sample_rows // 4)
    t0 = time.perf_counter()
    orc.suite_mt(host, list(unique_cols), one_rows, 1)
    dt1 = time.perf_counter() - t0
    return {"value": sample_rows / best, "unit": "rows/s", "cores": cores, "kind": "port",
            "single_thread_value": one_rows / dt1, "verified": bool(ok),
            "sample": "first %d rows x %d cols of the same table, oracle/suite_mt.c on %d threads (best of %d runs, "
                      "%.2f s each); single thread on the first %d rows (%.1f s)"
                      % (sample_rows, len(layout), cores, runs, best, one_rows, dt1)}


def launcher_command(n_gpus, port, script_args):
    """argv of the one-rank-per-GPU launch the driver's contract names (torch.distributed.run, one node)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(script_args)


def launcher_env(base=None):
    """environment of the launched ranks: rendezvous on 127.0.0.1 (the container's hostname may not resolve) and
    dmabuf IPC (the host driver supports nothing else: RCCL fails with hipIpcGetMemHandle otherwise)"""
    env = dict(os.environ if base is None else base)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK"):
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def free_port():
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def count_gpus(base="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs of this machine WITHOUT touching any of them: the KFD topology (nodes with SIMDs are GPUs; the CPUs' nodes
    have none), narrowed by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES when they list devices; torch's own count only
    where /sys has no topology (it does not initialise a device on this image either)"""
    have = None
    try:
        have = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                have += 1
    except (OSError, ValueError):
        have = None
    if have is None:
        import torch

        return torch.cuda.device_count()
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            have = min(have, len([x for x in v.split(",") if x.strip() != ""]))
    return have


def launch_ranks(args, script_args):
    """`python bench.py --gpus N` without a launcher around it: N fresh worker processes, one per GPU, started as
    CHILDREN (never an exec: this process stays the parent and only relays) before anything here has touched the GPU.
    Rank 0's JSON line is passed on as this process's own single line; any worker failing fails the run."""
    import subprocess

    have = count_gpus()
    if have < args.gpus:
        print("bench.py: --gpus %d but this machine exposes %d GPU(s): refusing to run fewer ranks than asked for "
              "(a 1-GPU number must never be recorded as an %d-GPU one)" % (args.gpus, have, args.gpus),
              file=sys.stderr, flush=True)
        return 2
    cmd = launcher_command(args.gpus, free_port(), script_args)
    proc = subprocess.run(cmd, env=launcher_env(), stdout=subprocess.PIPE, stderr=None, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    json_line = None
    for ln in reversed(lines):
        if ln.lstrip().startswith("{") and '"metric"' in ln:
            json_line = ln
            break
    for ln in lines:
        if ln is not json_line:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or json_line is None:
        print("bench.py: the %d-rank launch failed (exit code %d%s)" %
              (args.gpus, proc.returncode, "" if json_line else ", no result line"), file=sys.stderr, flush=True)
        return proc.returncode or 3
    if json.loads(json_line).get("n_gpus") != args.gpus:
        print("bench.py: the result line reports n_gpus=%r for --gpus %d" % (json.loads(json_line).get("n_gpus"), args.gpus),
              file=sys.stderr, flush=True)
        return 4
    sys.stdout.flush()
    print(json_line, flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=1_000_000_000, help="total rows of the table (all ranks)")
    ap.add_argument("--seed", type=int, default=0x7E570004)
    ap.add_argument("--cpu-sample-rows", type=int, default=1 << 26)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--secondary", action="store_true", help="measure the secondary configs at any --rows")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the other BASELINE.json configs (C2, C3, C4 on one GPU, C5) and the cold-step measurement "
                         "that follow the headline's timed region at N = 1")
    ap.add_argument("--force-distributed", action="store_true",
                    help="run the N>1 code path (range agreement, bitmap-slice exchange, all-gather merge) even "
                         "with one rank: a self-test of the multi-GPU step on a 1-GPU box")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    # Every rank, however it was launched (by launch_ranks below or by `python -m torch.distributed.run ... bench.py`),
    # before its first GPU call: the host driver of this pool only supports dmabuf IPC, and with the legacy mode RCCL /
    # device-memory sharing across processes fails with `hipIpcGetMemHandle: invalid argument` (stated by the pool's
    # environment notes; the value in force is reported in the result line's config)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            # invoked bare with --gpus N: start the N ranks ourselves (before anything here touches the GPU)
            raise SystemExit(launch_ranks(args, sys.argv[1:]))
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # never a line whose n_gpus disagrees with --gpus
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU "
                         "(python -m torch.distributed.run --nproc-per-node %d ... bench.py --gpus %d), or run "
                         "`python bench.py --gpus %d` bare and let it start the ranks" %
                         (args.gpus, world, args.gpus, args.gpus, args.gpus))

    import torch
    import term_amd as T
    from term_amd import synth
    from term_amd._lib import spec
    from term_amd.distributed import rccl_comm, sharded_suite_step

    torch.cuda.set_device(local_rank)
    dist = None
    distributed = world > 1 or args.force_distributed
    if distributed:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    layout, unique_cols = synth.COLUMNS_16, synth.UNIQUE_COLUMNS_16
    n_total = (args.rows // (64 * world)) * 64 * world  # 64-row aligned shard boundaries
    n_local = n_total // world
    row0 = rank * n_local

    T.init(device_id=local_rank, distinct_capacity_hint=n_local)
    specs = build_suite(T, spec, layout, unique_cols)
    stream = torch.cuda.Stream()
    # one fused plan on every rank: every column buffer is read by one scan, the unique columns feed the distinct pass
    plan = T.Plan(specs)
    st = T.State(plan, stream=stream.cuda_stream)
    comm = rccl_comm(dist, rank, world) if distributed else None  # the library's own RCCL communicator

    table = synth.make_table(layout, row0, n_local, n_total, args.seed, "cuda")
    columns = []
    for (kind, _), (vals, validity) in zip(layout, table):
        ctor = T.Column.float64 if kind.startswith("f_") else T.Column.int64
        columns.append(ctor(vals, validity, length=n_local))
    torch.cuda.synchronize()

    def step():
        if not distributed:
            st.reset()
            st.update(columns)
            return st.finalize()
        return sharded_suite_step(plan, st, columns, comm)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    first_step_ms = None
    for w in range(args.warmup):
        t1 = time.perf_counter()
        res = step()
        if w == 0:  # the process's first state: its buffers come from hipMalloc, the kernels' code objects are loaded
            first_step_ms = (time.perf_counter() - t1) * 1e3
    st.profile_enable(True)
    st.profile_reset()
    fence()
    step_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t1 = time.perf_counter()
        res = step()  # (returns the step's results: the device is through with it)
        step_ms.append((time.perf_counter() - t1) * 1e3)
    fence()
    dt = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    prof = st.profile_get("scan")
    prof_d = st.profile_get("distinct")
    # where a rank's step went (N > 1 and --force-distributed): the kernels' own clocks (events on the state's stream)
    # and the host clock around the phases of tgx_allreduce; per step, the MAX over ranks of each
    breakdown = None
    if distributed:
        names = [("scan_ms", "scan"), ("key_passes_ms", "distinct"), ("facts_ms", "xr_facts"), ("exchange_ms", "xr_exchange"),
                 ("pack_ms", "xr_pack"), ("state_gather_ms", "xr_gather"), ("merge_ms", "xr_merge")]
        mine = torch.tensor([st.profile_get(k)["total_ms"] / max(1, args.steps) for _, k in names], dtype=torch.float64,
                            device="cuda")
        dist.all_reduce(mine, op=dist.ReduceOp.MAX)
        breakdown = {n: float(v) for (n, _), v in zip(names, mine.cpu().tolist())}
        breakdown["note"] = ("per step, max over ranks.  scan / key_passes: GPU time of the kernels; facts .. merge: host "
                             "clock around the phases of tgx_allreduce -- facts includes the wait for the shard's key "
                             "passes, pack the wait for its scan (the exchange runs beside the scan)")
    st.profile_enable(False)

    # ---- verification outside the timed region: closed-form facts of the synthetic table ----
    verified = None
    if not args.no_verify:
        by_col = {}
        for s, r in zip(specs, res):
            by_col.setdefault(s.column, {})[s.kind] = r
        ok = True
        for ci, (kind, has_validity) in enumerate(layout):
            c, stt = by_col[ci][T.COUNT], by_col[ci][T.NUMERIC_STATS]
            ok &= c.total == n_total and stt.total == n_total and c.non_null == stt.non_null
            if not has_validity:
                ok &= c.non_null == n_total
            else:
                ok &= abs(c.non_null / n_total - (1 - synth.NULL_RATE)) < 1e-3
        d_id, d_k = by_col[0][T.DISTINCT], by_col[1][T.DISTINCT]
        ok &= d_id.distinct == n_total  # bijective id column: uniqueness ratio exactly 1.0
        ok &= by_col[0][T.NUMERIC_STATS].min_i == 0 and by_col[0][T.NUMERIC_STATS].max_i == n_total - 1
        ok &= by_col[0][T.NUMERIC_STATS].sum_i == n_total * (n_total - 1) // 2
        ok &= 0 < d_k.distinct <= max(1, n_total // 10)
        verified = bool(ok)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        rows_per_s = n_total * args.steps / dt
        alg_suite = synth.algorithmic_bytes(layout, n_total)
        scan_ms = prof["total_ms"] / max(1, prof["launches"])
        scan_bytes = prof["bytes"] / max(1, prof["launches"])
        achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        # HBM traffic of the dominant kernel: NOT measured in this run (PMC counters need their own rocprofv3 passes)
        # but read from the committed passes of this same command on the builder's box -- profiles/rNN_pmc_*.json, the
        # latest round's: separate --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 FETCH_SIZE x2 correction applied;
        # `traffic_source` says so
        traffic, traffic_source = None, None
        for name in ("r06_pmc_1Brows_16cols.json", "r05_pmc_1Brows_16cols.json", "r04_pmc_1Brows_16cols.json", "r03_pmc_1Brows_16cols.json",
                     "r02_pmc_1Brows_16cols.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    pmc = json.load(f)
                if pmc["rows_total"] == n_total and pmc["n_gpus"] == world and \
                        pmc["scan_kernel_algorithmic_bytes_per_launch"] == int(scan_bytes):
                    traffic = pmc["scan_kernel_traffic_bytes_per_launch"]
                    traffic_source = "profiles/%s (committed rocprofv3 --pmc passes of this command, not this run)" % name
                    break
            except (OSError, KeyError, ValueError):
                pass
        out = {
            "metric": "validated rows/sec, 16-col null+range+unique suite",
            "value": rows_per_s, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup,
            # rank 0's per-step wall clocks: the median is the line's ms_per_step (SURVEY 8d's protocol); `value` is
            # still rows / the whole timed region (max over ranks), i.e. it follows ms_mean
            "ms_per_step": sorted(step_ms)[len(step_ms) // 2], "ms_min": min(step_ms), "ms_mean": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "int64+f64", "data": "synthetic",
            "config": {"workload": "16-col (8 int64 + 8 float64, 12 nullable) null+range+unique suite: "
                                   "completeness x16, min/max/mean x16, uniqueness x2",
                       "rows_total": n_total, "rows_per_gpu": n_local, "cols": len(layout),
                       "parallelism": "row-range shards x%d" % world,
                       "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                       "suite_algorithmic_bytes": alg_suite,
                       "suite_hbm_gbs": alg_suite / (dt / args.steps) / 1e9,
                       "suite_frac_of_8TBs": alg_suite / (dt / args.steps) / 1e9 / HBM_PEAK_GBS / world,  # per GPU
                       "distinct_ms_per_step": prof_d["total_ms"] / max(1, args.steps),
                       "verified": verified},
            "roofline": {"bound": "hbm", "kernel": "scan_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "launch_ms": scan_ms, "algorithmic_bytes_per_launch": scan_bytes,
                         "columns_per_launch": "the 14 columns without a uniqueness check: the two key columns' "
                                               "aggregates come out of their DISTINCT pass (partition_kernel)"},
        }
        if breakdown is not None:
            out["config"]["rank_breakdown"] = breakdown
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed on rank 0 of the single-GPU run only
            sample = min(args.cpu_sample_rows, n_local)
            ref_line, why_not = reference_cpu_baseline(sample, args.seed)  # term-guard itself, where it can be built
            out["cpu_baseline"] = ref_line or cpu_baseline(torch, layout, unique_cols, table, sample)
            if ref_line is None:
                out["cpu_baseline"]["reference_probe"] = why_not
        if world == 1 and not distributed and not args.no_secondary and (n_total == 1_000_000_000 or args.secondary):
            # the other BASELINE.json configs and the cold step, after the headline's timed region (tools/secondary_bench.py);
            # the headline's own keys above are what they always were
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import secondary_bench

            warm_ms = out["ms_per_step"]
            st.close()  # (its blocks go to the library's cache: the cold states below start from there)
            out["secondary"] = secondary_bench.measure(
                T, torch, synth, spec, layout, unique_cols, plan, table, columns, n_total, args.seed,
                steps=5, warmup=2, warm_headline_ms=warm_ms,
                log=lambda m: print(m, file=sys.stderr, flush=True))
            out["secondary"]["cold"]["headline_first_state_step_ms"] = first_step_ms
        line = json.dumps(out)
    else:
        line = None
    if distributed:
        dist.barrier()
        del comm  # ncclCommDestroy before the process group goes away
        dist.destroy_process_group()
    if line is not None:
        # the ONE JSON line, last thing on stdout (RCCL prints its version banner on teardown)
        sys.stdout.flush()
        print(line, flush=True)


if __name__ == "__main__":
    main()
