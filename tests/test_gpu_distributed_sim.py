"""-m gpu: the N > 1 step of bench.py -- tgx_update on the rank's row shard, then tgx_allreduce (the C entry point of
the cross-rank step) -- with 2 .. 8 ranks simulated on one GPU: every rank is a thread with its own state and row
shard; the transport is the thread-barrier stand-in of term_amd.distributed.thread_comm (handed DEVICE pointers
like RCCL, or HOST pointers like an MPI-style transport), so the facts gather, the all-to-all of re-based
range-bitmap slices, the hash-owner record exchange, the agreed-capacity state gather and the rank-ordered merge
all run exactly as they do over RCCL.  Every rank's results must equal the single-state results of the table."""
import threading

import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import numeric_column
from term_amd.distributed import ThreadGroup, shard_rows, sharded_suite_step, thread_comm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3, 4, 8])  # 2 / 4 / 8: the row-shard layouts the scaling bench runs
@pytest.mark.parametrize("dense", [True, False])
@pytest.mark.parametrize("device_buffers", [True, False])
def test_simulated_ranks_match_one_state(dense, world, device_buffers):
    import torch

    rng = np.random.default_rng(12 + dense)
    n = 2_400_000 + 64 * 7
    ids = rng.permutation(n).astype(np.int64)                       # unique, dense range -> bitmap slices
    if dense:
        keys = rng.integers(0, n // 10, size=n, dtype=np.int64)     # duplicates across shards, dense range
    else:
        keys = rng.integers(-2**62, 2**62, size=n, dtype=np.int64)  # sparse range -> hash sets -> key records
        keys[: n // 4] = keys[n // 4: n // 2]
    flt = rng.standard_normal(n) * 100
    masks = [None, rng.random(n) >= 0.07, rng.random(n) >= 0.2]
    cols_np = [ids, keys, flt]
    valid = [None if m is None else orc.pack_validity(m) for m in masks]
    specs = []
    for ci in range(3):
        specs += [spec(T.COUNT, ci), spec(T.NUMERIC_STATS, ci)]
    # (no NUMERIC_STATS spec is needed next to a DISTINCT one: the ranks agree on the range inside tgx_allreduce)
    specs += [spec(T.DISTINCT, 0), spec(T.DISTINCT, 1, flags=T.FLAG_MULTIPLICITY)]
    T.init()
    plan = T.Plan(specs)  # ONE fused plan, as on a single GPU
    # reference: one state over the whole table
    whole = [numeric_column(c, v, True) for c, v in zip(cols_np, valid)]
    one = T.State(plan)
    one.update(whole)
    want = one.finalize()

    group = ThreadGroup(world)
    results, errors = [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            lo, hi = shard_rows(n, world, rank)
            shard = [numeric_column(c, v, True, offset=lo, length=hi - lo) for c, v in zip(cols_np, valid)]
            st = T.State(plan)
            comm = thread_comm(group, rank, device_buffers=device_buffers)
            for _ in range(2):  # twice: bitmap buffers, send / receive scratch and the agreed blob capacity are reused
                res = sharded_suite_step(plan, st, shard, comm)
            results[rank] = res
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            group.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors

    def key(r):
        return (r.kind, r.total, r.non_null, r.has_value, r.min_i, r.max_i, r.sum_i, r.distinct, r.groups_once)

    for rank in range(world):
        got = results[rank]
        assert [key(r) for r in got] == [key(r) for r in want], rank
        for g, w in zip(got, want):
            if g.kind == T.NUMERIC_STATS and g.is_float:
                assert (g.min_f, g.max_f) == (w.min_f, w.max_f)
                assert abs(g.sum_f - w.sum_f) <= 1e-9 * abs(w.sum_f) and abs(g.mean - w.mean) <= 1e-9 * abs(w.mean)
            if g.kind == T.NUMERIC_STATS:  # rank-ordered merge: bit-identical on every rank
                assert (g.sum_f, g.mean) == (results[0][got.index(g)].sum_f, results[0][got.index(g)].mean)
    # exact facts, independent of the device path
    assert want[-2].distinct == n
    d = orc.distinct_bits64(keys.view(np.uint64), valid[1])
    assert (want[-1].distinct, want[-1].groups_once) == (d.distinct, d.groups_once)


def _run_ranks(world, plan, shards_of, device_buffers=True, steps=1):
    import torch

    group = ThreadGroup(world)
    results, errors = [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            st = T.State(plan)
            comm = thread_comm(group, rank, device_buffers=device_buffers)
            shard = shards_of(rank)
            for _ in range(steps):
                res = sharded_suite_step(plan, st, shard, comm)
            results[rank] = (res, st)
        except Exception as e:  # noqa: BLE001
            import traceback

            errors.append((rank, traceback.format_exc()))
            group.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=150)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads), "a rank is stuck"
    return results


@pytest.mark.parametrize("world", [2, 5])
def test_ranks_with_unequal_ranges_empty_shards_and_mixed_key_sets(world):
    """What row shards of real tables look like and the round-1 exchange could not take: local value
    ranges that differ per rank (a sorted id column: every rank's bitmap has its own base), a rank whose shard of a
    column is all NULL (it holds no key set but owns a slice), ranks that chose different kinds of key set (small
    shards go to the hash set, big ones to a bitmap), and a DISTINCT column without any NUMERIC_STATS spec."""
    rng = np.random.default_rng(5)
    n = 1_200_000
    ids = np.arange(n, dtype=np.int64) * 3 - 1_000_000            # sorted: rank r holds its own sub-range
    grp = rng.integers(0, 50_000, size=n, dtype=np.int64)
    mask = rng.random(n) >= 0.1
    bounds = [0, 70_016]                                           # unequal shards
    for r in range(1, world - 1):
        bounds.append(bounds[-1] + (n - 70_016) // (world - 1) // 64 * 64)
    bounds.append(n)
    mask[bounds[-2]:] = False                                      # the last rank's shard of `grp` is all NULL
    valid = orc.pack_validity(mask)
    mix = rng.integers(0, 100_000, size=n, dtype=np.int64)         # dense everywhere ...
    mix[: bounds[1]] = rng.integers(-2**62, 2**62, size=bounds[1], dtype=np.int64)  # ... but sparse on rank 0: a hash set
    mix[5] = -1                                                    # the all-ones pattern travels in a side counter
    specs = [spec(T.DISTINCT, 0), spec(T.DISTINCT, 1, flags=T.FLAG_MULTIPLICITY), spec(T.COUNT, 1),
             spec(T.DISTINCT, 2, flags=T.FLAG_MULTIPLICITY)]
    T.init()
    plan = T.Plan(specs)

    def shards_of(rank):
        lo, hi = bounds[rank], bounds[rank + 1]
        return [numeric_column(ids, None, True, offset=lo, length=hi - lo),
                numeric_column(grp, valid, True, offset=lo, length=hi - lo),
                numeric_column(mix, None, True, offset=lo, length=hi - lo)]

    dm = orc.distinct_bits64(mix.view(np.uint64), None)
    d = orc.distinct_bits64(grp.view(np.uint64), valid)
    for res, _ in _run_ranks(world, plan, shards_of, steps=2):
        assert (res[0].total, res[0].non_null, res[0].distinct) == (n, n, n)
        assert (res[1].total, res[1].non_null, res[1].distinct, res[1].groups_once) == \
            (n, d.non_null, d.distinct, d.groups_once)
        assert (res[2].total, res[2].non_null) == (n, d.non_null)
        assert (res[3].total, res[3].distinct, res[3].groups_once) == (n, dm.distinct, dm.groups_once)


@pytest.mark.parametrize("world,device_buffers", [(1, True), (2, True), (3, False), (5, True)])
def test_spearman_over_ranks(world, device_buffers):
    """SQL RANK() over the union of the shards (tgx_allreduce: sort locally, agree on splitters, every key to the rank
    that owns its value range, ranks back to their rows): the five UInt64 sums must equal the oracle's on the whole
    table, bit for bit, on every rank -- heavy ties (equal keys meet on one rank), NULLs on either side, a sorted
    column (each rank's keys are one value range: almost everything travels), unequal shards and an empty one; with
    the suite's other checks in the same plan."""
    from gpu_util import make_f64, make_i64

    rng = np.random.default_rng(40 + world)
    n = 400_000
    xi, xv = make_i64(rng, n, -300, 300, null_frac=0.1)           # heavy ties
    yf, yv = make_f64(rng, n, "normal", null_frac=0.05)
    yf = np.round(yf, 2)
    srt = np.sort(rng.standard_normal(n))                         # rank r's shard is one value range
    bounds = [0, 1000] if world > 1 else [0]
    for r in range(1, world - 1):
        bounds.append(bounds[-1] + (n - 1000) // (world - 1) // 64 * 64)
    bounds.append(n)
    if world >= 3:
        bounds[2] = bounds[1]                                     # rank 1 holds no rows at all
    specs = [spec(T.SPEARMAN, 0, column2=1), spec(T.SPEARMAN, 2, column2=1, flags=T.FLAG_EXACT_RANK_SUMS),
             spec(T.COMOMENTS, 0, column2=1), spec(T.COUNT, 0), spec(T.DISTINCT, 0)]
    T.init()
    plan = T.Plan(specs)

    def shards_of(rank):
        lo, hi = bounds[rank], bounds[rank + 1]
        return [numeric_column(xi, xv, True, offset=lo, length=hi - lo),
                numeric_column(yf, yv, True, offset=lo, length=hi - lo),
                numeric_column(srt, None, True, offset=lo, length=hi - lo)]

    want = orc.spearman_state(xi, yf, xv, yv)
    want2 = orc.spearman_state(srt, yf, None, yv)
    single, _, _ = __import__("gpu_util").run_plan(specs, [[numeric_column(xi, xv, True), numeric_column(yf, yv, True),
                                                             numeric_column(srt, None, True)]])
    for res, _ in _run_ranks(world, plan, shards_of, device_buffers=device_buffers, steps=2):
        g = res[0]
        assert (g.total, g.non_null) == (n, want.n)
        assert (g.sum_x, g.sum_y, g.sum_x2, g.sum_y2, g.sum_xy) == \
            (want.sum_x, want.sum_y, want.sum_x2, want.sum_y2, want.sum_xy)
        e = res[1]
        assert (e.total, e.non_null) == (n, want2.n)
        assert (e.sum_x, e.sum_y, e.sum_x2, e.sum_y2, e.sum_xy) == \
            (single[1].sum_x, single[1].sum_y, single[1].sum_x2, single[1].sum_y2, single[1].sum_xy)
        assert res[2].non_null == want.n and res[3].non_null == single[3].non_null
        assert res[4].distinct == single[4].distinct


# ---- a rank that fails on its own between two collectives (allreduce.cpp: `local`, the status words, the deadline) ----
@pytest.mark.parametrize("site", [1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("device_buffers", [True, False])
def test_a_local_failure_fails_every_rank_at_the_same_point(site, device_buffers, monkeypatch):
    """TGX_FAULT_INJECT="rank:site" makes one rank fail where only IT can (preparing its key sets, allocating the
    exchange's buffers / its spare bitmaps, exporting its keys, allocating the receive buffer of the records, importing
    them, packing its state).  It must not simply return -- its peers would sit in the next collective for ever --:
    every rank returns an error, the same status, within the collective deadline, and none is left behind."""
    import torch

    world, bad = 4, 2
    monkeypatch.setenv("TGX_FAULT_INJECT", "%d:%d" % (bad, site))
    monkeypatch.setenv("TGX_COLLECTIVE_TIMEOUT_MS", "20000")
    rng = np.random.default_rng(site)
    n = 600_000
    ids = rng.permutation(n).astype(np.int64)                      # dense: bitmap slices (sites 2, 3)
    keys = rng.integers(-2**62, 2**62, size=n, dtype=np.int64)     # sparse: key records (sites 4, 5, 6)
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0), spec(T.DISTINCT, 1, flags=T.FLAG_MULTIPLICITY), spec(T.NUMERIC_STATS, 0)])
    group = ThreadGroup(world)
    outcome = [None] * world

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            lo, hi = shard_rows(n, world, rank)
            shard = [numeric_column(c, None, True, offset=lo, length=hi - lo) for c in (ids, keys)]
            st = T.State(plan)
            comm = thread_comm(group, rank, device_buffers=device_buffers)
            sharded_suite_step(plan, st, shard, comm)
            outcome[rank] = ("ok", "")
        except T.TgxError as e:
            outcome[rank] = (e.status, str(e))
        except Exception as e:  # noqa: BLE001  (a broken barrier: somebody was left behind)
            outcome[rank] = ("other", repr(e))
            group.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in a collective: %r" % (outcome,)
    statuses = {o[0] for o in outcome}
    assert statuses == {"TGX_OUT_OF_MEMORY"}, outcome
    assert "injected failure at site %d" % site in outcome[bad][1]
    for r in range(world):
        if r != bad:
            assert "rank %d failed" % bad in outcome[r][1], outcome[r]


def test_growing_ids_over_several_flushes_without_partition_passes(monkeypatch):
    """with the exchange on its second stream, the facts round read the key column's running MIN / MAX
    while the scan that produces them (a pass that is NOT partitioned: TGX_PARTITION_MIN_ROWS set high) could still be
    running, narrowed the agreed range with stale values and dropped the last pass's keys from the re-based bitmaps.
    `keys_ready` now stands only when the aggregates were produced before it."""
    monkeypatch.setenv("TGX_PARTITION_MIN_ROWS", str(1 << 40))
    world, n, batches = 4, 4_000_000, 5
    rng = np.random.default_rng(77)
    ids = np.arange(n, dtype=np.int64) + 1_000_000  # growing: every flush extends the range
    other = rng.standard_normal(n)
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0), spec(T.COUNT, 0), spec(T.NUMERIC_STATS, 1)])

    import torch

    group = ThreadGroup(world)
    results, errors = [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            lo, hi = shard_rows(n, world, rank)
            st = T.State(plan)
            comm = thread_comm(group, rank, device_buffers=True)
            for _ in range(3):
                st.reset()
                cut = np.linspace(lo, hi, batches + 1).astype(np.int64) // 64 * 64
                cut[0], cut[-1] = lo, hi
                for a, b in zip(cut[:-1], cut[1:]):
                    st.update([numeric_column(ids, None, True, offset=int(a), length=int(b - a)),
                               numeric_column(other, None, True, offset=int(a), length=int(b - a))])
                st.allreduce(comm)
                results[rank] = st.finalize()
                assert results[rank][0].distinct == n, (rank, results[rank][0].distinct)
        except Exception:  # noqa: BLE001
            import traceback

            errors.append((rank, traceback.format_exc()))
            group.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=150)
    assert not errors, errors
    for r in results:
        assert (r[0].distinct, r[1].total) == (n, n)
