"""-m gpu: the N > 1 step of bench.py (term_amd.distributed.sharded_suite_step) with THREE ranks simulated on one
GPU: every rank is a thread with its own states and row shard; the collectives are a thread-barrier stand-in for
torch.distributed (same call signatures, same data movement), so the range agreement, the bitmap-slice all-to-all
with its in-place strided adoption, the hash-owner fallback, the one-collective state gather and the rank-ordered
merge all run exactly as they do over RCCL.  The merged results must equal the single-state results of the table."""
import threading

import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import numeric_column
from term_amd.distributed import shard_rows, sharded_suite_step

pytestmark = pytest.mark.gpu


class FakeGroup:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class FakeDist:
    """the subset of torch.distributed the step uses, for threads sharing one device"""

    def __init__(self, group, rank):
        self.g, self.rank = group, rank

    def _publish(self, t):
        import torch

        torch.cuda.synchronize()
        self.g.slots[self.rank] = t
        self.g.barrier.wait()

    def _done(self):
        import torch

        torch.cuda.synchronize()
        self.g.barrier.wait()

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None):
        w = self.g.world
        isz = list(input_split_sizes) if input_split_sizes is not None else [inp.numel() // w] * w
        self._publish((inp, isz))
        pos = 0
        for r in range(w):
            src, sz = self.g.slots[r]
            start, n = sum(sz[: self.rank]), sz[self.rank]
            if output_split_sizes is not None:
                assert output_split_sizes[r] == n
            out[pos:pos + n] = src[start:start + n]
            pos += n
        self._done()

    def all_gather_into_tensor(self, out, inp):
        self._publish(inp)
        n = inp.numel()
        for r in range(self.g.world):
            out[r * n:(r + 1) * n] = self.g.slots[r].to(out.device)
        self._done()

    def all_to_all(self, outs, ins):
        self._publish(list(ins))
        for r in range(self.g.world):
            outs[r].copy_(self.g.slots[r][self.rank])
        self._done()

    def all_gather(self, outs, inp):
        self._publish(inp)
        for r in range(self.g.world):
            outs[r].copy_(self.g.slots[r])
        self._done()


@pytest.mark.parametrize("world", [2, 3, 4, 8])  # 2 / 4 / 8: the row-shard layouts the scaling bench runs
@pytest.mark.parametrize("dense", [True, False])
def test_simulated_ranks_match_one_state(dense, world):
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
    stat_specs = []
    for ci in range(3):
        stat_specs += [spec(T.COUNT, ci), spec(T.NUMERIC_STATS, ci)]
    distinct_specs = [spec(T.DISTINCT, 0), spec(T.DISTINCT, 1, flags=T.FLAG_MULTIPLICITY)]
    T.init()
    plan, plan_d = T.Plan(stat_specs), T.Plan(distinct_specs)
    # reference: one state over the whole table
    whole = [numeric_column(c, v, True) for c, v in zip(cols_np, valid)]
    one, one_d = T.State(plan), T.State(plan_d)
    one.update(whole)
    one_d.update(whole)
    want = one.finalize() + one_d.finalize()

    group = FakeGroup(world)
    results, errors = [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            lo, hi = shard_rows(n, world, rank)
            shard = [numeric_column(c, v, True, offset=lo, length=hi - lo) for c, v in zip(cols_np, valid)]
            st, st_d = T.State(plan), T.State(plan_d)
            for _ in range(2):  # twice: buffers, cached gather capacity and range hints are reused across steps
                res = sharded_suite_step(plan, st, plan_d, st_d, stat_specs, distinct_specs, shard, FakeDist(group, rank),
                                         world, rank, cache_key="sim%d_%d" % (dense, world))
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
    # exact facts, independent of the device path
    assert want[-2].distinct == n
    d = orc.distinct_bits64(keys.view(np.uint64), valid[1])
    assert (want[-1].distinct, want[-1].groups_once) == (d.distinct, d.groups_once)
