"""world_size-2 `gloo` test of the N>1 path on CPU: row-range shards -> tgx_allreduce (the C entry point of the
cross-rank step: facts all-gather, packed partial states in one all-gather with the agreed-capacity protocol,
rank-ordered merge) over a torch.distributed transport with host buffers (term_amd.distributed.torch_dist_comm),
none of which needs a device when the states are host-side.  Per-shard aggregates come from the oracle here (there
is no GPU in this container); the GPU box runs the same entry point on states produced by the kernels (bench.py
--gpus N over RCCL, tests/test_gpu_distributed_sim.py with threaded ranks)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch.distributed as dist
    import oracle_binding as orc
    import term_amd as T
    from term_amd import wire
    from term_amd._lib import spec
    from term_amd.distributed import shard_rows, shared_fingerprint_key, torch_dist_comm

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(123)            # same table on every rank; each rank reads only its shard
    n = 100_000 + 37
    iv = rng.integers(-5000, 5000, size=n, dtype=np.int64)
    fv = rng.standard_normal(n) * 10
    mask = rng.random(n) >= 0.07
    strs = ["u%d@x.org" % i if i % 3 else "bad" for i in range(n)]
    lo, hi = shard_rows(n, world, rank)
    assert lo % 64 == 0

    def shard_validity():
        return orc.pack_validity(mask[lo:hi])

    plan = T.Plan([spec(T.COUNT, 0), spec(T.NUMERIC_STATS, 0, flags=T.FLAG_VARIANCE), spec(T.NUMERIC_STATS, 1),
                   spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.COMOMENTS, 0, column2=1),
                   spec(T.KLL, 1, kll_k=200), spec(T.REGEX_MATCH, 2, pattern="@", flags=0)])
    v = shard_validity()
    si, sf = orc.stats(iv[lo:hi].copy(), v), orc.stats(fv[lo:hi].copy(), v)
    exact_sum = int(iv[lo:hi][mask[lo:hi]].astype(object).sum()) if si.non_null else 0
    var = (si.non_null, si.sum_hi / si.non_null, si.var_samp * (si.non_null - 1)) if si.has_variance else None
    # owner-partitioned distinct: this rank owns the keys with key % world == rank, over the WHOLE table
    # (what the hash-owner exchange leaves on each rank); row counts stay per shard
    allv = orc.pack_validity(mask)
    owned = (iv % world == rank) & mask
    d_owned = orc.distinct_bits64(iv[owned].copy().view(np.uint64))
    d_shard = orc.distinct_bits64(iv[lo:hi].copy().view(np.uint64), v)
    cm = orc.comoments(iv[lo:hi].copy(), fv[lo:hi].copy(), v, v)
    kept = fv[lo:hi][mask[lo:hi]]
    offs, data, _ = orc.utf8_from_list(strs[lo:hi])
    rx = orc.Regex("@").count_utf8(offs, data, None, null_is_valid=False)
    blob = wire.pack(
        scan=[wire.scan_acc(hi - lo, si.non_null, si.min_i, si.max_i, exact_sum, var=var),
              wire.scan_acc(hi - lo, sf.non_null, sf.min_f, sf.max_f, sf.sum_hi, is_float=True)],
        comoments=[wire.comoment_acc(hi - lo, cm.n, cm.sum_x, cm.sum_y, cm.sum_x2, cm.sum_y2, cm.sum_xy)],
        distinct=[wire.distinct_counts(hi - lo, d_shard.non_null, d_owned.distinct, d_owned.distinct - d_owned.groups_once)],
        kll=[wire.kll_state(200, len(kept), float(kept.min()), float(kept.max()), [sorted(kept.tolist())[:1000]] if len(kept) <= 1000 else
             [[], sorted(kept.tolist())[::2]])],
        regex=[wire.regex_counts(rx.total, rx.matches)])
    comm = torch_dist_comm(dist, rank, world)
    merged = T.State.deserialize(plan, blob)     # this rank's partial state (host-side: no device needed)
    merged.allreduce(comm)                       # -> the state of the whole table, on every rank
    res = merged.finalize()
    full_i, full_f = orc.stats(iv, allv), orc.stats(fv, allv)
    d = orc.distinct_bits64(iv.view(np.uint64), allv)
    out = dict(rank=rank,
               count=[res[0].total, res[0].non_null], want_count=[n, full_i.non_null],
               istats=[res[1].min_i, res[1].max_i, res[1].sum_i], want_istats=[full_i.min_i, full_i.max_i, full_i.sum_i_wrapping],
               var=res[1].var_samp, want_var=full_i.var_samp,
               fstats=[res[2].min_f, res[2].max_f, res[2].sum_f], want_fstats=[full_f.min_f, full_f.max_f, full_f.sum_hi],
               distinct=[res[3].total, res[3].non_null, res[3].distinct, res[3].groups_once],
               want_distinct=[d.total, d.non_null, d.distinct, d.groups_once],
               sum_xy=res[4].sum_xy, want_sum_xy=orc.comoments(iv, fv, allv, allv).sum_xy,
               kll_n=res[5].kll_n, want_kll_n=int(mask.sum()), median=merged.kll_quantile(5, 0.5),
               want_median=float(np.median(fv[mask])),
               regex=[res[6].total, res[6].matches], want_regex=[n, sum(1 for s in strs if "@" in s)],
               blob=merged.serialize().hex()[:64])
    # the agreed-capacity gather: a second plan on the same comm whose blobs differ in size per rank (rank 1 carries a
    # KLL level of 5000 items, rank 0 none) and then outgrow the capacity agreed in the first round
    plan2 = T.Plan([spec(T.KLL, 0, kll_k=200), spec(T.COUNT, 1)])
    for scale in (1, 4):
        items = sorted(float(i) for i in range(5000 * scale * rank))
        lv = [items[:500], items[500:]] if items else []
        part = T.State.deserialize(plan2, wire.pack(kll=[wire.kll_state(200, len(items), 0.0, max(1.0, float(len(items))), lv)] ,
                                                    count=[wire.count_acc(10 + rank, 7)]))
        part.allreduce(comm)
        r2 = part.finalize()
        assert r2[0].kll_n == 5000 * scale and (r2[1].total, r2[1].non_null) == (21, 14), (r2[0].kll_n, r2[1].total)
    # a plan with a SPEARMAN task goes through the same call (no rank holds a pair here: the ranking has nothing to
    # exchange, the other tasks are reduced as usual)
    plan3 = T.Plan([spec(T.SPEARMAN, 0, column2=1), spec(T.COUNT, 0)])
    sp = T.State.deserialize(plan3, wire.pack(count=[wire.count_acc(100 + rank, 90)]))
    sp.allreduce(comm)
    r3 = sp.finalize()
    assert (r3[0].total, r3[0].non_null, r3[1].total, r3[1].non_null) == (0, 0, 200 + world * (world - 1) // 2, 90 * world)
    # an allreduce of an all-reduced state is the W-fold sum: the call is a plain cross-rank merge, nothing is cached
    again = T.State.deserialize(plan, merged.serialize())
    again.allreduce(comm)
    assert again.finalize()[0].total == world * n
    # one fingerprint key for all ranks (string / tuple keys travel as fingerprints): rank 0's, broadcast
    key = shared_fingerprint_key(dist, rank)
    keyed = T.Plan([spec(T.DISTINCT, 0)], fingerprint_key=key)
    assert keyed.fingerprint_key() == key and T.Plan([spec(T.DISTINCT, 0)]).fingerprint_key() != key
    out["fp_key"] = key.hex()
    print("RESULT " + json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()
''')


def test_world2_gloo_allgather_merge(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text("ROOT = %r\n" % ROOT + WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-2000:]
        outs.append(o)
    import json

    results = [json.loads(line[len("RESULT "):]) for o in outs for line in o.splitlines() if line.startswith("RESULT ")]
    assert len(results) == 2
    for r in results:
        assert r["count"] == r["want_count"]
        assert r["istats"] == r["want_istats"]
        assert abs(r["var"] - r["want_var"]) <= 1e-9 * r["want_var"]
        assert r["fstats"][:2] == r["want_fstats"][:2] and abs(r["fstats"][2] - r["want_fstats"][2]) <= 1e-9 * abs(r["want_fstats"][2])
        assert r["distinct"] == r["want_distinct"]
        assert abs(r["sum_xy"] - r["want_sum_xy"]) <= 1e-9 * abs(r["want_sum_xy"])
        assert r["kll_n"] == r["want_kll_n"] and abs(r["median"] - r["want_median"]) < 0.5
        assert r["regex"] == r["want_regex"]
    # rank-ordered merge: both ranks hold the identical merged state
    assert results[0]["blob"] == results[1]["blob"]
    assert results[0]["fp_key"] == results[1]["fp_key"] and len(results[0]["fp_key"]) == 32
