"""world_size-2 `gloo` test of the N>1 path on CPU: row-range shards -> packed partial states -> all-gather ->
rank-ordered merge (term_amd/distributed.py + tgx_state_deserialize / tgx_merge / tgx_finalize, none of which
needs a device).  Per-shard aggregates come from the oracle here (there is no GPU in this container); the GPU
box runs the same merge code on states produced by the kernels (bench.py --gpus N)."""
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
    from term_amd.distributed import allgather_merge, shard_rows

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
    merged = allgather_merge(plan, blob, dist, world)
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
    # the one-collective gather: ragged sizes, a cached capacity that one rank outgrows, empty payloads
    from term_amd.distributed import agree_on_ranges, allgather_blobs, allgather_many
    g = allgather_blobs(b"x" * (10 + rank), dist, world, cache_key="t")
    assert [len(x) for x in g] == [10, 11] and g[1] == b"x" * 11
    g = allgather_blobs(b"y" * (5000 * (rank + 1)), dist, world, cache_key="t")
    assert [len(x) for x in g] == [5000, 10000] and g[1] == b"y" * 10000
    g = allgather_blobs(b"", dist, world, cache_key="t")
    assert g == [b"", b""]
    g = allgather_many([b"a" * rank, b"bb"], dist, world)
    assert g == [[b"", b"bb"], [b"a", b"bb"]]
    r = agree_on_ranges([(True, 5 - rank, 100 + rank), (rank == 1, -7, 7), (False, 0, 0)], dist, world, device="cpu")
    assert r == [(4, 101), (-7, 7), None], r
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
