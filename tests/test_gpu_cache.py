"""-m gpu: the library's cache of device / pinned blocks (csrc/devcache.cpp; include/tgx.h `tgx_trim`,
`tgx_cache_stats_get`).  `ValidationSuite::run` (core/suite.rs:399) is one state per table: from the second state of a
process on, create -> update -> finalize -> destroy performs no hipMalloc, and results do not depend on whose blocks a
state was handed."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import make_f64, make_i64, numeric_column

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_run(plan, cols):
    st = T.State(plan)
    st.update(cols)
    res = st.finalize()
    out = [(r.total, r.non_null, r.min_i, r.max_i, r.sum_i, r.distinct, r.groups_once, r.sum_f.hex()) for r in res]
    del st
    return out


def test_second_state_allocates_nothing_and_trim_gives_the_memory_back():
    import torch

    rng = np.random.default_rng(11)
    n = 3_000_003
    ids = rng.permutation(n).astype(np.int64)
    keys, kval = make_i64(rng, n, 0, n // 10, 0.05)
    wide, wval = make_i64(rng, n, -(1 << 40), 1 << 40, 0.05)
    f, fval = make_f64(rng, n, "normal", 0.05)
    T.init(distinct_capacity_hint=n)
    cols = [numeric_column(ids, None, True), numeric_column(keys, kval, True), numeric_column(wide, wval, True),
            numeric_column(f, fval, True)]
    specs = [spec(T.COUNT, c) for c in range(4)] + [spec(T.NUMERIC_STATS, c) for c in range(4)] + \
            [spec(T.DISTINCT, 0), spec(T.DISTINCT, 1, flags=T.FLAG_MULTIPLICITY), spec(T.DISTINCT, 2), spec(T.DISTINCT, 3),
             spec(T.KLL, 3, kll_k=200), spec(T.COMOMENTS, 2, column2=3)]
    plan = T.Plan(specs)
    T.trim()
    first = one_run(plan, cols)
    s1 = T.cache_stats()
    assert s1.device_cached_bytes > 0 and s1.device_cached_blocks > 0  # the destroyed state's blocks are kept
    second = one_run(plan, cols)
    s2 = T.cache_stats()
    assert second == first
    assert s2.device_misses == s1.device_misses, (s1.device_misses, s2.device_misses)  # no hipMalloc for the second state
    assert s2.device_hits > s1.device_hits
    assert s2.pinned_misses == s1.pinned_misses
    third = one_run(plan, cols)
    assert third == first and T.cache_stats().device_misses == s1.device_misses
    # against the oracle (whatever stale bytes the handed-out blocks held)
    od = orc.distinct_bits64(keys, kval)
    assert (first[9][5], first[9][6]) == (od.distinct, od.groups_once)
    assert first[8][5] == n
    # trim: the driver gets the memory back
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    cached = T.cache_stats().device_cached_bytes
    T.trim()
    s3 = T.cache_stats()
    assert s3.device_cached_bytes == 0 and s3.device_cached_blocks == 0 and s3.pinned_cached_bytes == 0
    free_after, _ = torch.cuda.mem_get_info()
    assert free_after - free_before >= cached // 2
    # and the next state simply allocates again
    assert one_run(plan, cols) == first
    assert T.cache_stats().device_misses > s2.device_misses


def test_host_batches_reuse_pinned_blocks():
    rng = np.random.default_rng(12)
    n = 200_000
    keys, kval = make_i64(rng, n, 0, 50_000, 0.1)
    T.init()
    plan = T.Plan([spec(T.COUNT, 0), spec(T.NUMERIC_STATS, 0), spec(T.DISTINCT, 0)])

    def run():
        st = T.State(plan)
        for lo in range(0, n, 8192):   # DataFusion-sized HOST batches: pinned arenas, descriptor tables
            hi = min(n, lo + 8192)
            st.update([numeric_column(keys, kval, False, offset=lo, length=hi - lo)])
        r = st.finalize()
        del st
        return [(x.total, x.non_null, x.distinct, x.sum_i) for x in r]

    a = run()
    s1 = T.cache_stats()
    b = run()
    s2 = T.cache_stats()
    assert a == b and s2.pinned_misses == s1.pinned_misses and s2.pinned_hits > s1.pinned_hits
    assert s2.device_misses == s1.device_misses
    assert a[2][2] == orc.distinct_bits64(keys, kval).distinct


def test_blocks_handed_out_with_stale_contents_change_nothing():
    """TGX_DEVICE_CACHE_POISON=1 fills every block the library hands out with 0xA5 (fresh ones too): the differential
    tester's plans -- every kind of check, HOST and DEVICE batches, merges, blobs -- against the oracle in a child."""
    env = dict(os.environ, TGX_DEVICE_CACHE_POISON="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_device.py"), "--first", "5150", "--count", "40",
                        "--max-rows", "1200000", "--seed-timeout", "90"], env=env, capture_output=True, text=True, timeout=540)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    assert "40 cases of 40 selected" in p.stdout and ", 0 failed" in p.stdout, p.stdout[-800:]
