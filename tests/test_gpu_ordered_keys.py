"""-m gpu: the dense uniqueness pass over keys that arrive IN ORDER (ids that grow with the row number, timestamps).

The rows a wave holds then share a bucket, so partition_kernel runs its CLUSTERED form (kernels/distinct.hip: a tile
whose keys span less than 2^20 values is OR-ed into a bitmap of that stretch in LDS and from there into the global
bitmap -- no lists; otherwise one LDS add per wave instead of 64 on one address, long runs streamed out by the whole
workgroup, bucket_apply_kernel merging the bits of neighbouring lanes) once partition_init_kernel's probe has seen it.  Whatever the order of the
rows, the answers are the oracle's (TG/constraints/uniqueness.rs:568-700 counts; statistics.rs:259-299 aggregates):
bit-exact."""
import zlib

import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import numeric_column, run_plan
from test_gpu_parity import check_stats

pytestmark = pytest.mark.gpu

N = 3_000_017  # (>= 2^20 rows: the partitioned pass)


def shapes(rng, shape):
    validity = None
    if shape == "ascending":
        vals = np.arange(N, dtype=np.int64) * 1 + 10**12
    elif shape == "ascending_step8":        # a tile spans 2^18 values: OR-ed into an LDS bitmap of the stretch
        vals = np.arange(N, dtype=np.int64) * 8 - 12345
    elif shape == "ascending_step64":       # a tile spans 2^21 values: too long for that, sorted into lists
        vals = np.arange(N, dtype=np.int64) * 64
    elif shape == "descending":
        vals = (10**9 - np.arange(N, dtype=np.int64))
    elif shape == "runs_of_three":          # every key three times, in order
        vals = np.arange(N, dtype=np.int64) // 3 - 500_000
    elif shape == "ascending_with_nulls":
        vals = np.arange(N, dtype=np.int64) * 2
        validity = orc.pack_validity(rng.random(N) >= 0.1)
    elif shape == "sorted_blocks":          # 40 000-row blocks in order, the blocks shuffled: probe says yes, buckets jump
        ids = np.arange(N, dtype=np.int64)
        cut = list(range(0, N, 40_000))
        vals = np.concatenate([ids[cut[k]:cut[k] + 40_000] for k in rng.permutation(len(cut))])
    elif shape == "half_ordered":           # first half in order, second half shuffled: a probe on the fence
        vals = np.arange(N, dtype=np.int64)
        vals[N // 2:] = rng.permutation(vals[N // 2:])
    elif shape == "ordered_with_strays":    # in order, 1 % of the rows hold keys from anywhere in the range
        vals = np.arange(N, dtype=np.int64)
        stray = rng.random(N) < 0.01
        vals[stray] = rng.integers(0, N, size=int(stray.sum()), dtype=np.int64)
    else:
        raise AssertionError(shape)
    return vals, validity


@pytest.mark.parametrize("mult", [False, True])
@pytest.mark.parametrize("shape", ["ascending", "ascending_step8", "ascending_step64", "descending", "runs_of_three", "ascending_with_nulls", "sorted_blocks",
                                   "half_ordered", "ordered_with_strays"])
def test_keys_in_order(shape, mult):
    rng = np.random.default_rng(zlib.crc32(shape.encode()))
    vals, validity = shapes(rng, shape)
    flags = T.FLAG_MULTIPLICITY if mult else 0
    res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=flags), spec(T.NUMERIC_STATS, 0)],
                         [[numeric_column(vals, validity, True)]], hint=N)
    d = orc.distinct_bits64(vals.view(np.uint64), validity, n=N)
    assert (res[0].total, res[0].non_null, res[0].distinct) == (d.total, d.non_null, d.distinct)
    if mult:
        assert res[0].groups_once == d.groups_once
    check_stats(res[1], orc.stats(vals, validity))


@pytest.mark.parametrize("stats", [False, True])
def test_later_batches_of_keys_that_grow(stats):
    """Three DEVICE batches of ids that keep growing: the first lays the bitmap out from its sample, the others lie
    outside it as a whole -- counted as outliers (their share of MIN / MAX / SUM collected per workgroup), repaired
    when the state is read."""
    ids = np.arange(3 * N, dtype=np.int64) * 1 - 7
    ids[N + 5] = ids[3]                      # one duplicate across batches
    specs = [spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)] + ([spec(T.NUMERIC_STATS, 0)] if stats else [])
    batches = [[numeric_column(ids, None, True, offset=k * N, length=N)] for k in range(3)]
    res, _, _ = run_plan(specs, batches)
    d = orc.distinct_bits64(ids.view(np.uint64))
    assert (res[0].total, res[0].distinct, res[0].groups_once) == (d.total, d.distinct, d.groups_once)
    if stats:
        check_stats(res[1], orc.stats(ids))


@pytest.mark.parametrize("mult", [False, True])
def test_a_reset_state_remembers_the_range_it_sampled(mult):
    """A state that is reset keeps the extremes its key column's sample showed and lays the next table's bitmap out over
    them without a sample (no read-back at the start of the step).  The next table may be anything: the same ids
    shuffled, ids from elsewhere (every key an outlier: repaired, the range forgotten), a slightly wider range, sparse
    keys -- the counts are the oracle's each time."""
    rng = np.random.default_rng(77 + mult)
    n = 1_500_000
    tables = [
        rng.permutation(n).astype(np.int64),                                 # the range is learned
        rng.permutation(n).astype(np.int64),                                 # ... and fits
        rng.permutation(n).astype(np.int64) + 10**9,                         # nothing fits
        rng.permutation(n).astype(np.int64) + 10**9,                         # (learned anew)
        np.concatenate([rng.permutation(n - 5).astype(np.int64) + 10**9,     # five keys far outside
                        np.array([-3, 5, 2 * 10**9, 2 * 10**9, 7], dtype=np.int64)]),
        rng.integers(-2**62, 2**62, size=n, dtype=np.int64),                 # no dense range at all
        rng.integers(0, n // 10, size=n, dtype=np.int64),                    # dense again, heavy repeats
    ]
    flags = T.FLAG_MULTIPLICITY if mult else 0
    T.init()
    plan = T.Plan([spec(T.DISTINCT, 0, flags=flags), spec(T.NUMERIC_STATS, 0)])
    st = T.State(plan)
    for vals in tables:
        st.reset()
        cols = [numeric_column(vals, None, True)]
        st.update(cols)
        res = st.finalize()
        d = orc.distinct_bits64(vals.view(np.uint64), None, n=len(vals))
        assert (res[0].total, res[0].non_null, res[0].distinct) == (d.total, d.non_null, d.distinct)
        if mult:
            assert res[0].groups_once == d.groups_once
        check_stats(res[1], orc.stats(vals, None))
