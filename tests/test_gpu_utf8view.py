"""-m gpu: checks on Utf8View columns (the layout DataFusion reads Parquet strings as) vs the oracle on the decoded
values: 16-byte views with inline short strings and out-of-line long ones spread over several data buffers."""
import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import pad_validity, run_plan, to_device
from test_gpu_strings import check, make_strings

pytestmark = pytest.mark.gpu

EMAIL = r"^[A-Za-z0-9._%+-]+@[A-Za-z0-9.-]+\.[A-Za-z]{2,}$"


def encode_views(vals, rng, n_buffers=3, junk_null_views=True):
    """list[str|None] -> (views uint8[16 n], [data buffers], validity or None)"""
    n = len(vals)
    views = np.zeros((n, 16), dtype=np.uint8)
    bufs = [bytearray(rng.integers(0, 255, size=int(rng.integers(0, 40)), dtype=np.uint8).tobytes())
            for _ in range(n_buffers)]  # some leading junk so offsets are not 0
    validity = np.zeros((n + 7) // 8, np.uint8)
    any_null = False
    for i, v in enumerate(vals):
        if v is None:
            any_null = True
            if junk_null_views:  # the view of a NULL slot is arbitrary: a huge length and a wild buffer index
                views[i] = np.frombuffer(np.array([2**31 - 1, -1, 2**31 - 1, -1], dtype=np.int32).tobytes(), np.uint8)
            continue
        validity[i >> 3] |= 1 << (i & 7)
        b = bytes(v) if isinstance(v, (bytes, bytearray)) else v.encode("utf-8")
        views[i, 0:4] = np.frombuffer(np.int32(len(b)).tobytes(), np.uint8)
        if len(b) <= 12:
            views[i, 4:4 + len(b)] = np.frombuffer(b, np.uint8)
        else:
            k = int(rng.integers(0, n_buffers))
            off = len(bufs[k])
            bufs[k] += b
            views[i, 4:8] = np.frombuffer(b[:4], np.uint8)
            views[i, 8:12] = np.frombuffer(np.int32(k).tobytes(), np.uint8)
            views[i, 12:16] = np.frombuffer(np.int32(off).tobytes(), np.uint8)
    data = [np.frombuffer(bytes(b) + b"\0" * 16, dtype=np.uint8).copy() for b in bufs]
    return views.reshape(-1), data, (validity if any_null else None)


def view_column(vals, rng, device, offset=0, length=None, **kw):
    views, data, validity = encode_views(vals, rng, **kw)
    v = pad_validity(validity) if validity is not None else None
    if device:
        views, data, v = to_device(views), [to_device(d) for d in data], (to_device(v) if v is not None else None)
    n = (len(vals) - offset) if length is None else length
    return T.Column.utf8_view(views, data, validity=v, length=n, offset=offset)


@pytest.mark.parametrize("device", [True, False])
@pytest.mark.parametrize("n,card", [(2000, 100), (200_000, 30_000)])
def test_view_equals_plain(n, card, device):
    rng = np.random.default_rng(n + card + device)
    vals = make_strings(rng, n, card)
    offs, data, validity = orc.utf8_from_list(vals)
    want_d = orc.distinct_utf8(offs, data, validity)
    for flags in (0, T.FLAG_TRIM | T.FLAG_NULL_IS_VALID):
        want_m = orc.Regex(EMAIL).count_utf8(offs, data, validity, trim=bool(flags & T.FLAG_TRIM),
                                             null_is_valid=bool(flags & T.FLAG_NULL_IS_VALID)).matches
        res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.COUNT, 0),
                              spec(T.REGEX_MATCH, 0, flags=flags, pattern=EMAIL)], [[view_column(vals, rng, device)]])
        check(res[0], want_d)
        assert (res[1].total, res[1].non_null) == (n, want_d.non_null)
        assert (res[2].total, res[2].matches) == (n, want_m)


def test_view_slices_batches_and_edge_cases():
    rng = np.random.default_rng(4)
    n = 60_000
    vals = make_strings(rng, n, 9000)
    # exactly 12 / 13 bytes (the inline boundary), empty strings, multi-byte characters across the boundary
    vals[:6] = ["123456789012", "1234567890123", "", "ÿÿÿÿÿÿ", "ÿÿÿÿÿÿx", None]
    offs, data, validity = orc.utf8_from_list(vals)
    want_d = orc.distinct_utf8(offs, data, validity)
    want_m = orc.Regex(r"^\d{12,13}$|^ÿ+x?$").count_utf8(offs, data, validity, null_is_valid=False).matches
    cuts = [0, 3, 20_000, 20_000, n]
    batches = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        lead = int(rng.integers(0, 100))
        batches.append([view_column([None] * lead + vals[lo:hi], rng, True, offset=lead, length=hi - lo)])
    res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY),
                          spec(T.REGEX_MATCH, 0, pattern=r"^\d{12,13}$|^ÿ+x?$")], batches)
    check(res[0], want_d)
    assert (res[1].total, res[1].matches) == (n, want_m)
    # empty batch; all-NULL batch with junk views and no data buffers at all
    res, _, _ = run_plan([spec(T.DISTINCT, 0), spec(T.REGEX_MATCH, 0, pattern="x", flags=T.FLAG_NULL_IS_VALID)],
                         [[view_column([], rng, True)], [view_column([None] * 500, rng, True, n_buffers=0)]])
    assert (res[0].total, res[0].non_null, res[0].distinct, res[1].total, res[1].matches) == (500, 0, 0, 500, 500)


def test_from_arrow_string_view():
    pa = pytest.importorskip("pyarrow")
    if not hasattr(pa, "string_view"):
        pytest.skip("pyarrow without string_view")
    rng = np.random.default_rng(8)
    vals = make_strings(rng, 30_000, 5000)
    arr = pa.array(vals, type=pa.string_view())
    offs, data, validity = orc.utf8_from_list(vals)
    want_d = orc.distinct_utf8(offs, data, validity)
    want_m = orc.Regex(EMAIL).count_utf8(offs, data, validity, null_is_valid=False).matches
    res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY), spec(T.REGEX_MATCH, 0, pattern=EMAIL)],
                         [[T.Column.from_arrow(arr)]])
    check(res[0], want_d)
    assert res[1].matches == want_m
    sl = arr.slice(777, 20_000)
    o2, d2, v2 = orc.utf8_from_list(vals[777:20_777])
    res, _, _ = run_plan([spec(T.DISTINCT, 0, flags=T.FLAG_MULTIPLICITY)], [[T.Column.from_arrow(sl)]])
    check(res[0], orc.distinct_utf8(o2, d2, v2))
