"""CPU: what Column.from_arrow / Column.validity_only make of pyarrow arrays whose types have no tgx_type of their own
(round 5): binary layouts under the string types, fixed-width values as w-byte strings under synthetic offsets (no copy
of the values), any other type as validity + length.  The device side of the same arrays: tests/test_gpu_narrow_types.py."""
import decimal

import numpy as np
import pytest

import term_amd as T

pa = pytest.importorskip("pyarrow")


def test_binary_layouts_are_the_string_layouts():
    vals = [b"ab", None, b"", b"\xff\x00z"]
    assert T.Column.from_arrow(pa.array(vals, pa.binary())).c.type == T.UTF8
    assert T.Column.from_arrow(pa.array(vals, pa.large_binary())).c.type == T.LARGE_UTF8
    assert T.Column.from_arrow(pa.array(vals, pa.binary_view())).c.type == T.UTF8_VIEW


@pytest.mark.parametrize("arr,width", [
    (pa.array([b"abcd", None, b"wxyz", b"0000", b"1111"], pa.binary(4)), 4),
    (pa.array([decimal.Decimal("1.25"), None, decimal.Decimal("-3.00"), decimal.Decimal("0"), decimal.Decimal("7.5")],
              pa.decimal128(10, 2)), 16),
    (pa.array([decimal.Decimal(10) ** 40, None, decimal.Decimal(-1), decimal.Decimal(0), decimal.Decimal(5)],
              pa.decimal256(50, 0)), 32),
])
def test_fixed_width_values_under_synthetic_offsets(arr, width):
    for view in (arr, arr.slice(1, 3), arr.slice(2)):
        col = T.Column.from_arrow(view)
        raw = col.c
        assert (raw.type, raw.length, raw.null_count) == (T.UTF8, len(view), view.null_count)
        offs = np.ctypeslib.as_array(T._lib.C.cast(raw.offsets, T._lib.C.POINTER(T._lib.C.c_int32)), shape=(raw.offset + raw.length + 1,))
        got = offs[raw.offset: raw.offset + raw.length + 1]
        assert list(np.diff(got)) == [width] * len(view)       # every value is `width` bytes ..
        assert got[0] == view.offset * width                   # .. and row 0 is where the slice starts in the buffer
        assert raw.data == view.buffers()[1].address           # the values are NOT copied


def test_any_type_as_validity_and_length():
    lists = pa.array([[1], None, [2, 3], None, []], pa.list_(pa.int64()))
    for view in (lists, lists.slice(1, 3)):
        col = T.Column.validity_only(view)
        assert (col.c.type, col.c.length, col.c.null_count, col.c.offset) == (T.INT64, len(view), view.null_count, view.offset)
        assert not col.c.values and col.c.validity
    col = T.Column.validity_only(pa.nulls(11))
    assert (col.c.length, col.c.null_count) == (11, 11) and col.c.validity
    col = T.Column.validity_only(pa.array([[1], [2]], pa.list_(pa.int64())))
    assert col.c.null_count == 0 and not col.c.validity


def test_record_batches_are_described_in_place():
    """suite._flatten_table over a pyarrow table of many record batches fills the column structs from the buffers'
    addresses (a Column object per (batch, column) cost 11 us: more than the library needs for the batch); what it
    writes equals what Column.from_arrow says, for sliced batches with NULLs, and every HOST struct is handed over as
    TGX_MEM_HOST_RETAINED (the host call runs the whole table and returns)."""
    import decimal

    from term_amd import suite as S

    rng = np.random.default_rng(2)
    n = 50_000
    mask = rng.random(n) < 0.1
    t = pa.table({
        "i64": pa.array(rng.integers(0, 100, n), mask=mask),
        "f32": pa.array(rng.standard_normal(n).astype(np.float32), mask=mask),
        "i16": pa.array(rng.integers(0, 100, n).astype(np.int16)),
        "u64": pa.array(rng.integers(0, 100, n).astype(np.uint64), mask=mask),
        "ts": pa.array(rng.integers(0, 10**15, n), pa.timestamp("us")),
        "d32": pa.array(rng.integers(0, 20000, n).astype(np.int32), pa.date32()),
        "s": pa.array(["x%d" % i for i in range(n)], mask=mask),
        "ls": pa.array(["y%d" % i for i in range(n)], pa.large_string()),
        "bin": pa.array([b"z%d" % i for i in range(n)], pa.binary(), mask=mask),
        "b": pa.array(rng.random(n) < 0.5, mask=mask),
        "dict": pa.array(["a", "b", "c"] * (n // 3) + ["a"] * (n % 3)).dictionary_encode(),
        "dec": pa.array([None if m else decimal.Decimal(i) for i, m in zip(range(n), mask)], pa.decimal128(12, 0)),
        "lst": pa.array([[1]] * n, pa.list_(pa.int8())),
        "sv": pa.array(["v%d" % i for i in range(n)], pa.string_view()),
    })
    tab = pa.Table.from_batches(t.slice(3, n - 10).to_batches(max_chunksize=8192))
    names, n_cols, arr, n_batches, _keep = S._flatten_table(tab)
    assert (n_cols, n_batches) == (14, 7)
    lean = {"i64", "f32", "i16", "u64", "ts", "d32", "s", "ls", "bin"}

    def long_way(a):
        try:
            return T.Column.from_arrow(a)
        except T.TgxError:
            return T.Column.validity_only(a)

    k = 0
    for rb in tab.to_batches():
        for ci in range(n_cols):
            want = long_way(rb.column(ci)).c
            got = arr[k]
            k += 1
            name = names[ci].decode()
            assert (got.type, got.length, got.offset, got.null_count, got.n_variadic) == \
                   (want.type, want.length, want.offset, want.null_count, want.n_variadic), name
            if name in lean:  # (the other layouts may own fresh helper buffers per call: realigned bits, synthetic offsets)
                assert (got.values, got.offsets, got.data, got.validity) == (want.values, want.offsets, want.data, want.validity), name
            assert got.mem == T.MEM_HOST_RETAINED, name
            if got.dictionary:
                assert got.dictionary.contents.mem == T.MEM_HOST_RETAINED
