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
