//! An Arrow array as the `tgx_column` view the kernels read in place (include/tgx.h, "column views").
//!
//! The view borrows the array's buffers: keep the `ColumnView` (and through it the array) alive until `State::update`
//! has returned.  DataFusion reads Parquet strings as `Utf8View` and streams 8192-row batches
//! (term-guard/src/core/context.rs:28-38): both are taken as they come, small batches are coalesced by the library.
use crate::sys::*;
use arrow::array::{make_array, Array, ArrayRef};
use arrow::buffer::Buffer;
use arrow::datatypes::DataType;
use std::ptr;

/// A `tgx_column` together with what it points into.
pub struct ColumnView {
    pub raw: tgx_column,
    _array: ArrayRef,
    _dictionary: Option<Box<ColumnView>>,
    _variadic: Vec<*const u8>,
    _variadic_sizes: Vec<i64>,
    _realigned_validity: Option<Buffer>,
    _offsets: Vec<i32>, // fixed-width values presented as w-byte strings: the offsets 0, w, 2 w, ..
}

impl ColumnView {
    /// The view as `TGX_MEM_HOST_RETAINED`: the caller promises to keep the array (this `ColumnView`) alive and
    /// unmodified until the library has run the batch (`State::pending`), and the library may copy it at the flush.
    pub fn retained(&self) -> tgx_column {
        let mut c = self.raw;
        c.mem = TGX_MEM_HOST_RETAINED;
        c
    }
}

/// `None`: a type outside the path (nested types, intervals, Float16 ..) -- evaluate the constraint with the stock SQL
/// (or, for a column only completeness / size look at, hand over `validity_only_view`).
pub fn column_view(arr: &ArrayRef) -> Option<ColumnView> {
    let d = arr.to_data();
    let mut view = ColumnView {
        raw: tgx_column {
            type_: 0,
            mem: TGX_MEM_HOST,
            length: d.len() as i64,
            offset: d.offset() as i64,
            null_count: d.null_count() as i64,
            validity: ptr::null(),
            values: ptr::null(),
            offsets: ptr::null(),
            data: ptr::null(),
            dictionary: ptr::null(),
            variadic: ptr::null(),
            variadic_sizes: ptr::null(),
            n_variadic: 0,
            reserved: 0,
        },
        _array: arr.clone(),
        _dictionary: None,
        _variadic: Vec::new(),
        _variadic_sizes: Vec::new(),
        _realigned_validity: None,
        _offsets: Vec::new(),
    };
    // tgx applies ONE offset to validity bits and value slots alike; a NullBuffer carries its own bit offset
    if let Some(nulls) = d.nulls() {
        if nulls.offset() == d.offset() {
            view.raw.validity = nulls.validity().as_ptr();
        } else {
            // re-align: bit `offset + i` of the new bitmap is the validity of slot i
            let bits = nulls.inner().sliced();
            let mut aligned = arrow::buffer::MutableBuffer::new_null(d.offset() + d.len());
            for i in 0..d.len() {
                if arrow::util::bit_util::get_bit(bits.as_slice(), i) {
                    arrow::util::bit_util::set_bit(aligned.as_slice_mut(), d.offset() + i);
                }
            }
            let buf: Buffer = aligned.into();
            view.raw.validity = buf.as_ptr();
            view._realigned_validity = Some(buf);
        }
    }
    let first = |i: usize| d.buffers()[i].as_ptr();
    match d.data_type() {
        DataType::Int64 | DataType::Timestamp(_, _) | DataType::Date64 | DataType::Time64(_) | DataType::Duration(_) => {
            view.raw.type_ = TGX_INT64;
            view.raw.values = first(0) as _;
        }
        DataType::Float64 => {
            view.raw.type_ = TGX_FLOAT64;
            view.raw.values = first(0) as _;
        }
        // 4-byte numerics: read in place by COUNT / NUMERIC_STATS, widened on the device for the other checks
        DataType::Int32 | DataType::Date32 | DataType::Time32(_) => {
            view.raw.type_ = TGX_INT32;
            view.raw.values = first(0) as _;
        }
        DataType::Float32 => {
            view.raw.type_ = TGX_FLOAT32;
            view.raw.values = first(0) as _;
        }
        // narrow and unsigned integers: widened to Int64 on the device (completeness / uniqueness take any column type:
        // completeness.rs:158-163, uniqueness.rs:612-617; what statistics the stock constraint can read off such a
        // column is the planner's `reference_extracts`)
        DataType::Int8 | DataType::Int16 | DataType::UInt8 | DataType::UInt16 | DataType::UInt32 => {
            view.raw.type_ = match d.data_type() {
                DataType::Int8 => TGX_INT8,
                DataType::Int16 => TGX_INT16,
                DataType::UInt8 => TGX_UINT8,
                DataType::UInt16 => TGX_UINT16,
                _ => TGX_UINT32,
            };
            view.raw.values = first(0) as _;
        }
        // COUNT / DISTINCT only (the library answers TGX_UNSUPPORTED to anything else: the binding stays unplanned)
        DataType::UInt64 => {
            view.raw.type_ = TGX_UINT64;
            view.raw.values = first(0) as _;
        }
        DataType::Boolean => {
            // the values buffer is bit-packed; ArrayData's offset applies to it as to the validity bitmap
            view.raw.type_ = TGX_BOOL;
            view.raw.values = first(0) as _;
        }
        DataType::Utf8 => {
            view.raw.type_ = TGX_UTF8;
            view.raw.offsets = first(0) as _;
            view.raw.data = first(1);
        }
        DataType::LargeUtf8 => {
            view.raw.type_ = TGX_LARGE_UTF8;
            view.raw.offsets = first(0) as _;
            view.raw.data = first(1);
        }
        DataType::Utf8View | DataType::BinaryView => {
            view.raw.type_ = TGX_UTF8_VIEW;
            view.raw.values = first(0) as _;
            for b in &d.buffers()[1..] {
                view._variadic.push(b.as_ptr());
                view._variadic_sizes.push(b.len() as i64);
            }
            view.raw.n_variadic = view._variadic.len() as i32;
            view.raw.variadic = view._variadic.as_ptr();
            view.raw.variadic_sizes = view._variadic_sizes.as_ptr();
        }
        DataType::Dictionary(k, v)
            if **k == DataType::Int32 && matches!(**v, DataType::Utf8 | DataType::LargeUtf8) =>
        {
            view.raw.type_ = TGX_DICT32_UTF8;
            view.raw.values = first(0) as _;
            let mut dict = Box::new(column_view(&make_array(d.child_data()[0].clone()))?);
            // (the dictionary lives as long as this view does: with `retained()` indices the whole column is kept; with
            //  the plain `raw` the library sees a batch that mixes the two and copies it before `update` returns)
            dict.raw.mem = TGX_MEM_HOST_RETAINED;
            view.raw.dictionary = &dict.raw as *const tgx_column;
            view._dictionary = Some(dict);
        }
        // Binary / LargeBinary / BinaryView have the string layouts: COUNT and COUNT(DISTINCT) compare bytes, as the
        // reference's SQL does (the planner plans no pattern or LENGTH check on them: `string_typed`)
        DataType::Binary => {
            view.raw.type_ = TGX_UTF8;
            view.raw.offsets = first(0) as _;
            view.raw.data = first(1);
        }
        DataType::LargeBinary => {
            view.raw.type_ = TGX_LARGE_UTF8;
            view.raw.offsets = first(0) as _;
            view.raw.data = first(1);
        }
        // fixed-width values of w bytes (FixedSizeBinary(w), Decimal128 = 16, Decimal256 = 32): one precision / scale per
        // column, so equal values are equal bytes and COUNT(DISTINCT) sees them as w-byte strings; the offsets are made
        // here (4 bytes per row), the values are not copied.  The validity bitmap keeps the array's offset, so the
        // offsets start at slot `offset` too.
        DataType::FixedSizeBinary(_) | DataType::Decimal128(_, _) | DataType::Decimal256(_, _) => {
            let w = match d.data_type() {
                DataType::FixedSizeBinary(w) => *w as i64,
                DataType::Decimal128(_, _) => 16,
                _ => 32,
            };
            let slots = (d.offset() + d.len() + 1) as i64;
            if slots * w >= i32::MAX as i64 {
                return None;
            }
            view._offsets = (0..slots).map(|i| (i * w) as i32).collect();
            view.raw.type_ = TGX_UTF8;
            view.raw.offsets = view._offsets.as_ptr() as _;
            view.raw.data = first(0);
        }
        _ => return None,
    }
    Some(view)
}

/// ANY array for the checks that read no values -- completeness and size need the validity bitmap and the length only
/// (`SELECT COUNT(*), COUNT(c)` takes every column type: constraints/completeness.rs:158-163): lists, structs, maps,
/// intervals .. no longer send a whole run back to SQL because one of them has an `is_complete` check.
pub fn validity_only_view(arr: &ArrayRef) -> ColumnView {
    let d = arr.to_data();
    let mut raw: tgx_column = unsafe { std::mem::zeroed() };
    raw.type_ = TGX_INT64;
    raw.mem = TGX_MEM_HOST;
    raw.length = d.len() as i64;
    raw.offset = d.offset() as i64;
    raw.null_count = d.null_count() as i64;
    let mut realigned = None;
    if let Some(nulls) = d.nulls() {
        if nulls.offset() == d.offset() {
            raw.validity = nulls.validity().as_ptr();
        } else {
            let bits = nulls.inner().sliced();
            let mut aligned = arrow::buffer::MutableBuffer::new_null(d.offset() + d.len());
            for i in 0..d.len() {
                if arrow::util::bit_util::get_bit(bits.as_slice(), i) {
                    arrow::util::bit_util::set_bit(aligned.as_slice_mut(), d.offset() + i);
                }
            }
            let buf: Buffer = aligned.into();
            raw.validity = buf.as_ptr();
            realigned = Some(buf);
        }
    }
    ColumnView {
        raw,
        _array: arr.clone(),
        _dictionary: None,
        _variadic: Vec::new(),
        _variadic_sizes: Vec::new(),
        _realigned_validity: realigned,
        _offsets: Vec::new(),
    }
}

/// Do pattern / length / containment checks apply to this type?  (The reference's `~` and `LENGTH` take strings.)
pub fn string_typed(t: &DataType) -> bool {
    match t {
        DataType::Utf8 | DataType::LargeUtf8 | DataType::Utf8View => true,
        DataType::Dictionary(_, v) => matches!(**v, DataType::Utf8 | DataType::LargeUtf8),
        _ => false,
    }
}

/// A column the plan does not read: zeroed (include/tgx.h, tgx_update).
pub fn unused_column() -> tgx_column {
    unsafe { std::mem::zeroed() }
}
