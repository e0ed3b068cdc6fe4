"""ctypes binding of libtgx.so (include/tgx.h).  Fails loudly when the library is missing."""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)

# enums of include/tgx.h
COUNT, NUMERIC_STATS, DISTINCT, REGEX_MATCH, KLL, COMOMENTS, SPEARMAN, LENGTH = 1, 2, 3, 4, 5, 6, 7, 8
APPROX_DISTINCT = 9
FLAG_VARIANCE, FLAG_MULTIPLICITY, FLAG_TRIM, FLAG_CASE_INSENSITIVE, FLAG_NULL_IS_VALID = 1, 2, 4, 8, 16
FLAG_EXACT_RANK_SUMS = 32
FLAG_EXACT_KEYS = 64  # DISTINCT over string / tuple keys: equal fingerprints confirmed byte by byte
ABI_VERSION = 6  # include/tgx.h TGX_ABI_VERSION: the struct layouts below
INT64, FLOAT64, UTF8, LARGE_UTF8, DICT32_UTF8, UTF8_VIEW, INT32, FLOAT32 = 1, 2, 3, 4, 5, 6, 7, 8
INT8, INT16, UINT8, UINT16, UINT32, UINT64, BOOL = 9, 10, 11, 12, 13, 14, 15  # (include/tgx.h: narrow / unsigned / Boolean)
MEM_HOST, MEM_DEVICE, MEM_HOST_RETAINED = 0, 1, 2
STATUS_NAMES = {0: "TGX_OK", 1: "TGX_INVALID_ARGUMENT", 2: "TGX_UNSUPPORTED", 3: "TGX_DEVICE_ERROR",
                4: "TGX_OUT_OF_MEMORY", 5: "TGX_INTERNAL", 6: "TGX_NO_DEVICE"}


class TgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (STATUS_NAMES.get(code, code), msg))
        self.code = code
        self.status = STATUS_NAMES.get(code, str(code))
        self.msg = msg


class _Error(C.Structure):
    _fields_ = [("code", C.c_int32), ("msg", C.c_char * 256)]


class _Column(C.Structure):
    pass


_Column._fields_ = [
    ("type", C.c_int32), ("mem", C.c_int32), ("length", C.c_int64), ("offset", C.c_int64),
    ("null_count", C.c_int64), ("validity", C.c_void_p), ("values", C.c_void_p), ("offsets", C.c_void_p),
    ("data", C.c_void_p), ("dictionary", C.POINTER(_Column)),
    ("variadic", C.POINTER(C.c_void_p)), ("variadic_sizes", C.POINTER(C.c_int64)), ("n_variadic", C.c_int32),
    ("reserved", C.c_int32),
]


class CheckSpec(C.Structure):
    _fields_ = [("kind", C.c_int32), ("column", C.c_int32), ("column2", C.c_int32), ("flags", C.c_uint32),
                ("pattern", C.c_char_p), ("pattern_len", C.c_uint64), ("kll_k", C.c_uint32),
                ("reserved", C.c_uint32), ("columns", C.POINTER(C.c_int32)), ("n_columns", C.c_uint32),
                ("reserved2", C.c_uint32), ("length_min", C.c_uint64), ("length_max", C.c_uint64)]


class Result(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("is_float", C.c_int32), ("total", C.c_int64), ("non_null", C.c_int64),
        ("has_value", C.c_int32), ("has_variance", C.c_int32), ("min_i", C.c_int64), ("max_i", C.c_int64),
        ("min_f", C.c_double), ("max_f", C.c_double), ("sum_i", C.c_int64), ("sum_f", C.c_double),
        ("mean", C.c_double), ("var_samp", C.c_double), ("stddev_samp", C.c_double),
        ("distinct", C.c_int64), ("groups_once", C.c_int64), ("matches", C.c_int64),
        ("sum_x", C.c_double), ("sum_y", C.c_double), ("sum_x2", C.c_double), ("sum_y2", C.c_double),
        ("sum_xy", C.c_double), ("kll_n", C.c_uint64),
        ("co_mean_x", C.c_double), ("co_mean_y", C.c_double), ("co_m2_x", C.c_double), ("co_m2_y", C.c_double),
        ("co_c_xy", C.c_double),
    ]


class _Options(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("flags", C.c_uint32), ("distinct_capacity_hint", C.c_uint64)]


def lib_path():
    # TGX_LIB: another build of the same library (the host-side sanitizer build, tools/run_host_asan.sh)
    return os.environ.get("TGX_LIB") or os.path.join(_HERE, "libtgx.so")


def abi_symbols():
    """Every function name include/tgx.h declares (parsed from the header)."""
    names = set()
    for header in ("tgx.h", "tgx_host.h"):
        with open(os.path.join(_ROOT, "include", header)) as f:
            text = f.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(tgx_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


_LIB = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch's wheel carries its own libamdhip64.so (same SONAME as
    /opt/rocm's); if libtgx.so pulled in the system copy first and torch its own copy later, the second
    runtime to initialise finds no device.  Loading torch's copy first makes both resolve to it."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        found = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        found = None
    if found is None or not found.origin:
        return
    cand = os.path.join(os.path.dirname(found.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError(
                "term_amd/libtgx.so is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C term_amd/csrc` (hipcc, gfx950). term_amd has no CPU fallback.")
        _preload_hip_runtime()
        L = C.CDLL(path)
        vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
        E = C.POINTER(_Error)
        L.tgx_abi_version.restype = C.c_uint32
        if L.tgx_abi_version() != ABI_VERSION:
            raise ImportError("term_amd/libtgx.so speaks ABI %d, this binding ABI %d: rebuild it (make -C term_amd/csrc)"
                              % (L.tgx_abi_version(), ABI_VERSION))
        L.tgx_status_name.restype = C.c_char_p
        L.tgx_status_name.argtypes = [C.c_int32]
        L.tgx_init.argtypes = [C.POINTER(_Options), E]
        L.tgx_plan_create.argtypes = [C.POINTER(CheckSpec), sz, C.POINTER(vp), E]
        L.tgx_plan_destroy.argtypes = [vp]
        L.tgx_plan_destroy.restype = None
        L.tgx_plan_num_specs.argtypes = [vp]
        L.tgx_plan_num_specs.restype = sz
        L.tgx_blob_fingerprint_key.argtypes = [C.c_char_p, sz, C.POINTER(C.c_uint8), C.POINTER(C.c_int32)]
        L.tgx_plan_set_fingerprint_key.argtypes = [vp, C.c_char_p, E]
        L.tgx_plan_get_fingerprint_key.argtypes = [vp, C.POINTER(C.c_uint8)]
        L.tgx_state_create.argtypes = [vp, vp, C.POINTER(vp), E]
        L.tgx_state_destroy.argtypes = [vp]
        L.tgx_state_destroy.restype = None
        L.tgx_update.argtypes = [vp, vp, C.POINTER(_Column), sz, E]
        L.tgx_merge.argtypes = [vp, vp, C.POINTER(vp), sz, E]
        L.tgx_finalize.argtypes = [vp, vp, C.POINTER(Result), sz, E]
        L.tgx_comm_create.argtypes = [C.POINTER(CommOps), C.POINTER(vp), E]
        L.tgx_comm_rccl_unique_id.argtypes = [vp, E]
        L.tgx_comm_create_rccl.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(vp), E]
        L.tgx_comm_adopt_rccl.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(vp), E]
        L.tgx_comm_destroy.argtypes = [vp]
        L.tgx_comm_destroy.restype = None
        L.tgx_allreduce.argtypes = [vp, vp, vp, E]
        L.tgx_state_sync.argtypes = [vp, E]
        L.tgx_state_reset.argtypes = [vp, vp, E]
        L.tgx_state_serialize.argtypes = [vp, vp, vp, sz, C.POINTER(sz), E]
        L.tgx_state_deserialize.argtypes = [vp, vp, sz, C.POINTER(vp), E]
        L.tgx_kll_quantile.argtypes = [vp, vp, sz, C.c_double, C.POINTER(C.c_double), E]
        L.tgx_kll_summary.argtypes = [vp, vp, sz, C.POINTER(u64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(u64), C.POINTER(u64), E]
        L.tgx_kll_level_items.argtypes = [vp, vp, sz, u64, vp, u64, C.POINTER(u64), E]
        L.tgx_kll_relative_error_bound.argtypes = [C.c_uint32]
        L.tgx_kll_relative_error_bound.restype = C.c_double
        L.tgx_distinct_export.argtypes = [vp, vp, sz, C.c_uint32, C.POINTER(vp), C.POINTER(u64), E]
        L.tgx_distinct_import.argtypes = [vp, vp, sz, vp, u64, E]
        L.tgx_distinct_range_hint.argtypes = [vp, vp, sz, C.c_int64, C.c_int64, E]
        L.tgx_distinct_bitmap_view.argtypes = [vp, vp, sz, C.POINTER(C.c_int64), C.POINTER(u64), C.POINTER(vp), C.POINTER(vp), E]
        L.tgx_distinct_adopt_slices.argtypes = [vp, vp, sz, C.c_int64, vp, vp, C.c_uint32, u64, u64, E]
        L.tgx_distinct_record_bytes.argtypes = [vp, vp, sz]
        L.tgx_distinct_record_bytes.restype = sz
        L.tgx_profile_enable.argtypes = [vp, C.c_int32]
        L.tgx_profile_get.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(u64), E]
        L.tgx_profile_reset.argtypes = [vp]
        L.tgx_regex_validate.argtypes = [C.c_char_p, sz, C.c_uint32, E]
        L.tgx_regex_is_match.argtypes = [C.c_char_p, sz, C.c_uint32, C.c_char_p, sz, C.POINTER(C.c_int32), E]
        L.tgx_cache_stats_get.argtypes = [C.POINTER(CacheStats)]
        L.tgx_state_pending.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        _LIB = L
    return _LIB


class CacheStats(C.Structure):
    """tgx_cache_stats (include/tgx.h)"""
    _fields_ = [("device_cached_bytes", C.c_uint64), ("device_cached_blocks", C.c_uint64), ("device_hits", C.c_uint64),
                ("device_misses", C.c_uint64), ("pinned_cached_bytes", C.c_uint64), ("pinned_hits", C.c_uint64),
                ("pinned_misses", C.c_uint64)]


def cache_stats():
    """tgx_cache_stats_get: what the library's cache of device / pinned blocks holds and has served"""
    out = CacheStats()
    rc = lib().tgx_cache_stats_get(C.byref(out))
    if rc != 0:
        raise TgxError(rc, "tgx_cache_stats_get")
    return out


def trim():
    """tgx_trim: the cached blocks of destroyed states go back to the driver"""
    rc = lib().tgx_trim()
    if rc != 0:
        raise TgxError(rc, "tgx_trim")


def _check(status, err):
    if status != 0:
        raise TgxError(status, err.msg.decode("utf-8", "replace"))


_INITED = False


OPT_NO_COALESCE = 1


def init(device_id=-1, distinct_capacity_hint=0, flags=0):
    """tgx_init: selects the gfx950 device. Raises TgxError(TGX_NO_DEVICE) when there is none.
    flags: OPT_NO_COALESCE = every batch is launched as it arrives"""
    global _INITED
    err = _Error()
    opts = _Options(device_id, flags, distinct_capacity_hint)
    _check(lib().tgx_init(C.byref(opts), C.byref(err)), err)
    _INITED = True


def _ptr_of(buf):
    """address of a numpy array / torch tensor / int / None, plus whether it is device memory"""
    if buf is None:
        return None, None
    if isinstance(buf, int):
        return buf, None
    if hasattr(buf, "data_ptr"):  # torch tensor
        return buf.data_ptr(), buf.is_cuda
    if hasattr(buf, "ctypes"):  # numpy
        return buf.ctypes.data, False
    raise TypeError("unsupported buffer type %r" % type(buf))


class Column:
    """A tgx_column view. Keeps the Python buffers alive while the view exists."""

    def __init__(self, type, length, values=None, validity=None, offsets=None, data=None, offset=0,
                 null_count=-1, mem=None, dictionary=None, variadic=None):
        self._keep = (values, validity, offsets, data, dictionary, variadic)
        c = _Column()
        c.type = type
        c.length = length
        c.offset = offset
        c.null_count = null_count
        spaces = set()
        for name, buf in (("values", values), ("validity", validity), ("offsets", offsets), ("data", data)):
            p, is_dev = _ptr_of(buf)
            setattr(c, name, p)
            if is_dev is not None:
                spaces.add(bool(is_dev))
        if variadic is not None:
            # Utf8View data buffers: a host array of their pointers (+ sizes, needed to stage HOST buffers)
            ptrs = (C.c_void_p * max(1, len(variadic)))()
            sizes = (C.c_int64 * max(1, len(variadic)))()
            for k, buf in enumerate(variadic):
                pk, is_dev = _ptr_of(buf)
                ptrs[k] = pk
                sizes[k] = int(buf.nbytes) if hasattr(buf, "nbytes") else int(buf.numel() * buf.element_size())
                if is_dev is not None:
                    spaces.add(bool(is_dev))
            c.variadic = C.cast(ptrs, C.POINTER(C.c_void_p))
            c.variadic_sizes = C.cast(sizes, C.POINTER(C.c_int64))
            c.n_variadic = len(variadic)
            self._keep += (ptrs, sizes)
        if mem is None:
            if len(spaces) > 1:
                raise ValueError("a column's buffers must all live in one memory space")
            mem = MEM_DEVICE if (spaces and spaces.pop()) else MEM_HOST
        c.mem = mem
        if dictionary is not None:
            c.dictionary = C.pointer(dictionary.c)
        self.c = c

    def sliced(self, offset, length):
        """Arrow's Array::slice: the same buffers viewed from `offset` (relative to this view) for `length` rows"""
        import copy

        if offset < 0 or length < 0 or offset + length > self.c.length:
            raise ValueError("slice [%d, %d) outside a column of %d rows" % (offset, offset + length, self.c.length))
        out = copy.copy(self)  # shares the keep-alive references of the buffers
        c = _Column()
        C.memmove(C.byref(c), C.byref(self.c), C.sizeof(_Column))
        c.offset = self.c.offset + offset
        c.length = length
        c.null_count = -1
        out.c = c
        return out

    @staticmethod
    def int64(values, validity=None, length=None, offset=0):
        n = (len(values) - offset) if length is None else length
        return Column(INT64, n, values=values, validity=validity, offset=offset)

    @staticmethod
    def float64(values, validity=None, length=None, offset=0):
        n = (len(values) - offset) if length is None else length
        return Column(FLOAT64, n, values=values, validity=validity, offset=offset)

    @staticmethod
    def int32(values, validity=None, length=None, offset=0):
        """Int32 / Date32 / Time32: widened to Int64 on the device"""
        n = (len(values) - offset) if length is None else length
        return Column(INT32, n, values=values, validity=validity, offset=offset)

    @staticmethod
    def float32(values, validity=None, length=None, offset=0):
        """Float32: widened to Float64 on the device"""
        n = (len(values) - offset) if length is None else length
        return Column(FLOAT32, n, values=values, validity=validity, offset=offset)

    @staticmethod
    def narrow(type_id, values, validity=None, length=None, offset=0):
        """Int8 / Int16 / UInt8 / UInt16 / UInt32 (widened to Int64 on the device) and UInt64 (read in place, COUNT /
        DISTINCT only): `values` holds one element per row"""
        n = (len(values) - offset) if length is None else length
        return Column(type_id, n, values=values, validity=validity, offset=offset)

    @staticmethod
    def boolean(bits, length, validity=None, offset=0):
        """Boolean: `bits` is the bit-packed values buffer (uint8, bit `offset + i` = row i); COUNT / DISTINCT only"""
        return Column(BOOL, length, values=bits, validity=validity, offset=offset)

    @staticmethod
    def utf8(offsets, data, validity=None, length=None, offset=0):
        n = (len(offsets) - 1 - offset) if length is None else length
        return Column(UTF8, n, offsets=offsets, data=data, validity=validity, offset=offset)

    @staticmethod
    def large_utf8(offsets, data, validity=None, length=None, offset=0):
        n = (len(offsets) - 1 - offset) if length is None else length
        return Column(LARGE_UTF8, n, offsets=offsets, data=data, validity=validity, offset=offset)

    @staticmethod
    def utf8_view(views, buffers, validity=None, length=None, offset=0):
        """Utf8View: `views` = 16 bytes per row (uint8 buffer of 16 * rows bytes), `buffers` = list of data buffers"""
        n = (len(views) // 16 - offset) if length is None else length
        return Column(UTF8_VIEW, n, values=views, validity=validity, offset=offset, variadic=list(buffers))

    @staticmethod
    def dict32_utf8(indices, dictionary, validity=None, length=None, offset=0):
        """Dictionary<Int32, Utf8>: int32 `indices` into `dictionary` (a utf8 / large_utf8 Column)."""
        n = (len(indices) - offset) if length is None else length
        return Column(DICT32_UTF8, n, values=indices, validity=validity, offset=offset, dictionary=dictionary)

    @staticmethod
    def from_arrow(arr):
        """pyarrow Array (Int64 / Float64 / Utf8 / LargeUtf8 / Dictionary<Int32, Utf8>, host memory) -> Column
        view of its buffers."""
        import numpy as np
        import pyarrow as pa

        bufs = arr.buffers()

        def view(b, dtype):
            return None if b is None else np.frombuffer(b, dtype=dtype)

        validity = view(bufs[0], np.uint8) if arr.null_count else None
        if pa.types.is_int64(arr.type):
            return Column(INT64, len(arr), values=view(bufs[1], np.int64), validity=validity, offset=arr.offset,
                          null_count=arr.null_count)
        if pa.types.is_float64(arr.type):
            return Column(FLOAT64, len(arr), values=view(bufs[1], np.float64), validity=validity,
                          offset=arr.offset, null_count=arr.null_count)
        t = arr.type
        if pa.types.is_int32(t) or pa.types.is_date32(t) or pa.types.is_time32(t):
            return Column(INT32, len(arr), values=view(bufs[1], np.int32), validity=validity, offset=arr.offset,
                          null_count=arr.null_count)
        if pa.types.is_float32(t):
            return Column(FLOAT32, len(arr), values=view(bufs[1], np.float32), validity=validity, offset=arr.offset,
                          null_count=arr.null_count)
        for is_t, type_id, dtype in ((pa.types.is_int8, INT8, np.int8), (pa.types.is_int16, INT16, np.int16),
                                     (pa.types.is_uint8, UINT8, np.uint8), (pa.types.is_uint16, UINT16, np.uint16),
                                     (pa.types.is_uint32, UINT32, np.uint32), (pa.types.is_uint64, UINT64, np.uint64)):
            if is_t(t):
                return Column(type_id, len(arr), values=view(bufs[1], dtype), validity=validity, offset=arr.offset,
                              null_count=arr.null_count)
        if pa.types.is_boolean(t):
            return Column(BOOL, len(arr), values=view(bufs[1], np.uint8), validity=validity, offset=arr.offset,
                          null_count=arr.null_count)
        if pa.types.is_timestamp(t) or pa.types.is_date64(t) or pa.types.is_time64(t) or pa.types.is_duration(t):
            # Int64-shaped: the checks see the stored integer (microseconds, milliseconds, ...)
            return Column(INT64, len(arr), values=view(bufs[1], np.int64), validity=validity, offset=arr.offset,
                          null_count=arr.null_count)
        if pa.types.is_string(arr.type):
            data = view(bufs[2], np.uint8) if bufs[2] is not None and bufs[2].size else np.zeros(1, np.uint8)
            return Column(UTF8, len(arr), offsets=view(bufs[1], np.int32), data=data, validity=validity,
                          offset=arr.offset, null_count=arr.null_count)
        if pa.types.is_large_string(arr.type):
            data = view(bufs[2], np.uint8) if bufs[2] is not None and bufs[2].size else np.zeros(1, np.uint8)
            return Column(LARGE_UTF8, len(arr), offsets=view(bufs[1], np.int64), data=data, validity=validity,
                          offset=arr.offset, null_count=arr.null_count)
        if hasattr(pa.types, "is_string_view") and pa.types.is_string_view(arr.type):
            return Column(UTF8_VIEW, len(arr), values=view(bufs[1], np.uint8), validity=validity, offset=arr.offset,
                          null_count=arr.null_count, variadic=[view(b, np.uint8) for b in bufs[2:]])
        if (pa.types.is_dictionary(arr.type) and pa.types.is_int32(arr.type.index_type)
                and (pa.types.is_string(arr.type.value_type) or pa.types.is_large_string(arr.type.value_type))):
            return Column(DICT32_UTF8, len(arr), values=view(bufs[1], np.int32), validity=validity,
                          offset=arr.offset, null_count=arr.null_count,
                          dictionary=Column.from_arrow(arr.dictionary))
        # Binary / LargeBinary / BinaryView have the string layouts: COUNT and COUNT(DISTINCT) compare bytes, which is what
        # the reference's SQL does with them (a pattern or LENGTH check on such a column is the caller's to refuse)
        if pa.types.is_binary(t) or pa.types.is_large_binary(t):
            data = view(bufs[2], np.uint8) if bufs[2] is not None and bufs[2].size else np.zeros(1, np.uint8)
            return Column(LARGE_UTF8 if pa.types.is_large_binary(t) else UTF8, len(arr),
                          offsets=view(bufs[1], np.int64 if pa.types.is_large_binary(t) else np.int32), data=data,
                          validity=validity, offset=arr.offset, null_count=arr.null_count)
        if hasattr(pa.types, "is_binary_view") and pa.types.is_binary_view(t):
            return Column(UTF8_VIEW, len(arr), values=view(bufs[1], np.uint8), validity=validity, offset=arr.offset,
                          null_count=arr.null_count, variadic=[view(b, np.uint8) for b in bufs[2:]])
        # fixed-width values of w bytes (FixedSizeBinary(w), Decimal128 = 16, Decimal256 = 32): equality of values is
        # equality of bytes (one precision / scale per column), so COUNT(DISTINCT) sees them as w-byte strings -- the
        # offsets 0, w, 2 w, ... are made here (4 bytes per row of host work; the values are not copied)
        width = t.byte_width if (pa.types.is_fixed_size_binary(t) or pa.types.is_decimal(t)) else 0
        if width:
            n = len(arr)
            offs = (np.arange(n + 1, dtype=np.int64) + arr.offset) * width
            if offs[-1] < 2**31:
                col = Column(UTF8, n, offsets=offs.astype(np.int32), data=view(bufs[1], np.uint8), validity=None, offset=0,
                             null_count=arr.null_count)
            else:
                col = Column(LARGE_UTF8, n, offsets=offs, data=view(bufs[1], np.uint8), validity=None, offset=0,
                             null_count=arr.null_count)
            if validity is not None:  # (the validity bitmap keeps the array's own offset: re-aligned to row 0)
                mask = np.unpackbits(validity, bitorder="little")[arr.offset:arr.offset + n]
                v = np.concatenate([np.packbits(mask, bitorder="little"), np.zeros(8, np.uint8)])
                col = Column(col.c.type, n, offsets=col._keep[2], data=col._keep[3], validity=v, offset=0,
                             null_count=arr.null_count)
            return col
        raise TgxError(2, "unsupported Arrow type %s" % arr.type)

    @staticmethod
    def validity_only(arr):
        """ANY pyarrow Array as a column for checks that read no values (completeness / COUNT: the validity bitmap and
        the length are all they need -- `SELECT COUNT(*), COUNT(c)` takes every column type, completeness.rs:158-163)"""
        import numpy as np

        bufs = arr.buffers()
        validity = np.frombuffer(bufs[0], dtype=np.uint8) if (arr.null_count and bufs and bufs[0] is not None) else None
        if validity is None and arr.null_count:  # NullArray: no bitmap, every row NULL
            return Column(INT64, len(arr), values=None, validity=np.zeros((len(arr) + 7) // 8 + 8, np.uint8), null_count=len(arr))
        return Column(INT64, len(arr), values=None, validity=validity, offset=arr.offset, null_count=arr.null_count)


def spec(kind, column, column2=-1, flags=0, pattern=None, kll_k=0, columns=None, length_min=0, length_max=None):
    """columns=[a, b, ...]: DISTINCT over the tuple of those columns (COUNT(DISTINCT (a, b)));
    length_min / length_max (None = unbounded): LENGTH bounds in characters"""
    pat = pattern.encode("utf-8") if isinstance(pattern, str) else pattern
    s = CheckSpec(kind, column, column2, flags, pat, len(pat) if pat else 0, kll_k, 0)
    s.length_min = length_min
    s.length_max = (1 << 64) - 1 if length_max is None else length_max
    if columns is not None and len(columns) >= 2:
        arr = (C.c_int32 * len(columns))(*columns)
        s.columns = C.cast(arr, C.POINTER(C.c_int32))
        s.n_columns = len(columns)
        s._keep_columns = arr
        s.column = columns[0]
    return s


# ---- the cross-rank step: transports and tgx_allreduce (include/tgx.h) ----------------------------------------
_ALLTOALL_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
_ALLTOALLV_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64),
                            C.c_size_t, C.c_void_p)
_ALLGATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class CommOps(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int32), ("world", C.c_int32), ("device_buffers", C.c_int32),
                ("reserved", C.c_int32), ("alltoall", _ALLTOALL_FN), ("alltoallv", _ALLTOALLV_FN),
                ("allgather", _ALLGATHER_FN)]


RCCL_UNIQUE_ID_BYTES = 128


class Comm:
    """tgx_comm: the transport tgx_allreduce runs over."""

    def __init__(self, handle, rank, world, keep=None):
        self.h, self.rank, self.world, self._keep = handle, rank, world, keep

    def __del__(self):
        if getattr(self, "h", None):
            lib().tgx_comm_destroy(self.h)
            self.h = None

    @staticmethod
    def rccl_unique_id():
        buf = (C.c_uint8 * RCCL_UNIQUE_ID_BYTES)()
        err = _Error()
        _check(lib().tgx_comm_rccl_unique_id(buf, C.byref(err)), err)
        return bytes(buf)

    @staticmethod
    def rccl(unique_id, rank, world):
        """RCCL over xGMI: the library calls ncclCommInitRank itself (collective: every rank must call)"""
        buf = (C.c_uint8 * RCCL_UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        h = C.c_void_p()
        err = _Error()
        _check(lib().tgx_comm_create_rccl(buf, rank, world, C.byref(h), C.byref(err)), err)
        return Comm(h, rank, world)

    @staticmethod
    def custom(rank, world, alltoall, alltoallv, allgather, device_buffers=False):
        """any transport: three Python callables over raw pointers (see tgx_comm_ops in include/tgx.h); they return
        nothing and raise on failure"""
        def guard(fn):
            def wrapped(*a):
                try:
                    fn(*a[1:])  # drop ctx
                    return 0
                except Exception:  # noqa: BLE001 -- must not unwind through the C frames
                    import traceback

                    traceback.print_exc()
                    return 1
            return wrapped

        ops = CommOps()
        ops.rank, ops.world, ops.device_buffers = rank, world, 1 if device_buffers else 0
        cbs = (_ALLTOALL_FN(guard(alltoall)), _ALLTOALLV_FN(guard(alltoallv)), _ALLGATHER_FN(guard(allgather)))
        ops.alltoall, ops.alltoallv, ops.allgather = cbs
        h = C.c_void_p()
        err = _Error()
        _check(lib().tgx_comm_create(C.byref(ops), C.byref(h), C.byref(err)), err)
        return Comm(h, rank, world, keep=cbs)  # the C side keeps the function pointers: keep the thunks alive


def blob_fingerprint_key(blob):
    """the fingerprint key a state blob was made under, or None when it holds no string / tuple keys"""
    out, keyed = (C.c_uint8 * 16)(), C.c_int32()
    rc = lib().tgx_blob_fingerprint_key(bytes(blob), len(blob), out, C.byref(keyed))
    if rc != 0:
        raise TgxError(rc, "not a tgx state blob of this version")
    return bytes(out) if keyed.value else None


class Plan:
    def __init__(self, specs, fingerprint_key=None):
        """`fingerprint_key`: 16 bytes -- the key of the plan's string / tuple fingerprints (tgx_plan_set_fingerprint_key);
        None: the one tgx_plan_create drew from the operating system"""
        self._specs = list(specs)
        arr = (CheckSpec * max(1, len(self._specs)))(*self._specs)
        h = C.c_void_p()
        err = _Error()
        _check(lib().tgx_plan_create(arr, len(self._specs), C.byref(h), C.byref(err)), err)
        self.h = h
        self.n = len(self._specs)
        if fingerprint_key is not None:
            self.set_fingerprint_key(fingerprint_key)

    def set_fingerprint_key(self, key):
        key = bytes(key)
        if len(key) != 16:
            raise ValueError("a fingerprint key is 16 bytes")
        err = _Error()
        _check(lib().tgx_plan_set_fingerprint_key(self.h, key, C.byref(err)), err)

    def fingerprint_key(self):
        out = (C.c_uint8 * 16)()
        lib().tgx_plan_get_fingerprint_key(self.h, out)
        return bytes(out)

    def __del__(self):
        if getattr(self, "h", None):
            lib().tgx_plan_destroy(self.h)
            self.h = None


class State:
    def __init__(self, plan, stream=None, _handle=None):
        self.plan = plan
        if _handle is not None:
            self.h = _handle
            return
        h = C.c_void_p()
        err = _Error()
        _check(lib().tgx_state_create(plan.h, stream, C.byref(h), C.byref(err)), err)
        self.h = h

    def close(self):
        """tgx_state_destroy now (not when the garbage collector gets to it)"""
        if getattr(self, "h", None):
            lib().tgx_state_destroy(self.h)
            self.h = None
        self._keep = None

    def __del__(self):
        self.close()

    def update(self, columns):
        cols = list(columns)
        arr = (_Column * max(1, len(cols)))(*[c.c if c is not None else _Column() for c in cols])
        err = _Error()
        _check(lib().tgx_update(self.plan.h, self.h, arr, len(cols), C.byref(err)), err)
        # DEVICE buffers must outlive the asynchronous kernels of EVERY batch queued since the last finalize / sync
        # (include/tgx.h): a streamed `update(b1); del b1; update(b2)` would otherwise hand b1's memory back to the
        # allocator while the scan of b1 is still reading it
        # HOST buffers are borrowed only until tgx_update returns: holding them would pin a whole streamed table in
        # host memory until finalize
        # (TGX_MEM_HOST_RETAINED buffers are promised to stay as they are until the next flushing call: held likewise)
        held = [c for c in cols if c is not None and (c.c.mem != 0 or (c.c.dictionary and c.c.dictionary.contents.mem != 0))]
        if held:
            if getattr(self, "_keep", None) is None:
                self._keep = []
            self._keep.append(held)

    def pending(self):
        """(batches, rows) that tgx_update has only noted so far: the last ones fed"""
        b, r = C.c_uint64(), C.c_uint64()
        lib().tgx_state_pending(self.h, C.byref(b), C.byref(r))
        return b.value, r.value

    def finalize(self):
        res = (Result * max(1, self.plan.n))()
        err = _Error()
        _check(lib().tgx_finalize(self.plan.h, self.h, res, self.plan.n, C.byref(err)), err)
        self._keep = None
        return list(res)[: self.plan.n]

    def sync(self):
        err = _Error()
        _check(lib().tgx_state_sync(self.h, C.byref(err)), err)
        self._keep = None

    def reset(self):
        err = _Error()
        _check(lib().tgx_state_reset(self.plan.h, self.h, C.byref(err)), err)  # waits for the stream
        self._keep = None

    def allreduce(self, comm):
        """tgx_allreduce: this rank's state becomes the state of the whole table (collective)"""
        err = _Error()
        _check(lib().tgx_allreduce(self.plan.h, self.h, comm.h, C.byref(err)), err)
        self._keep = None  # the step ends with the stream drained

    def merge(self, others):
        hs = (C.c_void_p * max(1, len(others)))(*[o.h for o in others])
        err = _Error()
        _check(lib().tgx_merge(self.plan.h, self.h, hs, len(others), C.byref(err)), err)

    def serialize(self):
        """packed partial state.  One library call when the blob fits the buffer kept from the last call (every call
        reads the accumulators back from the device), a second one with the reported size otherwise."""
        n = C.c_size_t()
        err = _Error()
        buf = getattr(self, "_ser_buf", None)
        if buf is None:
            buf = self._ser_buf = (C.c_uint8 * 8192)()
        rc = lib().tgx_state_serialize(self.plan.h, self.h, buf, len(buf), C.byref(n), C.byref(err))
        if rc != 0 and n.value > len(buf):  # "buffer too small": the needed size has been reported
            buf = self._ser_buf = (C.c_uint8 * (n.value + n.value // 4))()
            err = _Error()
            rc = lib().tgx_state_serialize(self.plan.h, self.h, buf, len(buf), C.byref(n), C.byref(err))
        _check(rc, err)
        return bytes(memoryview(buf)[: n.value])

    @staticmethod
    def deserialize(plan, blob):
        h = C.c_void_p()
        err = _Error()
        buf = (C.c_uint8 * max(1, len(blob))).from_buffer_copy(blob) if blob else (C.c_uint8 * 1)()
        _check(lib().tgx_state_deserialize(plan.h, buf, len(blob), C.byref(h), C.byref(err)), err)
        return State(plan, _handle=h)

    # KLL
    def kll_quantile(self, spec_index, phi):
        out = C.c_double()
        err = _Error()
        _check(lib().tgx_kll_quantile(self.plan.h, self.h, spec_index, phi, C.byref(out), C.byref(err)), err)
        return out.value

    def kll_summary(self, spec_index):
        n, lv, rt = C.c_uint64(), C.c_uint64(), C.c_uint64()
        mn, mx = C.c_double(), C.c_double()
        err = _Error()
        _check(lib().tgx_kll_summary(self.plan.h, self.h, spec_index, C.byref(n), C.byref(mn), C.byref(mx),
                                     C.byref(lv), C.byref(rt), C.byref(err)), err)
        return dict(n=n.value, min=mn.value, max=mx.value, num_levels=lv.value, num_retained=rt.value)

    def kll_level_items(self, spec_index, level):
        import numpy as np

        cnt = C.c_uint64()
        err = _Error()
        _check(lib().tgx_kll_level_items(self.plan.h, self.h, spec_index, level, None, 0, C.byref(cnt),
                                         C.byref(err)), err)
        out = np.zeros(max(1, cnt.value), dtype=np.float64)
        _check(lib().tgx_kll_level_items(self.plan.h, self.h, spec_index, level, out.ctypes.data, cnt.value,
                                         C.byref(cnt), C.byref(err)), err)
        return out[: cnt.value]

    # distinct key exchange
    def distinct_export(self, spec_index, world):
        ptr = C.c_void_p()
        counts = (C.c_uint64 * world)()
        err = _Error()
        _check(lib().tgx_distinct_export(self.plan.h, self.h, spec_index, world, C.byref(ptr), counts,
                                         C.byref(err)), err)
        return ptr.value, list(counts)

    def distinct_export_records(self, spec_index):
        """the key records of a DISTINCT task as a host array of uint64 rows: (key, count) for numeric keys,
        (fingerprint word a, word b, count, 0) for string / tuple keys (tests: the records are device memory)"""
        import numpy as np
        import torch

        ptr, counts = self.distinct_export(spec_index, 1)
        width = self.distinct_record_bytes(spec_index)
        total = counts[0]
        if total == 0:
            return np.zeros((0, width // 8), np.uint64)

        class P:
            __cuda_array_interface__ = {"shape": (total * width,), "typestr": "|u1", "data": (ptr, False), "version": 2}

        raw = torch.as_tensor(P(), device="cuda").clone().cpu().numpy()
        return raw.view(np.uint64).reshape(total, width // 8)

    def distinct_range_hint(self, spec_index, lo, hi):
        err = _Error()
        _check(lib().tgx_distinct_range_hint(self.plan.h, self.h, spec_index, int(lo), int(hi), C.byref(err)), err)

    def distinct_bitmap_view(self, spec_index):
        """(base, n_words, seen_ptr, twice_ptr or None); raises TGX_UNSUPPORTED when the set is a hash table"""
        base, n = C.c_int64(), C.c_uint64()
        seen, twice = C.c_void_p(), C.c_void_p()
        err = _Error()
        _check(lib().tgx_distinct_bitmap_view(self.plan.h, self.h, spec_index, C.byref(base), C.byref(n),
                                              C.byref(seen), C.byref(twice), C.byref(err)), err)
        return base.value, n.value, seen.value, twice.value

    def distinct_adopt_slices(self, spec_index, slice_base, seen_ptr, twice_ptr, n_slices, slice_words,
                              slice_stride_words=0):
        err = _Error()
        _check(lib().tgx_distinct_adopt_slices(self.plan.h, self.h, spec_index, int(slice_base), seen_ptr, twice_ptr,
                                               n_slices, slice_words, slice_stride_words, C.byref(err)), err)

    def distinct_record_bytes(self, spec_index):
        return lib().tgx_distinct_record_bytes(self.plan.h, self.h, spec_index)

    def distinct_import(self, spec_index, device_ptr, n_records):
        err = _Error()
        _check(lib().tgx_distinct_import(self.plan.h, self.h, spec_index, device_ptr, n_records, C.byref(err)),
               err)

    # profiling
    def profile_enable(self, on=True):
        lib().tgx_profile_enable(self.h, int(on))

    def profile_reset(self):
        lib().tgx_profile_reset(self.h)

    def profile_get(self, kernel):
        ms, n, b = C.c_double(), C.c_uint64(), C.c_uint64()
        err = _Error()
        _check(lib().tgx_profile_get(self.h, kernel.encode(), C.byref(ms), C.byref(n), C.byref(b), C.byref(err)),
               err)
        return dict(total_ms=ms.value, launches=n.value, bytes=b.value)
