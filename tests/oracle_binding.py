"""ctypes binding of oracle/libtgx_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import math
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "libtgx_oracle.so")
        if not os.path.exists(path):
            import subprocess

            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
        _LIB = C.CDLL(path)
        _declare(_LIB)
    return _LIB


class Count(C.Structure):
    _fields_ = [("total", C.c_int64), ("non_null", C.c_int64)]


class Stats(C.Structure):
    _fields_ = [
        ("total", C.c_int64), ("non_null", C.c_int64), ("has_value", C.c_int32), ("is_float", C.c_int32),
        ("min_i", C.c_int64), ("max_i", C.c_int64), ("min_f", C.c_double), ("max_f", C.c_double),
        ("sum_i_wrapping", C.c_int64), ("sum_f", C.c_double), ("sum_hi", C.c_double), ("mean", C.c_double),
        ("sumsq_f", C.c_double), ("has_variance", C.c_int32), ("var_samp", C.c_double),
        ("stddev_samp", C.c_double),
    ]


class Distinct(C.Structure):
    _fields_ = [("total", C.c_int64), ("non_null", C.c_int64), ("distinct", C.c_int64),
                ("groups_once", C.c_int64)]


class Comoments(C.Structure):
    _fields_ = [("n", C.c_uint64), ("sum_x", C.c_double), ("sum_y", C.c_double), ("sum_x2", C.c_double),
                ("sum_y2", C.c_double), ("sum_xy", C.c_double)]


class Corr(C.Structure):
    _fields_ = [("n", C.c_uint64), ("corr_has_value", C.c_int32), ("corr", C.c_double),
                ("covar_has_value", C.c_int32), ("covar_samp", C.c_double)]


class Match(C.Structure):
    _fields_ = [("total", C.c_int64), ("matches", C.c_int64)]


class SuiteColumn(C.Structure):
    _fields_ = [("values", C.c_void_p), ("validity", C.c_void_p), ("is_float", C.c_int32), ("reserved", C.c_int32)]


def _declare(L):
    vp, i64, u64, dbl = C.c_void_p, C.c_int64, C.c_uint64, C.c_double
    L.orc_suite_mt.argtypes = [C.POINTER(SuiteColumn), C.c_int32, C.POINTER(C.c_int32), C.c_int32, i64, C.c_int32,
                               C.POINTER(Count), C.POINTER(Stats), C.POINTER(Distinct)]
    L.orc_count.argtypes = [vp, i64, i64, C.POINTER(Count)]
    L.orc_stats_i64.argtypes = [vp, vp, i64, i64, C.POINTER(Stats)]
    L.orc_stats_f64.argtypes = [vp, vp, i64, i64, C.POINTER(Stats)]
    L.orc_distinct_bits64.argtypes = [vp, vp, i64, i64, C.POINTER(Distinct)]
    L.orc_distinct_utf8.argtypes = [vp, vp, vp, i64, i64, C.POINTER(Distinct)]
    L.orc_hll_registers.argtypes = [vp, vp, i64, i64, vp]
    L.orc_hll_registers.restype = None
    L.orc_hll_estimate.argtypes = [vp]
    L.orc_hll_estimate.restype = C.c_uint64
    L.orc_comoments.argtypes = [vp, C.c_int, vp, i64, vp, C.c_int, vp, i64, i64, C.POINTER(Comoments)]
    L.orc_pearson_from_state.argtypes = [C.POINTER(Comoments)]
    L.orc_pearson_from_state.restype = dbl
    L.orc_covariance_from_state.argtypes = [C.POINTER(Comoments)]
    L.orc_covariance_from_state.restype = dbl
    L.orc_corr_online.argtypes = [vp, C.c_int, vp, i64, vp, C.c_int, vp, i64, i64, C.POINTER(Corr)]
    L.orc_spearman_state.argtypes = [vp, C.c_int, vp, i64, vp, C.c_int, vp, i64, i64, C.POINTER(Comoments)]
    L.orc_kll_new.argtypes = [u64, C.c_int, u64]
    L.orc_kll_new.restype = vp
    L.orc_kll_free.argtypes = [vp]
    L.orc_kll_update.argtypes = [vp, dbl]
    L.orc_kll_update_many.argtypes = [vp, vp, vp, i64, i64]
    L.orc_kll_merge.argtypes = [vp, vp]
    L.orc_kll_quantile.argtypes = [vp, dbl, C.POINTER(dbl)]
    for name in ("count", "num_levels", "num_retained"):
        f = getattr(L, "orc_kll_" + name)
        f.argtypes, f.restype = [vp], u64
    for name in ("min", "max", "relative_error_bound"):
        f = getattr(L, "orc_kll_" + name)
        f.argtypes, f.restype = [vp], dbl
    L.orc_kll_level_items.argtypes = [vp, u64, vp, u64]
    L.orc_kll_level_items.restype = u64
    L.orc_kll_level_capacity.argtypes = [u64, u64]
    L.orc_kll_level_capacity.restype = u64
    L.orc_siphash.argtypes = [C.c_int, C.c_int, u64, u64, vp, C.c_size_t]
    L.orc_siphash.restype = u64
    if hasattr(L, "orc_regex_compile"):
        L.orc_regex_compile.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t]
        L.orc_regex_compile.restype = vp
        L.orc_regex_free.argtypes = [vp]
        L.orc_regex_is_match.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.orc_regex_count_utf8.argtypes = [vp, vp, vp, vp, i64, i64, C.c_int, C.c_int, C.POINTER(Match)]
        L.orc_length_count_utf8.argtypes = [vp, vp, vp, i64, i64, u64, u64, C.POINTER(Match)]


# ---------------------------------------------------------------- Arrow-layout helpers (numpy)
def pack_validity(mask, bit_offset=0):
    """mask: bool array (True = valid). Returns LSB-first bitmap with `bit_offset` leading pad bits."""
    mask = np.asarray(mask, dtype=bool)
    if bit_offset:
        mask = np.concatenate([np.zeros(bit_offset, dtype=bool), mask])
    return np.packbits(mask, bitorder="little")


def column_from_list(values, dtype):
    """[1, None, 3] -> (values ndarray, validity bitmap or None)"""
    mask = np.array([v is not None for v in values], dtype=bool)
    arr = np.array([0 if v is None else v for v in values], dtype=dtype)
    return arr, (None if mask.all() else pack_validity(mask))


def unpack_validity(validity, n):
    """LSB-first bitmap -> bool mask of n rows (None = all valid)"""
    if validity is None:
        return np.ones(n, dtype=bool)
    return np.unpackbits(np.asarray(validity, dtype=np.uint8), bitorder="little")[:n].astype(bool)


def utf8_from_list(values):
    """['a', None] -> (offsets int32, data uint8, validity or None)"""
    mask = np.array([v is not None for v in values], dtype=bool)
    enc = [b"" if v is None else (bytes(v) if isinstance(v, (bytes, bytearray)) else v.encode("utf-8")) for v in values]
    offsets = np.zeros(len(values) + 1, dtype=np.int32)
    if enc:
        offsets[1:] = np.cumsum([len(e) for e in enc])
    data = np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8).copy()
    return offsets, data, (None if mask.all() else pack_validity(mask))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def count(validity, n, offset=0):
    out = Count()
    lib().orc_count(_p(validity), offset, n, C.byref(out))
    return out


def stats(values, validity=None, n=None, offset=0):
    out = Stats()
    n = len(values) - offset if n is None else n
    if values.dtype == np.int64:
        lib().orc_stats_i64(_p(values), _p(validity), offset, n, C.byref(out))
    elif values.dtype == np.float64:
        lib().orc_stats_f64(_p(values), _p(validity), offset, n, C.byref(out))
    else:
        raise TypeError(values.dtype)
    return out


def distinct_bits64(values, validity=None, n=None, offset=0):
    out = Distinct()
    n = len(values) - offset if n is None else n
    assert values.dtype.itemsize == 8
    rc = lib().orc_distinct_bits64(_p(values), _p(validity), offset, n, C.byref(out))
    assert rc == 0
    return out


def hll_registers(values, validity=None, n=None, offset=0, registers=None):
    """HyperLogLog registers (16384 bytes) of the 8-byte values' bit patterns; `registers` continues a sketch"""
    import numpy as np

    n = len(values) - offset if n is None else n
    assert values.dtype.itemsize == 8
    regs = np.zeros(16384, np.uint8) if registers is None else registers
    lib().orc_hll_registers(_p(values), _p(validity), offset, n, _p(regs))
    return regs


def hll_estimate(registers):
    return int(lib().orc_hll_estimate(_p(registers)))


def distinct_utf8(offsets, data, validity=None, n=None, offset=0):
    out = Distinct()
    n = len(offsets) - 1 - offset if n is None else n
    rc = lib().orc_distinct_utf8(_p(offsets), _p(data), _p(validity), offset, n, C.byref(out))
    assert rc == 0
    return out


def suite_mt(columns, unique_cols, n, n_threads):
    """The null + range + unique suite on `n_threads` threads (oracle/suite_mt.c): row-range partitions, hash-set
    COUNT(DISTINCT), merge in partition order.  columns: [(values ndarray int64|float64, validity uint8 ndarray or
    None)]; returns (counts, stats, distincts)."""
    cols = (SuiteColumn * max(1, len(columns)))()
    for i, (v, b) in enumerate(columns):
        cols[i].values = v.ctypes.data
        cols[i].validity = None if b is None else b.ctypes.data
        cols[i].is_float = 1 if v.dtype == np.float64 else 0
    uc = (C.c_int32 * max(1, len(unique_cols)))(*unique_cols)
    counts = (Count * max(1, len(columns)))()
    stats_ = (Stats * max(1, len(columns)))()
    dist = (Distinct * max(1, len(unique_cols)))()
    rc = lib().orc_suite_mt(cols, len(columns), uc, len(unique_cols), n, n_threads, counts, stats_, dist)
    if rc != 0:
        raise MemoryError("orc_suite_mt failed (%d)" % rc)
    return list(counts)[: len(columns)], list(stats_)[: len(columns)], list(dist)[: len(unique_cols)]


def _isf(a):
    if a.dtype == np.float64:
        return 1
    if a.dtype == np.int64:
        return 0
    raise TypeError(a.dtype)


def comoments(x, y, xv=None, yv=None, n=None, xoff=0, yoff=0):
    out = Comoments()
    n = len(x) - xoff if n is None else n
    lib().orc_comoments(_p(x), _isf(x), _p(xv), xoff, _p(y), _isf(y), _p(yv), yoff, n, C.byref(out))
    return out


def corr_online(x, y, xv=None, yv=None, n=None, xoff=0, yoff=0):
    out = Corr()
    n = len(x) - xoff if n is None else n
    lib().orc_corr_online(_p(x), _isf(x), _p(xv), xoff, _p(y), _isf(y), _p(yv), yoff, n, C.byref(out))
    return out


def spearman_state(x, y, xv=None, yv=None, n=None, xoff=0, yoff=0):
    out = Comoments()
    n = len(x) - xoff if n is None else n
    rc = lib().orc_spearman_state(_p(x), _isf(x), _p(xv), xoff, _p(y), _isf(y), _p(yv), yoff, n,
                                  C.byref(out))
    assert rc == 0
    return out


def pearson(state):
    return lib().orc_pearson_from_state(C.byref(state))


def covariance(state):
    return lib().orc_covariance_from_state(C.byref(state))


class Kll:
    def __init__(self, k, parity_mode=0, seed=0):
        self._h = lib().orc_kll_new(k, parity_mode, seed)
        if not self._h:
            raise ValueError("k must be at least 2")
        self.k = k

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_kll_free(self._h)
            self._h = None

    def update(self, v):
        lib().orc_kll_update(self._h, float(v))

    def update_many(self, values, validity=None, n=None, offset=0):
        values = np.ascontiguousarray(values, dtype=np.float64)
        n = len(values) - offset if n is None else n
        lib().orc_kll_update_many(self._h, _p(values), _p(validity), offset, n)

    def merge(self, other):
        if lib().orc_kll_merge(self._h, other._h) != 0:
            raise ValueError("Cannot merge sketches with different k values")

    def quantile(self, phi):
        out = C.c_double()
        if lib().orc_kll_quantile(self._h, float(phi), C.byref(out)) != 0:
            raise ValueError("quantile error")
        return out.value

    count = property(lambda s: lib().orc_kll_count(s._h))
    num_levels = property(lambda s: lib().orc_kll_num_levels(s._h))
    num_retained = property(lambda s: lib().orc_kll_num_retained(s._h))
    min = property(lambda s: lib().orc_kll_min(s._h))
    max = property(lambda s: lib().orc_kll_max(s._h))
    error_bound = property(lambda s: lib().orc_kll_relative_error_bound(s._h))

    def level_items(self, level):
        n = lib().orc_kll_level_items(self._h, level, None, 0)
        buf = np.zeros(max(n, 1), dtype=np.float64)
        lib().orc_kll_level_items(self._h, level, _p(buf), n)
        return buf[:n]


class Regex:
    def __init__(self, pattern, case_insensitive=False):
        err = C.create_string_buffer(256)
        pb = pattern.encode("utf-8")
        self._h = lib().orc_regex_compile(pb, len(pb), int(case_insensitive), err, 256)
        if not self._h:
            raise ValueError(err.value.decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_regex_free(self._h)
            self._h = None

    def is_match(self, s):
        b = s.encode("utf-8") if isinstance(s, str) else bytes(s)
        return bool(lib().orc_regex_is_match(self._h, b, len(b)))

    def count_utf8(self, offsets, data, validity=None, n=None, offset=0, trim=False, null_is_valid=True):
        out = Match()
        n = len(offsets) - 1 - offset if n is None else n
        lib().orc_regex_count_utf8(self._h, _p(offsets), _p(data), _p(validity), offset, n, int(trim),
                                   int(null_is_valid), C.byref(out))
        return out


def length_count_utf8(offsets, data, validity=None, n=None, offset=0, min_chars=0, max_chars=None):
    out = Match()
    n = len(offsets) - 1 - offset if n is None else n
    lib().orc_length_count_utf8(_p(offsets), _p(data), _p(validity), offset, n, min_chars,
                                (1 << 64) - 1 if max_chars is None else max_chars, C.byref(out))
    return out


def nan_equal(a, b):
    return (math.isnan(a) and math.isnan(b)) or a == b
