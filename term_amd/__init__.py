"""term_amd -- MI355X-native execution path for term-guard's Arrow-batch check evaluator.

The compute path is libtgx.so (hand-written HIP kernels for gfx950 behind the C ABI of
include/tgx.h).  This package is the thin ctypes binding plus the host-side mirror of the
reference's ValidationSuite / Check / Constraint surface.  There is no CPU fallback: importing works
anywhere (so the ABI can be inspected), but every compute call needs a gfx950 device.
"""
from ._lib import (  # noqa: F401
    TgxError,
    Column,
    Plan,
    State,
    CheckSpec,
    Result,
    init,
    lib,
    lib_path,
    abi_symbols,
    cache_stats,
    trim,
    blob_fingerprint_key,
    COUNT,
    NUMERIC_STATS,
    DISTINCT,
    REGEX_MATCH,
    KLL,
    COMOMENTS,
    SPEARMAN,
    LENGTH,
    APPROX_DISTINCT,
    OPT_NO_COALESCE,
    FLAG_EXACT_RANK_SUMS,
    FLAG_EXACT_KEYS,
    FLAG_VARIANCE,
    FLAG_MULTIPLICITY,
    FLAG_TRIM,
    FLAG_CASE_INSENSITIVE,
    FLAG_NULL_IS_VALID,
    INT8,
    INT16,
    UINT8,
    UINT16,
    UINT32,
    UINT64,
    BOOL,
    INT32,
    INT64,
    FLOAT32,
    FLOAT64,
    UTF8,
    LARGE_UTF8,
    DICT32_UTF8,
    UTF8_VIEW,
    MEM_HOST,
    MEM_DEVICE,
    MEM_HOST_RETAINED,
)

__version__ = "0.1.0"
