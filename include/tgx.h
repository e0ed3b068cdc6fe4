/*
 * tgx.h -- C ABI of libtgx, the MI355X (gfx950) execution path for term-guard's Arrow-batch
 * check evaluator.
 *
 * The reference (term-guard 0.0.2, /root/reference/term-guard/src = "TG/") has no FFI of its own:
 * every check is a SQL string handed to DataFusion (`ctx.sql(..).collect()`), e.g.
 * TG/constraints/completeness.rs:158-167.  The drop-in boundary is therefore the reference's own
 * extension traits -- `Constraint::evaluate` (TG/core/constraint.rs:187-225) and
 * `Analyzer::{compute_state_from_data, merge_states, compute_metric_from_state}`
 * (TG/analyzers/traits.rs:65-148).  A `GpuConstraint: Constraint` on the Rust side (INTEGRATION.md
 * shows the binding) streams the table's RecordBatches, hands the raw Arrow buffers of each needed
 * column to `tgx_update`, and reads the aggregates back with `tgx_finalize`; each entry point
 * below names the reference interface it stands in for.
 *
 * Conventions
 *   - plain C, no C++ / torch / HIP types in any signature; HIP streams travel as `void*`.
 *   - every function returns a tgx_status and, when `err` is non-NULL, fills it on failure;
 *     nothing aborts or throws across the boundary (maps to TermError::Internal, TG/error.rs:89-90).
 *   - column buffers follow the Arrow C Data Interface layout (LSB-first validity bitmap, NULL
 *     when the array has no nulls; `offset` counts slots and applies to validity and values alike).
 *   - the caller owns column buffers; HOST buffers may be released when tgx_update returns, DEVICE
 *     buffers must stay alive AND UNMODIFIED until the next tgx_finalize / tgx_state_sync on that
 *     state (work is queued, small batches are only noted, a key set may walk a batch a second
 *     time, a SPEARMAN pair's first batch is ranked from the columns themselves: see tgx_update).
 *   - handles are not thread-safe; distinct handles are independent.  One HIP stream per state.
 *   - there is NO CPU fallback: without a usable gfx950 device tgx_init fails with TGX_NO_DEVICE
 *     and every compute entry point fails with it too.
 */
#ifndef TGX_H
#define TGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: tgx_column grew the Utf8View fields, tgx_check_spec the column list and LENGTH bounds,
 *    tgx_distinct_adopt_slices a slice stride */
/* 3: tgx_comm / tgx_allreduce (the cross-rank step behind the C ABI) */
/* 4: tgx_result grew the centred co-moments (co_*) at its end; state blobs are version 2 */
/* 5: tgx_type grew Int8 .. UInt64 / Boolean, tgx_memspace TGX_MEM_HOST_RETAINED; tgx_trim, tgx_cache_stats_get,
 *    tgx_state_pending (no struct changed its layout) */
/* 6: keyed fingerprints (tgx_plan_set_fingerprint_key / _get_, tgx_blob_fingerprint_key), TGX_FLAG_EXACT_KEYS; state
 *    blobs are version 3 (they carry the key) */
#define TGX_ABI_VERSION 6

typedef enum tgx_status {
  TGX_OK = 0,
  TGX_INVALID_ARGUMENT = 1,
  TGX_UNSUPPORTED = 2, /* type/shape/pattern outside the path: caller falls back to the stock SQL constraint */
  TGX_DEVICE_ERROR = 3,
  TGX_OUT_OF_MEMORY = 4,
  TGX_INTERNAL = 5,
  TGX_NO_DEVICE = 6
} tgx_status;

typedef struct tgx_error {
  int32_t code; /* tgx_status */
  char msg[256];
} tgx_error;

/* ---- column views (Arrow layout) ------------------------------------------------------- */
typedef enum tgx_type {
  TGX_INT64 = 1,
  TGX_FLOAT64 = 2,
  TGX_UTF8 = 3,        /* int32 offsets */
  TGX_LARGE_UTF8 = 4,  /* int64 offsets */
  /* Dictionary<Int32, Utf8|LargeUtf8>: int32 indices in `values` (validity/offset/length describe the
   * indices), the dictionary is a Utf8 column view of its own (any memory space). COUNT, DISTINCT and
   * REGEX_MATCH give the results of the decoded column: string work runs once per dictionary entry,
   * every batch may bring its own dictionary (unused and repeated entries allowed). A row whose dictionary
   * VALUE is NULL is a NULL row for every check (Arrow's logical nulls). */
  TGX_DICT32_UTF8 = 5,
  /* Utf8View (what DataFusion reads Parquet strings as): 16-byte views in `values` -- {int32 length, 12 inline
   * bytes} for length <= 12, else {int32 length, 4-byte prefix, int32 buffer index, int32 offset} -- and
   * `n_variadic` data buffers.  `variadic` is a HOST array of the buffers' pointers (the buffers themselves live
   * in `mem`); `variadic_sizes` (bytes, host array) is required for TGX_MEM_HOST columns, which are staged.
   * COUNT, DISTINCT and REGEX_MATCH give the results of the same values held as Utf8. */
  TGX_UTF8_VIEW = 6,
  /* 4-byte numerics: Int32 (also Date32, Time32) and Float32.  `values` holds 4 bytes per row.  Every check sees
   * exactly the Int64 / Float64 column the values stand for -- DataFusion's own coercion for SUM / AVG (Float64
   * accumulators), value-preserving for MIN / MAX / COUNT(DISTINCT): COUNT and NUMERIC_STATS read the 4-byte values
   * in place (widened in registers); for DISTINCT, KLL, COMOMENTS and SPEARMAN the batch's window is widened into a
   * staging buffer on the device first (one extra pass over those columns).  Int64-shaped Arrow types (Timestamp, Date64, Time64, Duration) are passed as
   * TGX_INT64 as they are. */
  TGX_INT32 = 7,
  TGX_FLOAT32 = 8,
  /* Narrow and unsigned integers (round 5): the reference's completeness / uniqueness SQL takes any column type
   * (TG/constraints/completeness.rs:158-163, uniqueness.rs:612-617) and Parquet tables are full of these.  `values`
   * holds 1 / 2 / 4 bytes per row; the batch's window is widened to Int64 on the device (value-preserving: every check
   * then sees the Int64 column the values stand for -- SUM in Int64 as DataFusion's for signed inputs; its UInt64 sum
   * of unsigned inputs has the same value while it fits).  Small HOST batches are coalesced like Int64 ones (widened
   * on their way into the pinned arena); DEVICE batches of these types are launched as they arrive. */
  TGX_INT8 = 9,
  TGX_INT16 = 10,
  TGX_UINT8 = 11,
  TGX_UINT16 = 12,
  TGX_UINT32 = 13,
  /* UInt64: 8 bytes per row read in place.  COUNT and DISTINCT only (a key is its bit pattern: COUNT(DISTINCT) is exact);
   * a plan that asks a statistic, a sketch or a correlation of such a column gets TGX_UNSUPPORTED from tgx_update --
   * the reference itself cannot read MIN / MAX / SUM of a UInt64 column (statistics.rs:277-308). */
  TGX_UINT64 = 14,
  /* Boolean: `values` is a bit-packed buffer like `validity` (bit `offset + i` = row i).  COUNT and DISTINCT only
   * (false and true are the keys 0 and 1); anything else is TGX_UNSUPPORTED (MIN / MAX of a Boolean column come back
   * Boolean in DataFusion: "Failed to extract statistic value" in the reference). */
  TGX_BOOL = 15
} tgx_type;

typedef enum tgx_memspace {
  TGX_MEM_HOST = 0,
  TGX_MEM_DEVICE = 1,
  /* HOST buffers the caller keeps alive AND unmodified until the next flushing call (tgx_finalize, tgx_state_sync,
   * tgx_state_serialize, tgx_merge, tgx_allreduce, tgx_state_reset) -- the contract DEVICE buffers have.  What it
   * buys: a small batch (coalesced, below) is only NOTED -- its windows are copied into the pinned arena by the
   * library's copy threads at some point between this call and the flush (every few MB of noted windows are handed to
   * the threads that are idle; the flush waits for them and copies the rest), instead of window by window on the
   * calling thread inside tgx_update (one core moves ~27 GB/s: the cap of a stream of 8192-row batches).  A consumer of
   * DataFusion's `execute_stream()` holds the RecordBatches (Arc'd buffers) until it syncs.  Batches that are not
   * coalesced are read before tgx_update returns, as TGX_MEM_HOST ones are.  All HOST columns of a batch should be
   * given the same way (a batch that mixes the two is taken as TGX_MEM_HOST). */
  TGX_MEM_HOST_RETAINED = 2
} tgx_memspace;

typedef struct tgx_column {
  int32_t type;            /* tgx_type */
  int32_t mem;             /* tgx_memspace: where every buffer below lives */
  int64_t length;          /* rows */
  int64_t offset;          /* Arrow `offset`: first logical slot */
  int64_t null_count;      /* -1 = unknown */
  const uint8_t *validity; /* may be NULL (no nulls) */
  const void *values;      /* fixed-width values / dictionary indices */
  const void *offsets;     /* Utf8: length+1 (+offset) int32/int64 offsets */
  const uint8_t *data;     /* Utf8: value bytes */
  const struct tgx_column *dictionary; /* TGX_DICT32_UTF8 only */
  const uint8_t *const *variadic;      /* TGX_UTF8_VIEW only: host array of n_variadic data buffer pointers */
  const int64_t *variadic_sizes;       /* TGX_UTF8_VIEW only: host array of their sizes in bytes (HOST columns) */
  int32_t n_variadic;
  int32_t reserved;
} tgx_column;

/* ---- check specs: the aggregates the reference's constraints emit as SQL ------------------ */
typedef enum tgx_check_kind {
  /* COUNT(*), COUNT(col)                       TG/constraints/completeness.rs:158-163,
   *                                            TG/analyzers/basic/{size,completeness}.rs */
  TGX_CHECK_COUNT = 1,
  /* MIN MAX SUM AVG [STDDEV VARIANCE]          TG/constraints/statistics.rs:45-74, :263;
   *                                            TG/analyzers/basic/{min_max,mean,sum}.rs */
  TGX_CHECK_NUMERIC_STATS = 2,
  /* COUNT(DISTINCT col) [+ GROUP BY col counts] TG/constraints/uniqueness.rs:612-617, 671-681, 709-715 */
  TGX_CHECK_DISTINCT = 3,
  /* COUNT(CASE WHEN [TRIM(]c[)] ~|~* 'pat' [OR c IS NULL] THEN 1 END), COUNT(*)
   *                                            TG/constraints/format.rs:750-776 */
  TGX_CHECK_REGEX_MATCH = 4,
  /* KllSketch::update over the column's non-NULL, non-NaN values
   *                                            TG/analyzers/advanced/kll_sketch.rs:195-229 */
  TGX_CHECK_KLL = 5,
  /* n, Sx, Sy, Sxx, Syy, Sxy over rows with both columns non-NULL (CAST AS DOUBLE)
   *                                            TG/analyzers/advanced/correlation.rs:239-249
   * and the centred moments behind CORR / COVAR_SAMP   TG/constraints/correlation.rs:260-275 */
  TGX_CHECK_COMOMENTS = 6,
  /* n and the sums of SQL RANK() ranks (min-rank ties) of both columns over rows with both non-NULL
   * (CAST AS DOUBLE), reported in sum_x .. sum_xy          TG/analyzers/advanced/correlation.rs:334-350 */
  TGX_CHECK_SPEARMAN = 7,
  /* COUNT(CASE WHEN LENGTH(c) >= length_min AND LENGTH(c) <= length_max OR c IS NULL THEN 1 END), COUNT(*):
   * LENGTH counts characters (code points), NULL rows always count    TG/constraints/length.rs:36-45, 167-171.
   * Result: total, matches. */
  TGX_CHECK_LENGTH = 8,
  /* APPROX_DISTINCT(col): a HyperLogLog estimate of COUNT(DISTINCT col)   TG/constraints/approx_count_distinct.rs:56-66.
   * One more lane of the numeric scan (2^14 one-byte registers like DataFusion's sketch; mergeable by max, so it
   * shards, merges and serializes like every other state): the column is read once at streaming speed instead of
   * going through the exact key set.  Result: `distinct` = the estimate (standard error 1.04 / sqrt(2^14) = 0.8 %; the
   * reference's own tests allow 3 %), total, non_null.  Int64 / Float64 / Int32 / Float32 columns (Float64 by bit
   * pattern, like DISTINCT); on string and dictionary columns -- and whenever the plan also holds an exact DISTINCT
   * check of the column -- the exact count answers (it satisfies every bound the estimate does). */
  TGX_CHECK_APPROX_DISTINCT = 9
} tgx_check_kind;

enum {
  TGX_FLAG_VARIANCE = 1u << 0,          /* NUMERIC_STATS: also sample variance / stddev */
  TGX_FLAG_MULTIPLICITY = 1u << 1,      /* DISTINCT: also #groups with cnt == 1 (NULL is a group) */
  TGX_FLAG_TRIM = 1u << 2,              /* REGEX: TRIM(col) (U+0020 only) before matching */
  TGX_FLAG_CASE_INSENSITIVE = 1u << 3,  /* REGEX: `~*` */
  TGX_FLAG_NULL_IS_VALID = 1u << 4,     /* REGEX: `OR col IS NULL` */
  /* SPEARMAN: exact sums instead of the reference's UInt64 arithmetic, whose rank products and sums wrap
   * modulo 2^64 (from about 3.8 M rows on) */
  TGX_FLAG_EXACT_RANK_SUMS = 1u << 5,
  /* DISTINCT over Utf8 / LargeUtf8 / Utf8View / Dictionary / tuple keys: an EXACT key set.  Without the flag such
   * keys are counted by their 128-bit KEYED fingerprints (tgx_plan_set_fingerprint_key): two distinct values count as
   * one only if all 128 bits agree under a key the data's producer does not know.  With it the state keeps the bytes of
   * every distinct key it has been fed (a device-side key store) and an equal fingerprint is confirmed by comparing
   * bytes -- `COUNT(DISTINCT c)` as DataFusion computes it (hash + equality, TG/constraints/uniqueness.rs:612-617,
   * 671-681, 709-715), for every input including one built against the fingerprint function.  What crosses a state
   * boundary (tgx_merge, state blobs, tgx_distinct_export / tgx_allreduce) travels as keyed fingerprints either way:
   * bytes never leave the device that was fed them.  One more place where keys go on as fingerprints: a first DEVICE
   * batch of 2 Mi rows or more is deduplicated in lists that refer to its rows (equal fingerprints are settled on the
   * rows' bytes); tgx_finalize hands the batch back to the caller, so a state that is FED AGAIN after tgx_finalize holds
   * that batch's keys by fingerprint from then on -- tgx_state_sync (instead of, or before, tgx_finalize) moves them
   * into the table with their bytes.  Numeric keys are exact regardless of the flag. */
  TGX_FLAG_EXACT_KEYS = 1u << 6
};

typedef struct tgx_check_spec {
  int32_t kind;         /* tgx_check_kind */
  int32_t column;       /* index into the columns array handed to tgx_update */
  int32_t column2;      /* COMOMENTS / SPEARMAN: second column; otherwise -1 */
  uint32_t flags;
  const char *pattern;  /* REGEX: pattern bytes (Rust `regex` syntax), not NUL-terminated */
  uint64_t pattern_len;
  uint32_t kll_k;       /* KLL: k >= 2 */
  uint32_t reserved;
  /* DISTINCT over a tuple of 2..8 columns -- COUNT(DISTINCT (a, b)), GROUP BY a, b
   * (TG/constraints/uniqueness.rs:557-562, 687-699, 709-715): `columns` lists them (then `column` is ignored).
   * A tuple is a value of its own: NULL components take part like any other value (SQL struct / GROUP BY
   * semantics), Float64 components compare by bit pattern, a string component is the row's string whatever the
   * layout (Utf8 / LargeUtf8 / Utf8View / Dictionary<Int32, Utf8>: a NULL dictionary entry is a NULL component).
   * Result: total = rows, non_null = rows whose every
   * component is non-NULL (the primary-key NULL check), distinct = tuples, groups_once with MULTIPLICITY.
   * n_columns == 0 or 1: the single-column check on `column`. */
  const int32_t *columns;
  uint32_t n_columns;
  uint32_t reserved2;
  uint64_t length_min;  /* LENGTH: inclusive bounds in characters; length_max = UINT64_MAX: no upper bound */
  uint64_t length_max;
} tgx_check_spec;

/* One result per spec.  Fields outside the spec's kind are zero. */
typedef struct tgx_result {
  int32_t kind;
  int32_t is_float;     /* NUMERIC_STATS: column type */
  int64_t total;        /* COUNT(*)  (rows seen) */
  int64_t non_null;     /* COUNT(col); COMOMENTS: rows with both sides non-NULL */
  /* NUMERIC_STATS */
  int32_t has_value;    /* 0 => MIN/MAX/SUM/AVG are SQL NULL (no non-NULL rows) */
  int32_t has_variance; /* 0 => STDDEV/VARIANCE are SQL NULL (fewer than 2 rows) */
  int64_t min_i, max_i; /* Int64 columns */
  double min_f, max_f;  /* Float64 columns: IEEE totalOrder; Int64 columns: the same as double */
  int64_t sum_i;        /* SUM(Int64), wrapping */
  double sum_f;         /* SUM(Float64) / SUM(CAST(Int64 AS DOUBLE)) */
  double mean;          /* AVG */
  double var_samp, stddev_samp;
  /* DISTINCT */
  int64_t distinct;     /* COUNT(DISTINCT col) */
  int64_t groups_once;  /* SUM(CASE WHEN cnt = 1 ...) over GROUP BY col; only with TGX_FLAG_MULTIPLICITY, else 0 */
  /* REGEX_MATCH */
  int64_t matches;
  /* COMOMENTS */
  double sum_x, sum_y, sum_x2, sum_y2, sum_xy;
  /* KLL: read through tgx_kll_* below */
  uint64_t kll_n;
  /* COMOMENTS, centred: means, M2x = SUM((x - mean_x)^2), M2y, Cxy = SUM((x - mean_x)(y - mean_y)) over the rows
   * with both sides non-NULL.  CORR(x, y) = Cxy / sqrt(M2x M2y) (0 when either M2 is 0), COVAR_SAMP = Cxy / (n - 1),
   * COVAR_POP = Cxy / n -- what DataFusion's online accumulators arrive at (TG/constraints/correlation.rs:260-275).
   * The library sums about a pivot near the data, so these hold their precision on offset columns (timestamps,
   * ids around 1e9) where n * sum_xy - sum_x * sum_y has cancelled; sum_x .. sum_xy above are the raw sums of
   * TG/analyzers/advanced/correlation.rs:239-249 and carry that analyzer's own cancellation. */
  double co_mean_x, co_mean_y, co_m2_x, co_m2_y, co_c_xy;
} tgx_result;

enum {
  /* every batch is launched as it arrives: no library-side coalescing of small batches (see tgx_update) */
  TGX_OPT_NO_COALESCE = 1u << 0
};

typedef struct tgx_options {
  int32_t device_id;       /* -1 = current HIP device */
  uint32_t flags;          /* TGX_OPT_* */
  uint64_t distinct_capacity_hint; /* expected rows per DISTINCT column (0 = grow on demand) */
} tgx_options;

typedef struct tgx_plan tgx_plan;
typedef struct tgx_state tgx_state;

/* ---- lifecycle ------------------------------------------------------------------------------ */
uint32_t tgx_abi_version(void);
/* Selects the device and checks it is gfx950.  No reference counterpart (the reference has no
 * device); called once per process before any other compute entry point. */
tgx_status tgx_init(const tgx_options *opts, tgx_error *err);
tgx_status tgx_shutdown(void);
const char *tgx_status_name(int32_t status);

/* Device memory between runs.  `ValidationSuite::run` (TG/core/suite.rs:399) is called once per table: a state is
 * created, fed, read and dropped.  The device (and pinned host) blocks of a destroyed state are kept by the library, by
 * size class, and handed to the next state -- from the second state of a process on, tgx_state_create .. tgx_finalize
 * performs no hipMalloc (a GB-sized hipMalloc costs milliseconds, hipFree waits for the whole device).  Bounded by
 * TGX_DEVICE_CACHE_MAX_BYTES (default: a quarter of the device's memory, at most 64 GiB); TGX_DEVICE_CACHE=0 turns it
 * off.  tgx_trim() returns everything cached to the driver (tgx_shutdown does too); no reference counterpart. */
typedef struct tgx_cache_stats {
  uint64_t device_cached_bytes;  /* held by the cache right now (not by live states) */
  uint64_t device_cached_blocks;
  uint64_t device_hits;          /* allocations served from the cache since the process started */
  uint64_t device_misses;        /* allocations that went to hipMalloc */
  uint64_t pinned_cached_bytes;
  uint64_t pinned_hits;
  uint64_t pinned_misses;
} tgx_cache_stats;
tgx_status tgx_trim(void);
tgx_status tgx_cache_stats_get(tgx_cache_stats *out);

/* Plan = the fused set of aggregates a ValidationSuite needs, grouped so each column buffer is
 * read once.  Replaces the per-constraint `format!("SELECT ...")` + `ctx.sql()` planning in
 * TG/core/suite.rs:67-100 (one scan per constraint). Regex patterns are validated
 * (TG/security.rs:152-183) and compiled here. */
tgx_status tgx_plan_create(const tgx_check_spec *specs, size_t n_specs, tgx_plan **out,
                           tgx_error *err);
void tgx_plan_destroy(tgx_plan *plan);
size_t tgx_plan_num_specs(const tgx_plan *plan);
/* The 128-bit key of the plan's string / tuple fingerprints (kernels/distinct128.hip: Chaskey-8).  tgx_plan_create
 * draws one from the operating system (getrandom); TGX_FINGERPRINT_KEY=<32 hex digits> fixes it for a process.
 * States, blobs and ranks can only be united when their fingerprints were made under ONE key: a plan that is to read
 * another process's blobs (tgx_state_deserialize says which key a blob carries in its error message and through
 * tgx_blob_fingerprint_key) or to take part in tgx_allreduce (every rank: the ranks compare keys in the facts round,
 * a mismatch is TGX_INVALID_ARGUMENT) is given the shared key here, BEFORE its first tgx_state_create -- afterwards
 * the call is refused.  No reference counterpart (DataFusion's hash sets keep the values themselves). */
tgx_status tgx_plan_set_fingerprint_key(tgx_plan *plan, const uint8_t key[16], tgx_error *err);
tgx_status tgx_plan_get_fingerprint_key(const tgx_plan *plan, uint8_t key_out[16]);

/* State = `Analyzer::State` for every spec of the plan (TG/analyzers/traits.rs:154-179).
 * `hip_stream` is a hipStream_t (NULL = a stream the library creates).  Everything the state does on the device is
 * queued on that stream and nowhere else: a DEVICE buffer handed to tgx_update has to be COMPLETE as far as that
 * stream is concerned -- written by work on the same stream, or by work the caller has waited for (an event the stream
 * waits on, or a synchronisation); the library's own stream does not wait for the legacy default stream. */
tgx_status tgx_state_create(const tgx_plan *plan, void *hip_stream, tgx_state **out, tgx_error *err);
void tgx_state_destroy(tgx_state *state);
/* How many of the batches handed to tgx_update so far are only NOTED (coalesced, not yet run): the LAST `*batches`
 * NON-EMPTY ones -- a batch of zero rows is never noted (tgx_update returns at once for it) and is not counted, so a
 * caller that keeps a window of retained batches must leave empty batches out of that window.
 * A caller that feeds TGX_MEM_HOST_RETAINED (or DEVICE) buffers of a long stream may release every batch before those
 * -- the library flushes by itself every few tens of MB -- instead of holding the whole table until tgx_finalize.
 * (DEVICE batches a sampled-range key set retains for a repair are the exception stated at tgx_update.)  Makes no
 * device call. */
tgx_status tgx_state_pending(const tgx_state *state, uint64_t *batches, uint64_t *rows);

/* One call per RecordBatch: replaces DataFusion's accumulator `update_batch` for the plan's
 * aggregates (`Analyzer::compute_state_from_data`, TG/analyzers/traits.rs:98-111).  `columns[i]` is
 * column i of the batch; unused columns may be zeroed.
 *
 * SMALL BATCHES ARE COALESCED.  DataFusion streams 8192-row RecordBatches (TG/core/context.rs:28-38): a batch of up to
 * 2^16 rows whose used columns are Int64 / Float64 / Int32 / Float32 (HOST or DEVICE) or HOST Utf8 / LargeUtf8 is only
 * NOTED by this call -- HOST windows are copied into a pinned arena first, so HOST buffers may still be released when
 * the call returns; DEVICE buffers must stay alive AND UNMODIFIED until the next tgx_finalize / tgx_state_sync (a noted
 * DEVICE window is read by a later flush, not by this call: a producer that recycles its device buffers in stream order
 * behind tgx_update -- fine for the kernels this call queues -- would have the flush read the next batch's bytes; such a
 * producer sets TGX_OPT_NO_COALESCE, or calls tgx_state_sync before it overwrites a buffer).  The pending
 * batches of every column are gathered into ONE contiguous device column and run as one batch when 4 Mi rows (or
 * 4096 batches, or a full arena) are pending, when a batch arrives that is not coalesced, and by every call that looks
 * at the state (tgx_finalize, tgx_state_sync, tgx_merge, tgx_state_serialize, tgx_allreduce, tgx_kll_*,
 * tgx_distinct_*).  Results do not depend on the batching: integers, key sets and pattern hits are those of the table;
 * float sums move in their last digits with the association, as they do between any two batchings.  Two arenas and two
 * sets of device regions take turns, so the host copies batch k+1 while the device works on flush k and nothing is
 * synchronised per batch: 8192-row batches of 8 columns run at 21 G rows/s from DEVICE buffers (0.4 us per call) and at
 * 0.70 G rows/s from HOST buffers (the copy into the arena, shared with three helper threads).  TGX_OPT_NO_COALESCE
 * turns it off.
 *
 * A batch that is NOT coalesced (more than 2^16 rows, Utf8View / dictionary columns, DEVICE strings) has its kernels
 * queued on the state's stream at once, and the call returns without waiting for them, except:
 *   - the FIRST batch (of 2^16 rows or more) an Int64 DISTINCT task sees waits for a sample of at most 2^16 of its
 *     values (that is: for whatever the stream still holds, plus ~20 us) to lay out the key set; later batches of
 *     the task never wait (keys outside the sampled range are counted and repaired at tgx_finalize /
 *     tgx_state_sync / tgx_state_serialize / tgx_merge / tgx_allreduce);
 *   - a hash key set that may have to grow reads its fill back first;
 *   (a Utf8 / LargeUtf8 / Utf8View DISTINCT task never waits: its first batch of 2^21 rows or more -- like that of
 *    a numeric task whose keys have no dense range -- is deduplicated in partitioned lists instead of the table, and
 *    should a list overflow -- heavily repeated values -- the batch is read once more at the next of the calls
 *    above: one more reason DEVICE buffers stay alive until then)
 *   - HOST batches: buffers copied straight from the caller's memory are borrowed only until the call returns, so
 *     it waits for the copies (small buffers travel through a pinned arena and do not). */
tgx_status tgx_update(const tgx_plan *plan, tgx_state *state, const tgx_column *columns,
                      size_t n_columns, tgx_error *err);

/* `AnalyzerState::merge` (TG/analyzers/traits.rs:160-170): folds srcs into dst.  Exact for every
 * kind, including DISTINCT (set union), unlike the reference's DistinctnessState::merge upper
 * bound (TG/analyzers/basic/distinctness.rs:77-92). */
tgx_status tgx_merge(const tgx_plan *plan, tgx_state *dst, tgx_state *const *srcs, size_t n_srcs,
                     tgx_error *err);

/* `Analyzer::compute_metric_from_state` inputs (TG/analyzers/traits.rs:113-122): waits for the
 * stream and writes one tgx_result per spec. Ratios, thresholds, assertions and messages stay
 * on the caller's side (TG/constraints/ *.rs). The state stays usable. */
tgx_status tgx_finalize(const tgx_plan *plan, tgx_state *state, tgx_result *results,
                        size_t n_results, tgx_error *err);
tgx_status tgx_state_sync(tgx_state *state, tgx_error *err);
tgx_status tgx_state_reset(const tgx_plan *plan, tgx_state *state, tgx_error *err);

/* Wire form of a state (the counterpart of the serde_json states the reference's
 * IncrementalAnalysisRunner stores, TG/analyzers/incremental/runner.rs:71-111); used to ship
 * partial states between ranks. `*len` receives the size needed/written. */
tgx_status tgx_state_serialize(const tgx_plan *plan, tgx_state *state, uint8_t *buf, size_t cap,
                               size_t *len, tgx_error *err);
tgx_status tgx_state_deserialize(const tgx_plan *plan, const uint8_t *buf, size_t len,
                                 tgx_state **out, tgx_error *err);
/* The fingerprint key a blob was made under (tgx_plan_set_fingerprint_key): `*keyed` = 0 when the blob holds no string /
 * tuple keys (then any plan of the same shape reads it), else 1 and `key_out` is the key.  tgx_state_deserialize refuses
 * a keyed blob under a plan with another key (TGX_INVALID_ARGUMENT). */
tgx_status tgx_blob_fingerprint_key(const uint8_t *buf, size_t len, uint8_t key_out[16], int32_t *keyed);

/* ---- KllSketch accessors (TG/analyzers/advanced/kll_sketch.rs:246-322, 368-399) ------------- */
tgx_status tgx_kll_quantile(const tgx_plan *plan, tgx_state *state, size_t spec_index, double phi,
                            double *out, tgx_error *err);
tgx_status tgx_kll_summary(const tgx_plan *plan, tgx_state *state, size_t spec_index, uint64_t *n,
                           double *min_value, double *max_value, uint64_t *num_levels,
                           uint64_t *num_retained, tgx_error *err);
/* copies level `level`'s items (weight 2^level each); *count receives the item count */
tgx_status tgx_kll_level_items(const tgx_plan *plan, tgx_state *state, size_t spec_index,
                               uint64_t level, double *out, uint64_t cap, uint64_t *count,
                               tgx_error *err);
double tgx_kll_relative_error_bound(uint32_t k);

/* ---- exact DISTINCT across ranks: hash-owner key exchange -------------
 * export: partitions this state's key set by owner = mix(key) % world into `world` contiguous
 *   runs of fixed-size records (tgx_distinct_record_bytes) in device memory the state owns (valid until the next call on the
 *   state); counts[r] = records for rank r.
 * import: replaces the state's key set with the union of the given records (device memory),
 *   marking the state "owner-partitioned" so that tgx_merge adds its counts instead of uniting. */
/* bytes per exchange record: 16 ({key, count}) for Int64/Float64 columns, 32 ({hash_a, hash_b, count, 0})
 * for Utf8 columns, whose values travel as 128-bit fingerprints */
size_t tgx_distinct_record_bytes(const tgx_plan *plan, const tgx_state *state, size_t spec_index);
tgx_status tgx_distinct_export(const tgx_plan *plan, tgx_state *state, size_t spec_index,
                               uint32_t world, const void **device_records, uint64_t *counts,
                               tgx_error *err);
tgx_status tgx_distinct_import(const tgx_plan *plan, tgx_state *state, size_t spec_index,
                               const void *device_records, uint64_t n_records, tgx_error *err);

/* Range-bitmap shortcut of the exchange, for Int64 columns whose global value range is dense:
 *   1. all ranks agree on the column's global [lo, hi] (all-reduce of their MIN/MAX) and call
 *      tgx_distinct_range_hint before the first batch: every rank then builds a bitmap with base = lo, so the
 *      bitmaps are congruent; keys outside [lo, hi] are counted and make tgx_finalize fail (never a wrong count);
 *   2. tgx_distinct_bitmap_view exposes the bitmap (device pointers, 32-bit words); ranks all-to-all equal
 *      slices of it (a few hundred MB per rank instead of 16 bytes per key);
 *   3. tgx_distinct_adopt_slices ORs the received slices (`n_slices` slices of `slice_words` words, slice i at
 *      word offset i * `slice_stride_words` -- 0 means packed, i.e. slice_words -- so the receive buffer of an
 *      all-to-all that carries several columns per peer is used in place; the "seen twice" slices too when
 *      the check wants multiplicity) into the rank's owned slice, whose first bit stands for key `slice_base`,
 *      and marks the state owner-partitioned.
 * tgx_distinct_bitmap_view returns TGX_UNSUPPORTED when the set is a hash table: use export / import then. */
tgx_status tgx_distinct_range_hint(const tgx_plan *plan, tgx_state *state, size_t spec_index, int64_t lo, int64_t hi,
                                   tgx_error *err);
tgx_status tgx_distinct_bitmap_view(const tgx_plan *plan, tgx_state *state, size_t spec_index, int64_t *base,
                                    uint64_t *n_words, const void **seen, const void **twice, tgx_error *err);
tgx_status tgx_distinct_adopt_slices(const tgx_plan *plan, tgx_state *state, size_t spec_index, int64_t slice_base,
                                     const void *seen_slices, const void *twice_slices, uint32_t n_slices,
                                     uint64_t slice_words, uint64_t slice_stride_words, tgx_error *err);

/* ---- the cross-rank step ------------------------------------------------
 * Rows shard by row range, one process per GPU.  Every rank runs tgx_update on its shard; tgx_allreduce then turns
 * each rank's state into the state of the WHOLE table -- `AnalyzerState::merge` (TG/analyzers/traits.rs:160-170)
 * applied across ranks:
 *   1. the ranks all-gather a few scalars per DISTINCT column: its MIN / MAX (the running values of the scan), the
 *      kind of key set the rank built, its rows;
 *   2. exact DISTINCT sets swap ONE all-to-all: where every rank holds a range bitmap, equal slices of the bitmaps,
 *      re-based on the agreed global range on the fly (the local bitmaps need not be congruent), all columns in one
 *      collective -- range / 8 bytes per rank and column; otherwise fixed-size key records by hash owner;
 *   3. the packed partial states (a few KiB; KLL about 100 KiB) travel in one all-gather and are folded in rank
 *      order, so every rank ends with bit-identical results.
 * After the call tgx_finalize(state) returns the global results on every rank; the state keeps its device buffers
 * for the next tgx_state_reset / tgx_update round.
 *   0. (before the above, for plans with SPEARMAN checks) RANK() over the union of the ranks' pairs: rank-based
 *      states do not merge (TG's neither, analyzers/advanced/correlation.rs:103-109), so the call runs a distributed
 *      sort per column -- local sort, world-1 splitters agreed from regular samples, every key to the rank that owns
 *      its value range (equal keys meet there), ranked, the rank back to its row: two all-to-all-v of 8 bytes per pair
 *      and column -- and adds up the five rank sums.  Every rank then answers with the sums of the whole table; such
 *      a state takes no further batches until it is reset, and still neither merges nor serializes.
 *
 * A tgx_comm is the transport: RCCL over xGMI (tgx_comm_create_rccl: the library dlopens librccl.so.1, calls
 * ncclCommInitRank itself and owns the communicator; tgx_comm_adopt_rccl wraps a caller-owned ncclComm_t), or any
 * transport of the caller's through tgx_comm_ops (MPI, gloo, the threaded stand-in of the tests).  All ranks must
 * call tgx_allreduce with the same plan, in the same order. */
typedef struct tgx_comm tgx_comm;
typedef struct tgx_comm_ops {
  void *ctx;
  int32_t rank, world;
  /* 1: the collectives below take DEVICE pointers and are ordered on `hip_stream`; 0: they take HOST pointers
   * (the library stages through pinned memory and synchronises the stream itself) and may block */
  int32_t device_buffers;
  int32_t reserved;
  /* every rank sends `bytes_per_peer` bytes to every rank: send / recv hold world blocks, block r goes to / comes
   * from rank r */
  int32_t (*alltoall)(void *ctx, const void *send, void *recv, size_t bytes_per_peer, void *hip_stream);
  /* block r of send holds send_counts[r] elements of elem_bytes for rank r (packed back to back); recv likewise */
  int32_t (*alltoallv)(void *ctx, const void *send, const uint64_t *send_counts, void *recv,
                       const uint64_t *recv_counts, size_t elem_bytes, void *hip_stream);
  /* every rank contributes `bytes` bytes; recv holds world blocks in rank order */
  int32_t (*allgather)(void *ctx, const void *send, void *recv, size_t bytes, void *hip_stream);
} tgx_comm_ops; /* every callback returns 0 on success */

#define TGX_RCCL_UNIQUE_ID_BYTES 128
tgx_status tgx_comm_create(const tgx_comm_ops *ops, tgx_comm **out, tgx_error *err);
/* rank 0 makes the id (ncclGetUniqueId) and hands it to the other ranks by any means */
tgx_status tgx_comm_rccl_unique_id(uint8_t id[TGX_RCCL_UNIQUE_ID_BYTES], tgx_error *err);
tgx_status tgx_comm_create_rccl(const uint8_t id[TGX_RCCL_UNIQUE_ID_BYTES], int32_t rank, int32_t world,
                                tgx_comm **out, tgx_error *err);
tgx_status tgx_comm_adopt_rccl(void *nccl_comm, int32_t rank, int32_t world, tgx_comm **out, tgx_error *err);
void tgx_comm_destroy(tgx_comm *comm);
tgx_status tgx_allreduce(const tgx_plan *plan, tgx_state *state, tgx_comm *comm, tgx_error *err);

/* ---- measurement ----------------------------------------------------------------------------
 * Per-kernel HIP-event timing on the state's stream (what bench.py's `roofline` uses).
 * Kernel names: "scan", "count", "distinct", "regex", "kll", "comoments"; "distinct_lists" is the share of
 * "distinct" spent on big Utf8 batches that were deduplicated through partitioned fingerprint lists. */
tgx_status tgx_profile_enable(tgx_state *state, int32_t on);
tgx_status tgx_profile_get(tgx_state *state, const char *kernel, double *total_ms,
                           uint64_t *launches, uint64_t *algorithmic_bytes, tgx_error *err);
tgx_status tgx_profile_reset(tgx_state *state);

/* ---- host-side rules the Rust shim shares with the reference -------------------------------- */
/* SqlSecurity::validate_regex_pattern (TG/security.rs:152-183) + "does the engine cover it". */
tgx_status tgx_regex_validate(const char *pattern, size_t len, uint32_t flags, tgx_error *err);
/* Host-side `is_match` of the compiled automaton for one value (debug / small inputs). */
tgx_status tgx_regex_is_match(const char *pattern, size_t plen, uint32_t flags, const uint8_t *value,
                              size_t vlen, int32_t *matched, tgx_error *err);

/* Host-side walk of the PRODUCT automaton of up to 4 patterns -- the form in which several pattern checks of one
 * column are evaluated on the device (one walk over the value decides all of them).  Bit k of *mask: pattern k
 * matches `value`.  *grouped = 0 when the product exceeds the device's table limit: the patterns then run one by
 * one (and *mask is computed that way).  TRIM is applied per pattern here; on the device only patterns with the same
 * TRIM flag share a walk. */
tgx_status tgx_regex_match_group(const char *const *patterns, const size_t *pattern_lens, const uint32_t *flags,
                                 size_t n_patterns, const uint8_t *value, size_t vlen, uint32_t *mask,
                                 int32_t *grouped, tgx_error *err);

#ifdef __cplusplus
}
#endif
#endif /* TGX_H */
