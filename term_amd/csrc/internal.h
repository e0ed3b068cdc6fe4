// internal.h -- host-side structures behind the opaque handles of include/tgx.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <deque>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tgx.h"
#include "kernels/device_types.h"
#include "kernels/distinct_types.h"
#include "kll_host.h"

namespace tgx {

// "nothing throws across the boundary" (include/tgx.h): every extern "C" entry point is a function-try-block whose
// handler lands here (std::bad_alloc from a host container, std::out_of_range, ...)
inline tgx_status abi_exception(tgx_error *err) {
  tgx_status code = TGX_INTERNAL;
  const char *what = "unknown C++ exception";
  try {
    throw;
  } catch (const std::bad_alloc &) {
    code = TGX_OUT_OF_MEMORY;
    what = "host allocation failed (std::bad_alloc)";
  } catch (const std::exception &e) {
    what = e.what();
  } catch (...) {
  }
  if (err) {
    err->code = (int32_t)code;
    snprintf(err->msg, sizeof(err->msg), "%s", what);
  }
  return code;
}

// ---- kernel launchers (defined in kernels/*.hip) ----------------------------------------------
void launch_scan_pivot(const ScanLaunch &L, int n_cols, double *d_pivots, int32_t *d_pivot_set, hipStream_t stream);
void launch_scan_main_only(const ScanLaunch &L, int n_cols, int blocks_per_col, ScanPartial *d_partials,
                           ScanAcc *d_accs, hipStream_t stream);
void launch_scan_kll(const ScanLaunch &L, int n_cols, int blocks_per_col, size_t lds_bytes, ScanPartial *d_partials,
                     hipStream_t stream);
void launch_scan_pairs(const ScanPairLaunch &L, int n_pairs, int blocks_per_pair, size_t lds_bytes,
                       ScanPartial *d_partials, void *d_como_partials, const ComomentAcc *d_como_accs,
                       hipStream_t stream);
void launch_scan_hll(const ScanLaunch &L, int n_cols, int blocks_per_col, ScanPartial *d_partials, hipStream_t stream);
void launch_como_pivot(const ComomentLaunch &L, int n_pairs, ComomentAcc *d_accs, hipStream_t stream);
void launch_comoments_reduce(const ComomentLaunch &L, int n_pairs, int blocks_per_pair, const void *d_partials,
                             ComomentAcc *d_accs, hipStream_t stream);
// mode: 0 Int32, 1 Float32, 2 Int8, 3 Int16, 4 UInt8, 5 UInt16, 6 UInt32, 7 Boolean (bits) -> Int64 / Float64
void launch_widen32(const void *src, void *dst, int64_t n, int mode, int n_cu, hipStream_t stream);
void launch_scan_reduce_only(const ScanLaunch &L, int n_cols, int blocks_per_col, ScanPartial *d_partials,
                             ScanAcc *d_accs, hipStream_t stream, const OutlierStats *outliers = nullptr);
void launch_count(const CountLaunch &L, int n_cols, int blocks_per_col, unsigned long long *d_block_counts,
                  CountAcc *d_accs, hipStream_t stream);
size_t comoments_partial_bytes();
void launch_comoments(const ComomentLaunch &L, int n_pairs, int blocks_per_pair, void *d_partials,
                      ComomentAcc *d_accs, hipStream_t stream);
void launch_distinct_hash(const DistinctColDesc &d, const HashSetView &t,
                          unsigned long long *d_counters, hipStream_t stream);
void launch_distinct_bitmap(const DistinctColDesc &d, const BitmapView &bm,
                            unsigned long long *d_counters, hipStream_t stream);
int partition_grid(int64_t length, int n_cu);  // workgroups of launch_partition (= ScanPartials it writes with stats)
void launch_distinct_init(DistinctSample *sample, OutlierStats *outliers, hipStream_t stream);
void launch_partition_init(const PartitionParams &p, unsigned long long *totals, hipStream_t stream);
void launch_distinct_sample(const DistinctColDesc &d, DistinctSample *out, hipStream_t stream);
void launch_distinct_outliers(const DistinctColDesc &d, int64_t base, uint64_t range, const HashSetView &t,
                              unsigned long long *d_counters, hipStream_t stream);
void launch_partition(const PartitionParams &p, unsigned long long *d_counters, int n_cu, hipStream_t stream);
hipError_t launch_bucket_apply(const PartitionParams &p, unsigned long long *d_counters, hipStream_t stream);
void launch_bitmap_adopt(const uint32_t *seen_slices, const uint32_t *twice_slices, uint32_t n_slices,
                         uint64_t slice_words, uint64_t stride_words, uint32_t *out_seen, uint32_t *out_twice,
                         unsigned long long *d_counters, hipStream_t stream);
void launch_bitmap_rebase(const uint32_t *src, uint64_t src_words, long long delta_bits, uint32_t world,
                          uint64_t slice_words, uint64_t row_words, uint64_t col_words, uint32_t *send,
                          hipStream_t stream);
void launch_distinct_utf8(const void *offsets, const uint8_t *data, const void *views,
                          const uint8_t *const *buffers, const uint8_t *validity, int64_t offset,
                          int64_t length, int large_offsets, int want_mult, const HashSetView &t, const FpKey &key,
                          unsigned long long *d_counters, hipStream_t stream);
void launch_exact_measure_utf8(const void *offsets, const uint8_t *data, const void *views,
                               const uint8_t *const *buffers, const uint8_t *validity, int64_t offset, int64_t length,
                               int large_offsets, const uint32_t *dict_seen, unsigned long long *out,
                               hipStream_t stream);
void launch_exact_measure_tuple(const TupleDesc &d, unsigned long long *out, hipStream_t stream);
void launch_gather_segments(const GatherSeg *d_segs, int n_segs, int parts, hipStream_t stream);
void launch_state_reset(const StateResetArgs &a, hipStream_t stream);
int tgx_num_cus();  // CUs of the device tgx_init bound (256 before init)
void launch_dict_count_hits(const int32_t *indices, const uint8_t *validity, int64_t offset, int64_t length,
                            int64_t dict_length, int dict_has_nulls, const uint8_t *hits, int null_is_valid,
                            unsigned long long *d_counters, int n_cu, hipStream_t stream);
void launch_dict_count(const int32_t *indices, const uint8_t *validity, int64_t offset, int64_t length,
                       const uint8_t *dict_validity, int64_t dict_offset, int64_t dict_length, CountAcc *acc, int n_cu,
                       hipStream_t stream);
int dict_fuse_capacity(int64_t length, int64_t dict_length, int want_mult, int n_cu);
void launch_dict_usage_fused(const int32_t *indices, const uint8_t *validity, int64_t offset, int64_t length,
                             int64_t dict_length, int want_mult, int n_patterns, const uint8_t *const *hits,
                             unsigned long long *const *pattern_counters, const int32_t *null_is_valid, uint32_t *seen,
                             uint32_t *twice, uint32_t *scratch, unsigned long long *d_counters, int n_cu,
                             hipStream_t stream);
size_t dict_usage_words(int64_t dict_length);
size_t dict_usage_scratch_bytes(int64_t length, int64_t dict_length, int want_mult, int n_cu);
void launch_dict_usage(const int32_t *indices, const uint8_t *validity, int64_t offset, int64_t length,
                       const uint8_t *dict_validity, int64_t dict_offset, int64_t dict_length, int want_mult,
                       uint32_t *seen, uint32_t *twice, uint32_t *scratch, unsigned long long *d_counters, int n_cu,
                       hipStream_t stream);
void launch_dict_insert(const void *offsets, const uint8_t *data, const uint8_t *validity, int64_t offset,
                        int64_t length, int large_offsets, int want_mult, const uint32_t *seen,
                        const uint32_t *twice, const HashSetView &t, const FpKey &key, unsigned long long *d_counters,
                        hipStream_t stream);
void launch_distinct_tuple(const TupleDesc &d, const HashSetView &t, unsigned long long *d_counters,
                           hipStream_t stream);
// big Utf8 batches: fingerprints partitioned into lists, deduplicated list by list in LDS (distinct128.hip, fp_*)
void launch_fp_partition_strings(const void *offsets, const uint8_t *data, const uint8_t *validity, int64_t offset,
                                 int64_t length, int large_offsets, const FpLists &level1, const FpKey &key,
                                 uint32_t *exact_fb_lo, unsigned long long *d_counters, hipStream_t stream);
void launch_fp_partition_views(const void *views, const uint8_t *const *buffers, const uint8_t *validity,
                               int64_t offset, int64_t length, const FpLists &level1, const FpKey &key,
                               uint32_t *exact_fb_lo, unsigned long long *d_counters, hipStream_t stream);
void launch_fp_partition_tuples(const TupleDesc &d, const FpLists &level1, uint32_t *exact_fb_lo,
                                unsigned long long *d_counters, hipStream_t stream);
// exact sets: the lists' records carry rows; equal fingerprints are settled on the rows' bytes
void launch_fp_count_exact_utf8(const FpLists &level2, int want_mult, uint2 *per_list, const uint32_t *offered1,
                                const void *offsets, const uint8_t *data, const void *views, const uint8_t *const *buffers,
                                int64_t offset, int64_t length, int large_offsets, unsigned long long *d_counters,
                                hipStream_t stream);
void launch_fp_count_exact_tuple(const FpLists &level2, int want_mult, uint2 *per_list, const TupleDesc &d,
                                 unsigned long long *d_counters, hipStream_t stream);
void launch_fp_demote(const FpLists &level2, const uint32_t *fb_lo, const HashSetView &t, int want_mult,
                      unsigned long long *d_counters, hipStream_t stream);
void launch_fp_partition_lists(const FpLists &level1, const FpLists &level2, unsigned long long *d_counters,
                               hipStream_t stream);
void launch_fp_count(const FpLists &level2, int want_mult, uint2 *per_list, const uint32_t *offered1,
                     unsigned long long *d_counters, hipStream_t stream);
void launch_fp_insert(const FpLists &level2, const HashSetView &t, int want_mult, hipStream_t stream);
// the same for big batches of keys without a dense range (distinct.hip, key_*): 8-byte records (the mixed key)
void launch_key_lists(const DistinctColDesc &d, const FpLists &level1, const FpLists &level2, int want_mult,
                      uint2 *per_list, unsigned long long *d_counters, hipStream_t stream);
void launch_key_insert(const FpLists &level2, const HashSetView &t, int want_mult, hipStream_t stream);
void launch_hash_rehash128(const HashSetView &src, const HashSetView &dst, int want_mult,
                           unsigned long long *d_counters, hipStream_t stream);
void launch_hash_import128(const KeyRecord128 *recs, uint64_t n, const HashSetView &dst, int want_mult,
                           unsigned long long *d_counters, hipStream_t stream);
void launch_hash_export_count128(const HashSetView &src, uint32_t world, unsigned long long *d_counts,
                                 hipStream_t stream);
void launch_hash_export_scatter128(const HashSetView &src, uint32_t world, int want_mult,
                                   unsigned long long *d_cursors, KeyRecord128 *out, hipStream_t stream);
void launch_hash_rehash(const HashSetView &src, const HashSetView &dst, int want_mult,
                        unsigned long long *d_counters, hipStream_t stream);
void launch_bitmap_to_hash(const BitmapView &bm, const HashSetView &dst, int want_mult,
                           unsigned long long *d_counters, hipStream_t stream);
void launch_hash_import(const KeyRecord *recs, uint64_t n, const HashSetView &dst, int want_mult,
                        unsigned long long *d_counters, hipStream_t stream);
void launch_hash_export_count(const HashSetView &src, uint32_t world, unsigned long long *d_counts,
                              hipStream_t stream);
void launch_hash_export_scatter(const HashSetView &src, uint32_t world, int want_mult,
                                unsigned long long *d_cursors, KeyRecord *out, hipStream_t stream);
void launch_bitmap_export_count(const BitmapView &bm, uint32_t world, unsigned long long *d_counts,
                                hipStream_t stream);
void launch_bitmap_export_scatter(const BitmapView &bm, uint32_t world, int want_mult,
                                  unsigned long long *d_cursors, KeyRecord *out, hipStream_t stream);

// ---- plan ---------------------------------------------------------------------------------------
struct ScanTask {
  int column;
  bool variance;
  bool stats_needed = false;  // a NUMERIC_STATS spec or a key set's range reads its MIN / MAX / SUM (else COUNT only)
};
struct CountTask {
  int column;
};
struct DistinctTask {
  int column;
  bool multiplicity;
  int scan_slot;  // scan task that provides MIN/MAX for the bitmap decision
  std::vector<int> tuple;  // >= 2 columns: COUNT(DISTINCT (a, b, ...)); `column` is then tuple[0]
  // only APPROX_DISTINCT specs point here: the task runs for the column kinds the HyperLogLog lane does not take
  // (strings, dictionaries) and stays idle on numeric columns
  bool approx_only = false;
  // TGX_FLAG_EXACT_KEYS on any DISTINCT spec of the task: string / tuple keys are kept with their bytes and equal
  // fingerprints are confirmed byte by byte (numeric keys are exact either way)
  bool exact = false;
};
// APPROX_DISTINCT: the HyperLogLog lane of the numeric scan (kernels/scan.hip, scan_hll_kernel).  When the plan also
// holds an exact DISTINCT check of the column -- or the column turns out to be a string column -- the exact key set
// answers instead (`distinct_slot`).
struct HllTask {
  int column;
  int scan_slot;
  int distinct_slot;
};
struct ComomentTask {
  int col_x, col_y;
};
struct KllTask {
  int column;
  uint32_t k;
};

enum class Source { kScan, kCount };

struct SpecBinding {
  int kind;
  int slot;         // task index within its kind
  Source count_src; // TGX_CHECK_COUNT: where total / non_null come from
};

}  // namespace tgx

struct tgx_plan {
  std::vector<tgx_check_spec> specs;
  std::vector<std::string> patterns;
  std::vector<tgx::SpecBinding> bind;
  std::vector<tgx::ScanTask> scan;
  std::vector<tgx::CountTask> count;
  std::vector<tgx::DistinctTask> distinct;
  std::vector<tgx::ComomentTask> como;
  std::vector<tgx::KllTask> kll;
  std::vector<tgx::HllTask> hll;
  int n_columns_needed = 0;  // 1 + max column index
  // per plan column, fixed at tgx_plan_create (tgx_update runs once per 8192-row batch: nothing is allocated there)
  std::vector<char> used, reads_values, needs_wide;
  std::vector<char> key_column;  // a single-column DISTINCT check reads it (range tracking of coalesced HOST batches)
  std::vector<char> stats_on;    // a statistic, sketch, correlation or ranking reads it (TGX_UINT64 / TGX_BOOL columns may not)
  // the key of the string / tuple fingerprints (kernels/distinct128.hip): drawn from the OS at tgx_plan_create, or set
  // by tgx_plan_set_fingerprint_key before the plan's first state exists
  tgx::FpKey fp_key;
  mutable std::atomic<bool> fp_key_locked{false};  // a state has been created: the key may not change any more
  void *regex = nullptr;     // tgx::RegexPlan (regex_device.cpp)
  void *spearman = nullptr;  // tgx::SpearmanPlan (spearman_device.cpp)
};

namespace tgx {

// devcache.cpp: the process-wide cache of device / pinned allocations
size_t cache_size_class(size_t bytes);
hipError_t dev_alloc(void **p, size_t *cap, size_t bytes);  // `*cap` = the block's size class (>= bytes)
void dev_free(void *p, size_t cap);                          // waits for the device unless inside a QuiescedScope
hipError_t pinned_alloc(void **p, size_t bytes);
void pinned_free(void *p, size_t bytes);                     // `bytes` as asked for
// a non-blocking stream from / back to the library's pool (the caller has waited for everything queued on it)
hipError_t stream_acquire(hipStream_t *out, bool high_priority);
void stream_release(hipStream_t s, bool high_priority);
void dev_cache_trim();
void dev_cache_set_live(bool on);  // tgx_init: true; tgx_shutdown (after its trim): false
void dev_cache_stats(tgx_cache_stats *out);
// "the device has been waited for and this thread queues nothing until the scope ends": releases inside it skip the
// wait that keeps a cached block away from work still in flight (tgx_state_destroy)
struct QuiescedScope {
  QuiescedScope();
  ~QuiescedScope();
  QuiescedScope(const QuiescedScope &) = delete;
  QuiescedScope &operator=(const QuiescedScope &) = delete;
};

// device buffer that grows on demand (its memory comes from and returns to the cache)
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  bool owned = true;  // false: a slice of somebody else's allocation (borrow())
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap), owned(o.owned) {
    o.p = nullptr;
    o.cap = 0;
    o.owned = true;
  }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) {
      release();
      p = o.p;
      cap = o.cap;
      owned = o.owned;
      o.p = nullptr;
      o.cap = 0;
      o.owned = true;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p && owned) dev_free(p, cap);
    p = nullptr;
    cap = 0;
    owned = true;
  }
  // a slice of a pool that outlives this object (the per-task counters of a state live in one allocation, so that
  // finalize reads all of them back with ONE copy)
  void borrow(void *ptr, size_t bytes) {
    release();
    p = ptr;
    cap = bytes;
    owned = false;
  }
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    release();
    return dev_alloc(&p, &cap, bytes);
  }
  // for buffers that are re-sized flush after flush (a growth frees the old block: a wait for the whole device): ask
  // for a quarter more than is needed, so that sizes that creep upwards settle after a few flushes
  hipError_t reserve_roomy(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    return reserve(bytes + bytes / 4);
  }
  template <class T>
  T *as() const {
    return (T *)p;
  }
};

enum class DistinctMode { kUndecided, kBitmap, kHash, kHostOnly };

struct DistinctState {
  DistinctMode mode = DistinctMode::kUndecided;
  int col_type = 0;
  // bitmap
  DevBuf seen, twice;
  int64_t base = 0;
  uint64_t range = 0;
  uint64_t bitmap_words = 0;  // allocated 32-bit words of `seen` (whole 2^20-bit slices)
  // caller-declared global value range (tgx_distinct_range_hint): congruent bitmaps on every rank
  bool has_hint = false;
  int64_t hint_lo = 0, hint_hi = 0;
  // range-partitioned population of the bitmap (big batches)
  DevBuf lists, cursors;
  // hash (wide = 128-bit fingerprint keys of a Utf8 column: two words per slot, 32-byte records)
  bool wide = false;
  DevBuf keys, dup;
  // EXACT string / tuple key sets (TGX_FLAG_EXACT_KEYS; kernels/distinct128.hip "EXACT key sets"): slots hold (first
  // fingerprint word, reference) and the keys' bytes live in `key_store` (8-byte words; word offsets stay valid when
  // the store moves to a bigger block).  key_cursor: [0] next free word (device), [1..2] scratch of the measuring pass.
  bool exact = false;  // from the plan's task; only consulted when the set is `wide`
  DevBuf key_store, key_cursor;
  uint64_t key_store_words = 0;
  // (slot, second word) of a batch's new keys between the insert and the commit: a region per wave of the insert kernel
  DevBuf key_pending, key_pending_counts;
  uint64_t key_pending_region = 0;
  uint32_t key_pending_waves = 0;
  // host-side bound on the store's fill (the cursor lives on the device): a batch whose worst case the host can bound --
  // a coalesced flush of a Utf8 / dictionary column: its bytes were counted as they were noted (`batch_data_bytes`) --
  // reserves against this bound and waits for nothing; only when the bound no longer fits is the real fill read back
  uint64_t key_words_ub = 0;
  bool batch_bytes_known = false;
  int64_t batch_data_bytes = 0;
  uint64_t capacity = 0;         // slots (power of two)
  uint64_t rows_upper_bound = 0; // host-side bound on keys in the table
  // counters (device) + host-side totals
  DevBuf counters;
  int64_t total_rows = 0;  // COUNT(*) over batches handled on this device
  // host-only part (deserialized / merged-in owner-partitioned partial states)
  bool partitioned = false;
  uint64_t h_total = 0, h_non_null = 0, h_distinct = 0, h_twice = 0, h_empty_rows = 0;
  // export scratch
  DevBuf export_records, export_counts;
  // Dictionary<Int32, Utf8> batches: per-entry reference counts (saturating at 2)
  DevBuf dict_usage, dict_scratch;  // seen | twice bitmaps over the entries; per-workgroup slices
  // The bitmap's range comes from a SAMPLE of the first batch (no scan of the column has to finish first, and the
  // DISTINCT pass takes the column's range aggregates along): keys that fall outside it after all are counted by the
  // kernels (kCntOutOfRange) and repaired at the next point the host looks at the state (distinct_resolve) from the
  // batches retained here -- DEVICE views, which the caller keeps alive until tgx_finalize / tgx_state_sync
  // (include/tgx.h); HOST batches are resolved before tgx_update returns.
  bool speculative = false;
  // the sampled extremes the bitmap was last laid out over.  They outlive tgx_state_reset: the next first batch of the
  // column takes them instead of a sample of its own (a read-back, i.e. the stream's latency, at the start of every
  // step of a runner that checks table after table of one shape); forgotten as soon as a repair finds a key outside
  bool remembered = false;
  int64_t remembered_lo = 0, remembered_hi = 0;
  // ... and which form the partition pass of the column's last batch took (counters[kCntForm]: 1 plain, 2 keys in
  // order; 0: not known): the next pass launches only that form (PartitionParams::force_form); outlives tgx_state_reset
  // like the range
  int32_t remembered_form = 0;
  // some batch since the last look at the counters may have left keys outside the bitmap's range (its range was a
  // sample's, or unknown: DEVICE buffers).  The range must then stay as it is until they have been repaired: the
  // repair walks the retained batches for the keys outside the range, so a bitmap grown over them in the meantime
  // would make it skip them (a stream of HOST and DEVICE batches of growing ids lost a whole batch that way).
  bool outliers_possible = false;
  // the value range of the batch about to be run, when the host knows it (a coalesced flush of HOST windows)
  bool batch_range_known = false;
  int64_t batch_lo = 0, batch_hi = 0;
  // the batch about to be run is a coalesced flush with DEVICE windows of this Int64 key column: the host has not seen
  // its values, so -- while that can still keep the key set on the bitmap -- the device takes the flush's exact
  // MIN / MAX before the pass (distinct_sample_all: one wait, the one a sample would cost)
  bool flush_device_keys = false;
  // the views, and for each the coalescing region set it points into (-1: the caller's own memory)
  struct Retained {
    std::vector<tgx_column> cols;
    std::vector<int8_t> region_set;
    void push_back(const tgx_column &c) {
      cols.push_back(c);
      region_set.push_back(-1);
    }
    void clear() {
      cols.clear();
      region_set.clear();
    }
    bool empty() const { return cols.empty(); }
    size_t size() const { return cols.size(); }
    std::vector<tgx_column>::const_iterator begin() const { return cols.begin(); }
    std::vector<tgx_column>::const_iterator end() const { return cols.end(); }
    const tgx_column &operator[](size_t i) const { return cols[i]; }
  } retained;
  DevBuf sample;         // DistinctSample
  bool sample_ready = false;  // `sample_host` holds this batch's sample (tgx_update reads all tasks' samples at once)
  DistinctSample sample_host;
  DevBuf stat_partials;  // ScanPartial per workgroup of the partition pass (PartitionParams::stats) + one for outliers
  DevBuf outlier_stats;  // OutlierStats
  // A big first Utf8 batch leaves its key set as partitioned fingerprint lists (kernels/distinct128.hip, fp_*): the
  // counters are exact, the table is filled from the lists only when something needs it (distinct_resolve).  A list
  // that overflowed (kCntOutOfRange) means the batch is redone through the table from the view retained above.
  bool fp_staged = false;
  uint64_t fp_cap1 = 0, fp_cap2 = 0;
  DevBuf fp_level1, fp_level2, fp_offered, fp_per_list;
  DevBuf fp_buffers;  // Utf8View batches: the retained view's table of data-buffer pointers (the update's own is staged)
  // exact sets on the lists: records carry (half of the second fingerprint word, row); the other half waits here, per
  // row, for the day the batch is released while the lists are still the key set (fp_demote_kernel)
  DevBuf fp_fb_lo;
  bool fp_exact_lists = false;
  // second bitmap pair: tgx_distinct_adopt_slices builds the owned slice here and swaps, so a state that is
  // reset and refilled every step never frees or allocates (hipMalloc/hipFree of 125 MB cost ~0.3 ms a step)
  DevBuf spare_seen, spare_twice;
};

// pattern checks of a dictionary column whose row gather rides on the column's DISTINCT pass (kernels/dict.hip)
struct DictGather {
  const uint8_t *hits;          // per-entry verdict bytes
  unsigned long long *counters;  // counters[0] receives the matches
  int32_t null_is_valid;
};
struct DictFuse {
  std::map<int, int> capacity;                         // column -> pattern checks the fused pass can take
  std::map<int, std::vector<DictGather>> by_column;    // filled by regex_update
};

// ---- library-side coalescing of small batches (tgx_api.cpp "coalescing", kernels/gather.hip) ----
// DataFusion streams 8192-row RecordBatches (TG/core/context.rs:28-38).  tgx_update only notes a small batch: one
// segment per used column (a HOST batch's windows are first copied into a pinned arena -- its buffers are borrowed only
// until the call returns); a flush uploads the arena, gathers every column's segments into one contiguous device
// column and runs the ordinary fused pass on that.  Two arenas and two sets of device regions take turns, so the host
// keeps copying batch k+1 while the device still works on flush k; nothing is synchronised per batch.
struct CoalesceSegment {
  const void *values;       // window start (device address: the caller's buffer or the arena's device twin)
  const uint8_t *validity;  // byte of the first validity bit, or nullptr
  const uint8_t *data;      // strings: first value byte of the window
  int64_t bit0;             // first validity bit within *validity
  int64_t length;
  int64_t data_first, data_len;
  int32_t index_shift = 0;  // dictionary index windows: first entry of the window's dictionary in the coalesced one
  int32_t stretches = -1;   // Utf8View windows: index into the column's `stretches`
};
// a Utf8View window's stretches of the variadic buffers its long views point into (GatherSeg, kind 3); kept beside the
// segments, not in them: a segment is noted per (column, batch), half a microsecond a batch all told
struct CoalesceStretches {
  int32_t vb_count = 0;
  int32_t vb_index[kGatherViewBufs] = {0, 0, 0, 0};
  int64_t vb_min[kGatherViewBufs] = {0, 0, 0, 0}, vb_len[kGatherViewBufs] = {0, 0, 0, 0};
  const uint8_t *vb_src[kGatherViewBufs] = {nullptr, nullptr, nullptr, nullptr};
};
// the dictionaries of a Dictionary<Int32, Utf8> column's pending windows, coalesced like a Utf8 column of their own
struct CoalesceDict {
  int type = 0;             // TGX_UTF8 / TGX_LARGE_UTF8
  bool any_validity = false;
  int64_t data_bytes = 0, entries = 0;
  std::vector<CoalesceSegment> segs;
  DevBuf values[2], validity[2], data[2];
  tgx_column view[2];       // what the coalesced column's `dictionary` points at (per region set)
  // the dictionary of the last noted window: batches of one file share theirs, it is taken once per flush
  const void *last_offsets = nullptr;
  const uint8_t *last_data = nullptr, *last_validity = nullptr;
  int64_t last_offset = 0, last_length = -1, last_base = 0;
};
struct CoalesceColumn {
  int type = 0;             // tgx_type of the pending segments
  bool any_validity = false;
  int64_t data_bytes = 0;   // strings: value bytes pending
  std::vector<CoalesceSegment> segs;
  std::vector<CoalesceStretches> stretches;  // (Utf8View columns)
  DevBuf values[2], validity[2], data[2];  // the coalesced column, per region set
  const uint8_t *view_buf[2] = {nullptr, nullptr};  // Utf8View: the coalesced column's one data buffer (`variadic`)
  std::unique_ptr<CoalesceDict> dict;               // Dictionary<Int32, Utf8>
  // Int64 key columns (a numeric DISTINCT check reads them): MIN / MAX of the pending HOST windows' non-NULL values,
  // taken while the windows are copied -- the flush then lays the range bitmap out (or grows it) for what it is about
  // to see instead of sampling the device copy and waiting for the answer.  Unknown once a DEVICE window is pending.
  bool range_known = true;
  int64_t range_lo = INT64_MAX, range_hi = INT64_MIN;
};
struct CoalesceCopy {
  void *dst;
  const void *src;
  size_t bytes;
  // mm_col >= 0: `src` is a window of Int64 key column mm_col; whoever copies the piece also takes the MIN / MAX of its
  // non-NULL values (row 0 of the piece = bit mm_bit0 of *mm_validity) into lo / hi
  int32_t mm_col = -1;
  const uint8_t *mm_validity = nullptr;
  int64_t mm_bit0 = 0;
  int64_t lo = INT64_MAX, hi = INT64_MIN;
  // widen != 0: not a byte copy -- `src` holds narrow integers (widen_mode(): Int8 .. UInt32) or Boolean bits (row 0 = bit
  // src_bit0 of *src) and `dst` takes the Int64 values they stand for; `bytes` counts the DESTINATION
  int32_t widen = 0, src_bit0 = 0;
  CoalesceCopy() = default;
  CoalesceCopy(void *d, const void *s, size_t b) : dst(d), src(s), bytes(b) {}
};
struct Coalescer {
  bool disabled = false;
  int64_t flush_rows = 0;   // 0: the default threshold (TGX_COALESCE_FLUSH_ROWS overrides it, for tests)
  int64_t rows = 0;         // rows pending
  size_t batches = 0;       // batches pending
  std::vector<int64_t> batch_rows;  // rows of each pending batch
  std::vector<CoalesceColumn> cols;  // per plan column
  // HOST batches: pinned arenas (and their device twins), guarded by an event recorded after the flush that used them
  void *arena_host[2] = {nullptr, nullptr};
  DevBuf arena_dev[2];
  size_t arena_cap[2] = {0, 0};
  size_t arena_want = 8u << 20;  // grows (to kCoalesceArenaMax) when flushes are forced by a full arena
  hipEvent_t arena_event[2] = {nullptr, nullptr};
  bool arena_busy[2] = {false, false};
  int arena_cur = 0;
  size_t arena_used = 0;
  // an arena goes up on a stream of its own (from the library's pool), the state's stream waits for it before the
  // gather: the upload of one flush then runs beside the kernels of the flush before it instead of behind them
  hipStream_t copy_stream = nullptr;
  hipEvent_t upload_done[2] = {nullptr, nullptr};
  uint64_t host_flushes = 0;  // flushes that carried HOST windows since the last reset (the first is a short one)
  // the segment table of a flush: pinned, uploaded, one per arena turn
  void *desc_host[2] = {nullptr, nullptr};
  size_t desc_cap[2] = {0, 0};
  DevBuf desc_dev[2];
  // device region set of the next flush, and what is known about the flush that last used each set: a snapshot of the
  // DISTINCT counters taken right after it (views retained into the set can be dropped without a wait when the
  // snapshot shows no key outside its bitmap and no overflowed list)
  int set_cur = 0;
  void *snap_host[2] = {nullptr, nullptr};
  size_t snap_cap[2] = {0, 0};
  hipEvent_t snap_event[2] = {nullptr, nullptr};
  bool snap_pending[2] = {false, false};
  bool flushing = false;
  std::vector<CoalesceCopy> copy_jobs, copy_tail;  // the HOST windows of the batch being noted (scratch)
  // TGX_MEM_HOST_RETAINED batches: the copies into the arena being filled, and the key columns' windows whose MIN / MAX
  // the host takes, both left for the flush (all of them at once, on every copy thread)
  std::vector<CoalesceCopy> deferred;
  // ... and the ones the copy threads are already working on while the caller goes on noting batches (started every
  // few MB of retained windows, waited for at the flush): the pieces a worker was given stay where they are until then
  struct Inflight {
    std::vector<CoalesceCopy> cut;  // every worker's share, one behind the other
    int ids[8];
    int helpers = 0;
  };
  std::vector<Inflight> inflight;
  size_t deferred_bytes = 0;  // bytes of `deferred`
  uint64_t flushes = 0, coalesced_batches = 0;  // statistics (tgx_profile_get "coalesce_flushes" / "coalesced_batches")
};

struct ProfileEntry {
  double total_ms = 0;
  uint64_t launches = 0;
  uint64_t bytes = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  std::vector<uint64_t> pending_bytes;
};

}  // namespace tgx

struct tgx_state {
  const tgx_plan *plan = nullptr;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool device_ready = false;  // device-side buffers allocated
  int64_t batches = 0;
  std::vector<int> col_types;  // per plan column, 0 = not seen yet

  // device accumulators
  tgx::DevBuf d_scan_acc, d_count_acc, d_como_acc, d_pivots, d_pivot_set;
  tgx::DevBuf d_scan_identity;  // (unused since round 5: state_reset_kernel writes the identities itself)
  struct Widen {
    const void *src;
    void *dst;
    int64_t n;
    int mode;  // widen_mode()
  };
  std::vector<int> como_pivot_tries;  // per COMOMENTS task: batches that offered the pivot kernel a look (<= 4)
  std::vector<Widen> pending_widen;  // TGX_INT32 / TGX_FLOAT32 windows of the current update (stage_column)
  tgx::DevBuf d_distinct_counters;  // [distinct task][kNumDistinctCounters]: every DistinctState::counters is a slice
  // per-update scratch
  tgx::DevBuf d_scan_partials, d_count_blocks, d_como_partials;
  std::vector<std::unique_ptr<tgx::DevBuf>> staging;  // host columns copied to the device
  size_t staging_used = 0;
  // small HOST buffers of one update are gathered in a pinned arena and cross PCIe in ONE copy (a DataFusion batch
  // is 8192 rows: a dozen 64-KiB buffers, 12 us of call overhead each when copied one by one)
  // two arenas take turns, each guarded by an event recorded after the update that used it, so an update whose
  // HOST buffers all fitted returns without synchronising the stream
  void *arena_host[2] = {nullptr, nullptr};  // pinned_alloc(kArenaBytes)
  tgx::DevBuf arena_dev[2];
  hipEvent_t arena_event[2] = {nullptr, nullptr};
  bool arena_busy[2] = {false, false};
  int arena_cur = 0;
  size_t arena_used = 0;
  // small read-backs (the samples of the key columns, the accumulators at finalize) land in pinned memory: copies
  // into pageable memory are staged one by one (each waits for the one before), pinned ones are queued together and
  // cost ONE wait
  void *h_pinned = nullptr;
  size_t h_pinned_cap = 0;
  bool host_direct = false;  // this update copied a HOST buffer straight from the caller's memory
  std::deque<tgx_column> dict_views;  // device views of the dictionaries of the batch being updated
  // host copies of Utf8View buffer-pointer tables whose asynchronous upload may still be pending; dropped
  // wherever the stream is synchronized (gather / reset)
  std::deque<std::vector<const uint8_t *>> ptr_tables;

  // host accumulators: contributions merged in from other states / deserialized blobs
  std::vector<tgx::ScanAcc> h_scan;
  std::vector<tgx::CountAcc> h_count;
  std::vector<tgx::ComomentAcc> h_como;
  std::vector<tgx::DistinctState> distinct;
  std::vector<tgx::KllHost> h_kll;
  // HyperLogLog tasks: running registers on the device ([task][kHllRegisters] bytes), the per-launch rows of the
  // workgroups, and what was merged in on the host (byte-wise max); which side answers each task: 0 undecided,
  // 1 the registers, 2 the exact key set of `distinct_slot`
  tgx::DevBuf d_hll, d_hll_rows;
  std::vector<std::vector<uint8_t>> h_hll;
  std::vector<int> hll_mode;
  void *kll = nullptr;    // tgx::KllDeviceState (kll_device.cpp)
  void *regex = nullptr;  // tgx::RegexState (regex_device.cpp)
  void *spearman = nullptr;  // tgx::SpearmanState (spearman_device.cpp)

  tgx::Coalescer coalesce;
  std::vector<tgx::DevBuf> parked;  // buffers replaced while the stream may still read them; freed once it has drained
  // cross-rank overlap (allreduce.cpp): `keys_ready` is recorded right after the key columns' uniqueness passes of an
  // update; tgx_allreduce runs its facts round and the key-set exchange on `aux_stream` behind that event while the
  // state's own stream is still scanning, and joins the two with `aux_done`
  hipStream_t aux_stream = nullptr;
  hipEvent_t keys_ready = nullptr, aux_done = nullptr;
  // the key columns' uniqueness passes BESIDE the scan (update.cpp, round 6): they are queued on `key_stream` behind
  // `batch_in` (what the state's stream held when the update began) and the state's stream waits for `keys_ready`
  // once the scan is queued
  hipStream_t key_stream = nullptr;
  hipEvent_t batch_in = nullptr;
  bool keys_ready_recorded = false;  // since the last reset, and standing for EVERY key set of the plan
  int64_t passes = 0;                // fused passes (batches or flushes) with rows since the last reset
  bool exchange_expected = false;    // the state has been through tgx_allreduce: its scans leave room for the exchange
  bool profiling = false;
  std::map<std::string, tgx::ProfileEntry> profile;
};

// ---- shared between tgx_api.cpp and allreduce.cpp ------------------------------------------------
namespace tgx {
tgx_status fail(tgx_error *err, tgx_status code, const char *fmt, ...);
tgx_status need_device(tgx_error *err);
tgx_status state_init_device(tgx_state *st, tgx_error *err);
// partitions the task's key set by owner = mix(key) % world into runs of KeyRecord / KeyRecord128 (device memory
// owned by the state); counts[r] = records for rank r
tgx_status distinct_export_impl(tgx_state *st, size_t slot, uint32_t world, const void **device_records,
                                uint64_t *counts, tgx_error *err);
// unites `n` device records into the task's set (switching it to hash mode)
tgx_status distinct_import_records(tgx_state *st, size_t slot, const void *d_recs, uint64_t n, bool wide,
                                   tgx_error *err);
// brings in the keys that fell outside a sampled bitmap range (see DistinctState::speculative); a no-op otherwise
tgx_status distinct_resolve(tgx_state *st, size_t slot, tgx_error *err);
tgx_status distinct_resolve_all(tgx_state *st, tgx_error *err);
// runs the small batches tgx_update has only noted so far (no-op when none are pending); called by every entry point
// that looks at the state
tgx_status coalesce_flush(tgx_state *st, tgx_error *err);
int num_cus();
int device_id();
void bind_thread();  // hipSetDevice(the device tgx_init selected) for the calling thread
}  // namespace tgx

