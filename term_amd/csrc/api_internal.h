// api_internal.h -- what the translation units behind include/tgx.h share: tgx_api.cpp (handles, plans, states,
// profiling, finalize), update.cpp (staging and the fused pass of a batch), distinct_state.cpp (the key sets' host-side
// bookkeeping, export / import / merge), coalesce.cpp (small batches noted and gathered, the copy pool) and wire.cpp
// (state blobs).  The helpers declared here are shared between those files only (hidden visibility).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "internal.h"
#include "kll_device.h"
#include "regex_device.h"
#include "spearman_device.h"

using namespace tgx;

#define TGX_HIDDEN __attribute__((visibility("hidden")))

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(err, e_ == hipErrorOutOfMemory ? TGX_OUT_OF_MEMORY : TGX_DEVICE_ERROR,      \
                  "%s failed: %s", #expr, hipGetErrorString(e_));                             \
  } while (0)

#define TGX_TRY(expr)                  \
  do {                                 \
    tgx_status s_ = (expr);            \
    if (s_ != TGX_OK) return s_;       \
  } while (0)

typedef long double xdouble;
constexpr size_t kArenaBytes = 8u << 20;        // pinned staging arena per state
constexpr size_t kArenaMaxBuffer = 256u << 10;  // buffers up to this size go through it
constexpr int64_t kCoalesceMaxRows = 1 << 16;        // batches up to this many rows are coalesced
constexpr int64_t kCoalesceFlushRows = 4 << 20;      // pending rows that trigger a flush
constexpr size_t kCoalesceFlushBatches = 4096;       // pending batches that trigger a flush
constexpr size_t kCoalesceArenaMax = 128u << 20;     // pinned staging per arena turn (HOST batches)
constexpr uint32_t kWireMagic = 0x53584754;  // "TGXS"
constexpr uint32_t kWireVersion = 3;  // 2: ComomentAcc carries its pivots; 3: { u32 keyed, u8 key[16] } behind the head

struct Context {
  std::mutex mu;
  bool inited = false;
  int device = -1;
  int n_cu = 256;
  uint64_t distinct_hint = 0;
  bool no_coalesce = false;
  char arch[64] = {0};
};
extern TGX_HIDDEN Context g_ctx;

struct NumericPrep {
  bool prepared = false;
  bool partitioned = false;  // the batch goes through partition_kernel / bucket_apply_kernel
  uint32_t sub_bits = 0;
  bool key16 = false;
  uint64_t n_buckets = 0;
};

struct BatchTraits {
  bool any_host = false, any_utf8 = false;
  bool any_strings = false;  // a used column is Utf8 / LargeUtf8 / Utf8View / a dictionary
  bool coalescible = true;  // every used column is of a kind the segment gather takes (kernels/gather.hip)
  bool retained = false;    // the batch's HOST columns were all given as TGX_MEM_HOST_RETAINED: their copies may wait for the flush
};

struct DistinctTotals {
  uint64_t total = 0, non_null = 0, distinct = 0, twice = 0, empty_rows = 0;
};
struct Gathered {
  std::vector<ScanAcc> scan;
  std::vector<CountAcc> count;
  std::vector<ComomentAcc> como;
  std::vector<DistinctTotals> distinct;
  std::vector<std::vector<uint8_t>> hll;  // per task: kHllRegisters bytes, or empty (nothing seen)
};

struct WindowPrep {
  bool ok = true;  // false: more than kGatherViewBufs buffers referenced -- the batch takes the immediate path
  int32_t vb_count = 0;
  int32_t vb_index[kGatherViewBufs];
  int64_t vb_min[kGatherViewBufs], vb_end[kGatherViewBufs];
  bool new_dict = false;
  int64_t dict_first = 0, dict_end = 0;  // value bytes of the new dictionary's window
};

static inline bool is_numeric(int t) { return t == TGX_INT64 || t == TGX_FLOAT64; }
static inline bool is_numeric32(int t) { return t == TGX_INT32 || t == TGX_FLOAT32; }
// the narrow types of round 5: no kernel reads them in place, every pass sees them widened to Int64
static inline bool is_narrow_int(int t) { return (t >= TGX_INT8 && t <= TGX_UINT32) || t == TGX_BOOL; }
// everything stage_column widens into a staging buffer (4-byte numerics only for the passes that need 8-byte values)
static inline bool is_widened(int t) { return is_numeric32(t) || is_narrow_int(t); }
// COUNT and DISTINCT only (include/tgx.h)
static inline bool is_keys_only(int t) { return t == TGX_UINT64 || t == TGX_BOOL; }
// bytes the first `slots` slots of a values buffer of type `t` take (Boolean: bits)
static inline size_t narrow_bytes(int t, int64_t slots) {
  switch (t) {
    case TGX_INT8: case TGX_UINT8: return (size_t)slots;
    case TGX_INT16: case TGX_UINT16: return (size_t)slots * 2;
    case TGX_BOOL: return (size_t)((slots + 7) / 8);
    default: return (size_t)slots * 4;  // Int32, Float32, UInt32
  }
}
// the type a widened column is seen as, and the widen kernel's mode (kernels/scan.hip, widen_kernel)
static inline int widened_type(int t) { return t == TGX_FLOAT32 ? TGX_FLOAT64 : TGX_INT64; }
static inline int widen_mode(int t) {
  switch (t) {
    case TGX_FLOAT32: return 1;
    case TGX_INT8: return 2;
    case TGX_INT16: return 3;
    case TGX_UINT8: return 4;
    case TGX_UINT16: return 5;
    case TGX_UINT32: return 6;
    case TGX_BOOL: return 7;
    default: return 0;  // Int32
  }
}
static inline bool is_string(int t) { return t == TGX_UTF8 || t == TGX_LARGE_UTF8; }
static inline bool is_any_string(int t) { return is_string(t) || t == TGX_UTF8_VIEW; }

TGX_HIDDEN void copy_pool_shutdown();
TGX_HIDDEN void coalesce_drop(tgx_state *st);
TGX_HIDDEN ScanAcc scan_acc_identity();
TGX_HIDDEN void host_two_sum(double &s, double &c, double x);
TGX_HIDDEN void scan_acc_merge(ScanAcc &a, const ScanAcc &b);
TGX_HIDDEN xdouble como_sum(const ComomentAcc &a, int k);
TGX_HIDDEN void como_store(ComomentAcc &a, int k, xdouble v);
TGX_HIDDEN void como_rebase(ComomentAcc &b, double px, double py);
TGX_HIDDEN void como_acc_merge(ComomentAcc &a, const ComomentAcc &b_in);
TGX_HIDDEN void state_init_host(tgx_state *st, const tgx_plan *plan);
TGX_HIDDEN void prof_begin(tgx_state *st, const char *name, uint64_t bytes, hipEvent_t *e0, hipEvent_t *e1);
TGX_HIDDEN void prof_end(tgx_state *st, const char *name, hipEvent_t e0, hipEvent_t e1);
TGX_HIDDEN void prof_resolve(tgx_state *st);
TGX_HIDDEN tgx_status distinct_totals(tgx_state *st, size_t slot, DistinctTotals *t, tgx_error *err,
                                  const unsigned long long *pre = nullptr);
TGX_HIDDEN tgx_status gather(tgx_state *st, Gathered *g, tgx_error *err);
TGX_HIDDEN double key_to_double(int64_t k);
TGX_HIDDEN double i128_to_double(uint64_t lo, int64_t hi);
TGX_HIDDEN void fill_stats(const ScanAcc &a, bool variance, tgx_result *r);
TGX_HIDDEN double hll_sigma(double x);
TGX_HIDDEN double hll_tau(double x);
TGX_HIDDEN uint64_t hll_estimate(const std::vector<uint8_t> &regs);
TGX_HIDDEN tgx_status stage_column(tgx_state *st, const tgx_column &c, tgx_column *out, tgx_error *err,
                               bool widen32 = true);
TGX_HIDDEN void fill_scan_desc(const tgx_column &c, bool variance, const double *pivot, ScanColDesc *d);
TGX_HIDDEN int scan_blocks_for(const ScanColDesc &d, int n_cols_in_launch, int per_cu = 8);
TGX_HIDDEN tgx_status como_pivots(tgx_state *st, const ComomentLaunch &L, int n_pairs, tgx_error *err);
TGX_HIDDEN tgx_status update_validate(const tgx_plan *plan, tgx_state *st, const tgx_column *columns, size_t n_columns,
                                  int64_t *nrows_out, BatchTraits *traits, tgx_error *err);
TGX_HIDDEN tgx_status update_impl(const tgx_plan *plan, tgx_state *st, const tgx_column *columns, int64_t nrows,
                              tgx_error *err);
TGX_HIDDEN tgx_status coalesce_append(const tgx_plan *plan, tgx_state *st, const tgx_column *columns, int64_t nrows,
                                  const BatchTraits &traits, bool *taken, tgx_error *err);
TGX_HIDDEN uint64_t next_pow2(uint64_t x);
TGX_HIDDEN tgx_status distinct_read_counters(tgx_state *st, DistinctState &ds, unsigned long long *out,
                                         tgx_error *err);
TGX_HIDDEN HashSetView hash_view(const DistinctState &ds);
TGX_HIDDEN BitmapView bitmap_view(const DistinctState &ds);
TGX_HIDDEN tgx_status hash_alloc(tgx_state *st, DevBuf &keys, DevBuf &dup, uint64_t capacity, bool mult,
                             bool wide, tgx_error *err);
TGX_HIDDEN tgx_status hash_ensure(tgx_state *st, DistinctState &ds, bool mult, uint64_t incoming,
                              tgx_error *err);
TGX_HIDDEN tgx_status bitmap_to_hash(tgx_state *st, DistinctState &ds, bool mult, uint64_t incoming,
                                 tgx_error *err);
TGX_HIDDEN tgx_status tuple_desc_of(const std::vector<const tgx_column *> &cols, bool mult, TupleDesc *d, tgx_error *err);
TGX_HIDDEN bool fp_lists_fit_rows(int64_t rows);
TGX_HIDDEN tgx_status fp_lists_tuple_update(tgx_state *st, size_t slot, const TupleDesc &d,
                                        const std::vector<const tgx_column *> &cols, tgx_error *err);
TGX_HIDDEN tgx_status distinct_tuple_update(tgx_state *st, size_t slot, const tgx_column *dev, tgx_error *err,
                                        const tgx_column *orig = nullptr);
TGX_HIDDEN tgx_status distinct_prepare_numeric(tgx_state *st, size_t slot, const tgx_column &c, NumericPrep *prep,
                                           tgx_error *err);
TGX_HIDDEN tgx_status distinct_run_numeric(tgx_state *st, size_t slot, const tgx_column &c, const NumericPrep &prep,
                                       int stats_slot, tgx_error *err, const tgx_column *orig = nullptr);
TGX_HIDDEN uint64_t fp_list_cap(int64_t rows, uint64_t lists);
TGX_HIDDEN bool fp_lists_fit(const tgx_column &c);
TGX_HIDDEN void fp_views(const DistinctState &ds, FpLists *l1, FpLists *l2);
TGX_HIDDEN tgx_status fp_lists_prepare(tgx_state *st, DistinctState &ds, int64_t rows, size_t rec_bytes, tgx_error *err);
TGX_HIDDEN tgx_status fp_lists_update(tgx_state *st, size_t slot, const tgx_column &c, tgx_error *err);
TGX_HIDDEN tgx_status distinct_update(tgx_state *st, size_t slot, const tgx_column &c, tgx_error *err,
                                  const std::vector<DictGather> *gathers = nullptr, const NumericPrep *ready = nullptr,
                                  int stats_slot = -1, const tgx_column *orig = nullptr);
TGX_HIDDEN void bitmap_shape(const DistinctState &ds, int64_t length, bool mult, uint32_t *sub_bits_out, bool *key16_out,
                         uint64_t *n_buckets_out, bool *partitioned_out);
TGX_HIDDEN tgx_status pinned_readback(tgx_state *st, size_t bytes, tgx_error *err);
TGX_HIDDEN bool distinct_wants_sample(const DistinctState &ds, const tgx_column &c);
TGX_HIDDEN bool distinct_wants_exact_range(const DistinctState &ds, const tgx_column &c);
TGX_HIDDEN tgx_status distinct_sample_all(tgx_state *st, const tgx_column *dev, tgx_error *err);
TGX_HIDDEN tgx_status bitmap_grow(tgx_state *st, DistinctState &ds, bool mult, int64_t lo, int64_t hi, int64_t incoming,
                              tgx_error *err);
TGX_HIDDEN tgx_status retained_numeric_view(tgx_state *st, const tgx_column &col, std::vector<std::unique_ptr<DevBuf>> &tmp,
                                        tgx_column *out, tgx_error *err);
TGX_HIDDEN tgx_status distinct_slot_of(const tgx_plan *plan, tgx_state *st, size_t spec_index, size_t *slot,
                                   tgx_error *err);
TGX_HIDDEN void stream_copy(void *dst, const void *src, size_t bytes);
TGX_HIDDEN void host_minmax_i64_plain(const int64_t *v, const uint8_t *validity, int64_t bit0, int64_t n, int64_t *lo,
                                  int64_t *hi);
TGX_HIDDEN void host_minmax_i64(const int64_t *v, const uint8_t *validity, int64_t bit0, int64_t n, int64_t *lo, int64_t *hi);
TGX_HIDDEN tgx_status coalesce_arena_ready(tgx_state *st, tgx_error *err);
TGX_HIDDEN tgx_status coalesce_prepare_window(const tgx_column &c, int64_t nrows, const CoalesceColumn &cc, int col,
                                          WindowPrep *w, tgx_error *err);
TGX_HIDDEN size_t coalesce_host_bytes(const tgx_plan *plan, const tgx_column *columns, int64_t nrows,
                                  const std::vector<WindowPrep> &prep);
TGX_HIDDEN tgx_status coalesce_release_set(tgx_state *st, int set, tgx_error *err);

struct ProfScope {
  tgx_state *st;
  const char *name;
  hipEvent_t e0, e1;
  ProfScope(tgx_state *s, const char *n, uint64_t bytes) : st(s), name(n) { prof_begin(s, n, bytes, &e0, &e1); }
  ~ProfScope() { prof_end(st, name, e0, e1); }
};

