// tgx_api.cpp -- the C ABI of include/tgx.h: plans, states, batch updates, merges, results.
//
// Host-side only bookkeeping lives here; every per-row computation is a HIP kernel under
// kernels/.  There is no CPU fallback: without a gfx950 device the compute entry points fail.
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

// ------------------------------------------------------------------------------------------------
// errors
tgx_status tgx::fail(tgx_error *err, tgx_status code, const char *fmt, ...) {
  if (err) {
    err->code = (int32_t)code;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err->msg, sizeof(err->msg), fmt, ap);
    va_end(ap);
  }
  return code;
}

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

extern "C" const char *tgx_status_name(int32_t s) {
  switch (s) {
    case TGX_OK: return "TGX_OK";
    case TGX_INVALID_ARGUMENT: return "TGX_INVALID_ARGUMENT";
    case TGX_UNSUPPORTED: return "TGX_UNSUPPORTED";
    case TGX_DEVICE_ERROR: return "TGX_DEVICE_ERROR";
    case TGX_OUT_OF_MEMORY: return "TGX_OUT_OF_MEMORY";
    case TGX_INTERNAL: return "TGX_INTERNAL";
    case TGX_NO_DEVICE: return "TGX_NO_DEVICE";
    default: return "TGX_UNKNOWN";
  }
}

extern "C" uint32_t tgx_abi_version(void) { return TGX_ABI_VERSION; }

// ------------------------------------------------------------------------------------------------
// device context
namespace {
struct Context {
  std::mutex mu;
  bool inited = false;
  int device = -1;
  int n_cu = 256;
  uint64_t distinct_hint = 0;
  bool no_coalesce = false;
  char arch[64] = {0};
} g_ctx;
}  // namespace

namespace tgx {
int tgx_num_cus() { return g_ctx.n_cu > 0 ? g_ctx.n_cu : 256; }
int num_cus() { return tgx_num_cus(); }
int device_id() { return g_ctx.device < 0 ? 0 : g_ctx.device; }
}  // namespace tgx

extern "C" tgx_status tgx_init(const tgx_options *opts, tgx_error *err) try {
  std::lock_guard<std::mutex> lock(g_ctx.mu);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(err, TGX_NO_DEVICE, "no HIP device visible (%s); libtgx has no CPU path",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  int dev = opts ? opts->device_id : -1;
  if (dev < 0) {
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  }
  if (dev >= count) return fail(err, TGX_INVALID_ARGUMENT, "device_id %d out of range (%d devices)", dev, count);
  HIP_TRY(hipSetDevice(dev));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(err, TGX_NO_DEVICE, "device %d is %s; libtgx kernels are built for gfx950 only", dev,
                prop.gcnArchName);
  g_ctx.device = dev;
  g_ctx.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  snprintf(g_ctx.arch, sizeof(g_ctx.arch), "%s", prop.gcnArchName);
  g_ctx.distinct_hint = opts ? opts->distinct_capacity_hint : 0;
  g_ctx.no_coalesce = opts && (opts->flags & TGX_OPT_NO_COALESCE) != 0;
  g_ctx.inited = true;
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

static void copy_pool_shutdown();
extern "C" tgx_status tgx_shutdown(void) try {
  copy_pool_shutdown();  // (the helper threads of the coalescing arenas' copies: stopped and joined)
  std::lock_guard<std::mutex> lock(g_ctx.mu);
  g_ctx.inited = false;
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(nullptr);
}

// HIP's current device is per THREAD: the caller's evaluate() may run on any tokio worker ("callable
// from arbitrary threads"), and a thread that never called hipSetDevice launches on device 0 against streams and
// buffers of device k.  Every entry point that touches HIP binds its thread to the device tgx_init selected first.
void tgx::bind_thread() {
  if (g_ctx.inited) (void)hipSetDevice(g_ctx.device);
}

tgx_status tgx::need_device(tgx_error *err) {
  if (!g_ctx.inited)
    return fail(err, TGX_NO_DEVICE, "tgx_init has not succeeded: no gfx950 device, and libtgx has no CPU path");
  return TGX_OK;
}

// ------------------------------------------------------------------------------------------------
// plan
extern "C" tgx_status tgx_plan_create(const tgx_check_spec *specs, size_t n_specs, tgx_plan **out,
                                      tgx_error *err) try {
  if (!out) return fail(err, TGX_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (n_specs > 0 && !specs) return fail(err, TGX_INVALID_ARGUMENT, "specs is NULL");
  std::unique_ptr<tgx_plan> plan(new tgx_plan());
  plan->specs.assign(specs, specs + n_specs);
  plan->patterns.resize(n_specs);
  plan->bind.resize(n_specs);
  auto find_scan = [&](int col) -> int {
    for (size_t i = 0; i < plan->scan.size(); i++)
      if (plan->scan[i].column == col) return (int)i;
    return -1;
  };
  auto need_scan = [&](int col, bool var) -> int {
    int s = find_scan(col);
    if (s < 0) {
      plan->scan.push_back({col, var});
      s = (int)plan->scan.size() - 1;
    } else if (var) {
      plan->scan[s].variance = true;
    }
    return s;
  };
  int max_col = -1;
  // pass 1: everything except COUNT
  for (size_t i = 0; i < n_specs; i++) {
    tgx_check_spec &sp = plan->specs[i];
    std::vector<int> tuple;
    if (sp.kind == TGX_CHECK_DISTINCT && sp.n_columns >= 2) {
      if (sp.n_columns > (uint32_t)kMaxTupleCols || !sp.columns)
        return fail(err, TGX_UNSUPPORTED, "spec %zu: DISTINCT over %u columns (2..%d supported)", i, sp.n_columns,
                    kMaxTupleCols);
      for (uint32_t k = 0; k < sp.n_columns; k++) {
        if (sp.columns[k] < 0) return fail(err, TGX_INVALID_ARGUMENT, "spec %zu: negative column index", i);
        tuple.push_back(sp.columns[k]);
        max_col = std::max(max_col, sp.columns[k]);
      }
      sp.column = tuple[0];
    } else if (sp.kind != TGX_CHECK_DISTINCT && sp.n_columns >= 2) {
      return fail(err, TGX_INVALID_ARGUMENT, "spec %zu: only DISTINCT takes a column list", i);
    }
    sp.columns = nullptr;  // the caller's array is not kept: the task holds its own copy
    sp.n_columns = (uint32_t)tuple.size();
    if (sp.column < 0) return fail(err, TGX_INVALID_ARGUMENT, "spec %zu: negative column index", i);
    max_col = std::max(max_col, sp.column);
    plan->bind[i].kind = sp.kind;
    plan->bind[i].slot = -1;
    plan->bind[i].count_src = Source::kCount;
    switch (sp.kind) {
      case TGX_CHECK_COUNT:
        break;
      case TGX_CHECK_NUMERIC_STATS:
        plan->bind[i].slot = need_scan(sp.column, (sp.flags & TGX_FLAG_VARIANCE) != 0);
        break;
      case TGX_CHECK_DISTINCT: {
        int slot = -1;
        for (size_t d = 0; d < plan->distinct.size(); d++)
          if (plan->distinct[d].column == sp.column && plan->distinct[d].tuple == tuple) slot = (int)d;
        bool mult = (sp.flags & TGX_FLAG_MULTIPLICITY) != 0;
        if (slot < 0) {
          plan->distinct.push_back({sp.column, mult, -1, tuple});
          slot = (int)plan->distinct.size() - 1;
        } else if (mult) {
          plan->distinct[slot].multiplicity = true;
        }
        plan->distinct[slot].approx_only = false;  // (an APPROX_DISTINCT spec may have created it)
        plan->bind[i].slot = slot;
        break;
      }
      case TGX_CHECK_APPROX_DISTINCT: {
        int slot = -1;
        for (size_t h = 0; h < plan->hll.size(); h++)
          if (plan->hll[h].column == sp.column) slot = (int)h;
        if (slot < 0) {
          // the exact key set that answers where the lane does not apply: shared with a DISTINCT check of the column
          int ds = -1;
          for (size_t d = 0; d < plan->distinct.size(); d++)
            if (plan->distinct[d].column == sp.column && plan->distinct[d].tuple.empty()) ds = (int)d;
          if (ds < 0) {
            DistinctTask t{sp.column, false, -1, {}};
            t.approx_only = true;
            plan->distinct.push_back(t);
            ds = (int)plan->distinct.size() - 1;
          }
          plan->hll.push_back({sp.column, -1, ds});
          slot = (int)plan->hll.size() - 1;
        }
        plan->bind[i].slot = slot;
        break;
      }
      case TGX_CHECK_COMOMENTS: {
        if (sp.column2 < 0) return fail(err, TGX_INVALID_ARGUMENT, "spec %zu: COMOMENTS needs column2", i);
        max_col = std::max(max_col, sp.column2);
        int slot = -1;
        for (size_t c = 0; c < plan->como.size(); c++)
          if (plan->como[c].col_x == sp.column && plan->como[c].col_y == sp.column2) slot = (int)c;
        if (slot < 0) {
          plan->como.push_back({sp.column, sp.column2});
          slot = (int)plan->como.size() - 1;
        }
        plan->bind[i].slot = slot;
        break;
      }
      case TGX_CHECK_KLL: {
        if (sp.kll_k < 2) return fail(err, TGX_INVALID_ARGUMENT, "spec %zu: k must be at least 2", i);
        int slot = -1;
        for (size_t c = 0; c < plan->kll.size(); c++)
          if (plan->kll[c].column == sp.column && plan->kll[c].k == sp.kll_k) slot = (int)c;
        if (slot < 0) {
          plan->kll.push_back({sp.column, sp.kll_k});
          slot = (int)plan->kll.size() - 1;
        }
        plan->bind[i].slot = slot;
        break;
      }
      case TGX_CHECK_LENGTH: {
        int slot = -1;
        tgx_status st = regex_plan_add(plan.get(), (int)i, &slot, err);
        if (st != TGX_OK) return st;
        plan->bind[i].slot = slot;
        break;
      }
      case TGX_CHECK_REGEX_MATCH: {
        if (!sp.pattern && sp.pattern_len) return fail(err, TGX_INVALID_ARGUMENT, "spec %zu: pattern is NULL", i);
        plan->patterns[i].assign(sp.pattern ? sp.pattern : "", sp.pattern_len);
        int slot = -1;
        tgx_status st = regex_plan_add(plan.get(), (int)i, &slot, err);
        if (st != TGX_OK) return st;
        plan->bind[i].slot = slot;
        break;
      }
      case TGX_CHECK_SPEARMAN: {
        max_col = std::max(max_col, sp.column2);
        int slot = -1;
        tgx_status st = spearman_plan_add(plan.get(), (int)i, &slot, err);
        if (st != TGX_OK) return st;
        plan->bind[i].slot = slot;
        break;
      }
      default:
        return fail(err, TGX_INVALID_ARGUMENT, "spec %zu: unknown check kind %d", i, sp.kind);
    }
  }
  // Int64 DISTINCT columns want the running MIN/MAX for the range-bitmap decision
  for (auto &d : plan->distinct)
    if (d.tuple.empty()) d.scan_slot = need_scan(d.column, false);
  for (auto &h : plan->hll) h.scan_slot = need_scan(h.column, false);
  for (size_t i = 0; i < n_specs; i++)
    if (plan->specs[i].kind == TGX_CHECK_NUMERIC_STATS) plan->scan[plan->bind[i].slot].stats_needed = true;
  for (auto &d : plan->distinct)
    if (d.tuple.empty() && !d.approx_only && d.scan_slot >= 0) plan->scan[d.scan_slot].stats_needed = true;
  // pass 2: COUNT rides on a scan of the same column when there is one
  for (size_t i = 0; i < n_specs; i++) {
    tgx_check_spec &sp = plan->specs[i];
    if (sp.kind != TGX_CHECK_COUNT) continue;
    int s = find_scan(sp.column);
    if (s >= 0) {
      plan->bind[i].slot = s;
      plan->bind[i].count_src = Source::kScan;
      continue;
    }
    int slot = -1;
    for (size_t c = 0; c < plan->count.size(); c++)
      if (plan->count[c].column == sp.column) slot = (int)c;
    if (slot < 0) {
      plan->count.push_back({sp.column});
      slot = (int)plan->count.size() - 1;
    }
    plan->bind[i].slot = slot;
    plan->bind[i].count_src = Source::kCount;
  }
  plan->n_columns_needed = max_col + 1;
  regex_plan_finish(plan.get());
  // which columns the plan touches, whose values it reads, and which 4-byte numeric columns a pass needs widened
  {
    tgx_plan *P = plan.get();
    P->used.assign(P->n_columns_needed, 0);
    P->reads_values.assign(P->n_columns_needed, 0);
    P->needs_wide.assign(P->n_columns_needed, 0);
    P->key_column.assign(P->n_columns_needed, 0);
    for (auto &t : P->distinct)
      if (t.tuple.empty() && !t.approx_only) P->key_column[t.column] = 1;
    for (auto &t : P->scan) P->used[t.column] = P->reads_values[t.column] = 1;
    for (auto &t : P->count) P->used[t.column] = 1;
    for (auto &t : P->distinct) {
      P->used[t.column] = P->reads_values[t.column] = 1;
      // (an approx_only key set of a numeric column only runs when the column's scan carries variance lanes)
      if (!t.approx_only || (t.scan_slot >= 0 && P->scan[t.scan_slot].variance)) P->needs_wide[t.column] = 1;
      for (int c : t.tuple) P->used[c] = P->reads_values[c] = P->needs_wide[c] = 1;
    }
    for (auto &t : P->como)
      P->used[t.col_x] = P->used[t.col_y] = P->reads_values[t.col_x] = P->reads_values[t.col_y] =
          P->needs_wide[t.col_x] = P->needs_wide[t.col_y] = 1;
    for (auto &t : P->kll) P->used[t.column] = P->reads_values[t.column] = P->needs_wide[t.column] = 1;
    regex_mark_used(P, P->used);
    std::vector<char> sp_used(P->n_columns_needed, 0), sp_vals(P->n_columns_needed, 0);
    spearman_mark_used(P, sp_used, sp_vals);
    for (int i = 0; i < P->n_columns_needed; i++) {
      P->used[i] |= sp_used[i];
      P->reads_values[i] |= sp_vals[i];
      P->needs_wide[i] |= sp_used[i];
    }
  }
  // re-point pattern pointers at the plan-owned copies
  for (size_t i = 0; i < n_specs; i++) {
    plan->specs[i].pattern = plan->patterns[i].empty() ? nullptr : plan->patterns[i].data();
    plan->specs[i].pattern_len = plan->patterns[i].size();
  }
  *out = plan.release();
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

extern "C" void tgx_plan_destroy(tgx_plan *plan) {
  if (!plan) return;
  regex_plan_free(plan);
  spearman_plan_free(plan);
  delete plan;
}
extern "C" size_t tgx_plan_num_specs(const tgx_plan *plan) { return plan ? plan->specs.size() : 0; }

// ------------------------------------------------------------------------------------------------
// state
static void coalesce_drop(tgx_state *st);
static ScanAcc scan_acc_identity() {
  ScanAcc a;
  memset(&a, 0, sizeof(a));
  a.min_k = INT64_MAX;
  a.max_k = INT64_MIN;
  return a;
}

static void host_two_sum(double &s, double &c, double x) {
  double t = s + x;
  double bp = t - s;
  c += (s - (t - bp)) + (x - bp);
  s = t;
}

static void scan_acc_merge(ScanAcc &a, const ScanAcc &b) {
  if (b.total == 0) return;
  a.is_float = b.is_float;
  a.total += b.total;
  a.non_null += b.non_null;
  a.min_k = std::min(a.min_k, b.min_k);
  a.max_k = std::max(a.max_k, b.max_k);
  uint64_t lo = a.sum_lo + b.sum_lo;
  a.sum_hi += b.sum_hi + (lo < a.sum_lo ? 1 : 0);
  a.sum_lo = lo;
  double c = a.comp + b.comp;
  host_two_sum(a.sum, c, b.sum);
  a.comp = c;
  if (b.var_n > 0) {
    if (a.var_n == 0) {
      a.var_n = b.var_n;
      a.var_mean = b.var_mean;
      a.var_m2 = b.var_m2;
    } else {
      double na = (double)a.var_n, nb = (double)b.var_n, n = na + nb;
      double delta = b.var_mean - a.var_mean;
      a.var_mean += delta * nb / n;
      a.var_m2 += b.var_m2 + delta * delta * na * nb / n;
      a.var_n += b.var_n;
    }
  }
}

// ---- co-moments: sums about a pivot (kernels/device_types.h, ComomentAcc) -------------------------------------
// Host-side arithmetic on them runs in x87 extended precision (64-bit mantissa): re-basing and centring are a
// handful of operations per state, and a sum travels as the (double, remainder) pair the kernels keep anyway.
typedef long double xdouble;
static xdouble como_sum(const ComomentAcc &a, int k) {
  return std::isfinite(a.s[k]) ? (xdouble)a.s[k] + (xdouble)a.c[k] : (xdouble)a.s[k];
}
static void como_store(ComomentAcc &a, int k, xdouble v) {
  a.s[k] = (double)v;
  a.c[k] = std::isfinite(a.s[k]) ? (double)(v - (xdouble)a.s[k]) : 0.0;
}
// the same sums about (px, py): x - px = (x - b.px) + dx
static void como_rebase(ComomentAcc &b, double px, double py) {
  if (b.px == px && b.py == py) return;
  const xdouble n = (xdouble)b.n, dx = (xdouble)b.px - (xdouble)px, dy = (xdouble)b.py - (xdouble)py;
  const xdouble s0 = como_sum(b, 0), s1 = como_sum(b, 1), s2 = como_sum(b, 2), s3 = como_sum(b, 3), s4 = como_sum(b, 4);
  como_store(b, 0, s0 + n * dx);
  como_store(b, 1, s1 + n * dy);
  como_store(b, 2, s2 + 2 * dx * s0 + n * dx * dx);
  como_store(b, 3, s3 + 2 * dy * s1 + n * dy * dy);
  como_store(b, 4, s4 + dx * s1 + dy * s0 + n * dx * dy);
  b.px = px;
  b.py = py;
}
// AnalyzerState::merge for the pair (TG/analyzers/advanced/correlation.rs:64-101 adds the raw sums): the right-hand
// side is re-based onto the left one's pivots first; an empty left side takes the other's pivots as they are
static void como_acc_merge(ComomentAcc &a, const ComomentAcc &b_in) {
  a.total += b_in.total;
  if (b_in.n == 0) return;
  if (a.n == 0) {
    const int64_t total = a.total;
    a = b_in;
    a.total = total;
    return;
  }
  ComomentAcc b = b_in;
  como_rebase(b, a.px, a.py);
  a.n += b.n;
  for (int k = 0; k < 5; k++) como_store(a, k, como_sum(a, k) + como_sum(b, k));
}

static void state_init_host(tgx_state *st, const tgx_plan *plan) {
  st->plan = plan;
  st->col_types.assign(plan->n_columns_needed, 0);
  st->h_scan.assign(plan->scan.size(), scan_acc_identity());
  st->h_count.assign(plan->count.size(), CountAcc{0, 0});
  ComomentAcc z;
  memset(&z, 0, sizeof(z));
  st->h_como.assign(plan->como.size(), z);
  st->distinct.clear();
  st->distinct.resize(plan->distinct.size());
  st->h_kll.clear();
  st->h_kll.resize(plan->kll.size());
  for (size_t i = 0; i < plan->kll.size(); i++) st->h_kll[i].k = plan->kll[i].k;
  st->h_hll.assign(plan->hll.size(), std::vector<uint8_t>());
  st->hll_mode.assign(plan->hll.size(), 0);
  regex_state_init(st);
  kll_state_init(st);
  spearman_state_init(st);
}

tgx_status tgx::state_init_device(tgx_state *st, tgx_error *err) {
  if (st->device_ready) return TGX_OK;
  TGX_TRY(need_device(err));
  const tgx_plan *plan = st->plan;
  if (!st->stream) {
    HIP_TRY(hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking));
    st->own_stream = true;
  }
  // (zero-fills go through the state's own stream: it is non-blocking, so a null-stream hipMemset could still be
  //  in flight when the first kernel on it starts)
  if (!plan->scan.empty()) {
    std::vector<ScanAcc> init(plan->scan.size(), scan_acc_identity());
    HIP_TRY(st->d_scan_acc.reserve(init.size() * sizeof(ScanAcc)));
    // the identities are kept on the device: tgx_state_reset then restores them with a copy ordered on the state's
    // stream (a synchronous hipMemcpy in reset held the caller until every stream of the device had drained)
    HIP_TRY(st->d_scan_identity.reserve(init.size() * sizeof(ScanAcc)));
    HIP_TRY(hipMemcpy(st->d_scan_identity.p, init.data(), init.size() * sizeof(ScanAcc), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpyAsync(st->d_scan_acc.p, st->d_scan_identity.p, init.size() * sizeof(ScanAcc),
                           hipMemcpyDeviceToDevice, st->stream));
    HIP_TRY(st->d_pivots.reserve(plan->scan.size() * sizeof(double)));
    HIP_TRY(st->d_pivot_set.reserve(plan->scan.size() * sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(st->d_pivots.p, 0, plan->scan.size() * sizeof(double), st->stream));
    HIP_TRY(hipMemsetAsync(st->d_pivot_set.p, 0, plan->scan.size() * sizeof(int32_t), st->stream));
  }
  if (!plan->count.empty()) {
    HIP_TRY(st->d_count_acc.reserve(plan->count.size() * sizeof(CountAcc)));
    HIP_TRY(hipMemsetAsync(st->d_count_acc.p, 0, plan->count.size() * sizeof(CountAcc), st->stream));
  }
  if (!plan->como.empty()) {
    HIP_TRY(st->d_como_acc.reserve(plan->como.size() * sizeof(ComomentAcc)));
    HIP_TRY(hipMemsetAsync(st->d_como_acc.p, 0, plan->como.size() * sizeof(ComomentAcc), st->stream));
  }
  if (!plan->hll.empty()) {
    HIP_TRY(st->d_hll.reserve(plan->hll.size() * (size_t)kHllRegisters));
    HIP_TRY(hipMemsetAsync(st->d_hll.p, 0, plan->hll.size() * (size_t)kHllRegisters, st->stream));
  }
  if (!st->distinct.empty()) {
    const size_t each = kNumDistinctCounters * sizeof(unsigned long long);
    HIP_TRY(st->d_distinct_counters.reserve(st->distinct.size() * each));
    HIP_TRY(hipMemsetAsync(st->d_distinct_counters.p, 0, st->distinct.size() * each, st->stream));
    for (size_t i = 0; i < st->distinct.size(); i++)
      st->distinct[i].counters.borrow((char *)st->d_distinct_counters.p + i * each, each);
  }
  st->device_ready = true;
  return TGX_OK;
}

extern "C" tgx_status tgx_state_create(const tgx_plan *plan, void *hip_stream, tgx_state **out,
                                       tgx_error *err) try {
  if (!plan || !out) return fail(err, TGX_INVALID_ARGUMENT, "plan/out is NULL");
  *out = nullptr;
  tgx_state *st = new tgx_state();
  state_init_host(st, plan);
  st->stream = (hipStream_t)hip_stream;
  st->own_stream = false;
  // TGX_COALESCE=0 (or tgx_options.flags & TGX_OPT_NO_COALESCE): every batch is launched as it arrives
  const char *ce = getenv("TGX_COALESCE");
  st->coalesce.disabled = g_ctx.no_coalesce || (ce && ce[0] == '0');
  if (const char *fe = getenv("TGX_COALESCE_FLUSH_ROWS"))  // (tests: many flushes from little data)
    st->coalesce.flush_rows = std::max<int64_t>(1, atoll(fe));
  *out = st;
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

extern "C" void tgx_state_destroy(tgx_state *st) {
  if (!st) return;
  bind_thread();
  if (st->device_ready && st->stream) (void)hipStreamSynchronize(st->stream);
  for (auto &kv : st->profile)
    for (auto &ev : kv.second.pending) {
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
  regex_state_free(st);
  kll_state_free(st);
  spearman_state_free(st);
  for (int k = 0; k < 2; k++) {
    if (st->arena_event[k]) (void)hipEventDestroy(st->arena_event[k]);
    if (st->arena_host[k]) (void)hipHostFree(st->arena_host[k]);
    tgx::Coalescer &co = st->coalesce;
    if (co.arena_event[k]) (void)hipEventDestroy(co.arena_event[k]);
    if (co.snap_event[k]) (void)hipEventDestroy(co.snap_event[k]);
    if (co.arena_host[k]) (void)hipHostFree(co.arena_host[k]);
    if (co.desc_host[k]) (void)hipHostFree(co.desc_host[k]);
    if (co.snap_host[k]) (void)hipHostFree(co.snap_host[k]);
  }
  if (st->h_pinned) (void)hipHostFree(st->h_pinned);
  if (st->keys_ready) (void)hipEventDestroy(st->keys_ready);
  if (st->aux_done) (void)hipEventDestroy(st->aux_done);
  if (st->aux_stream) (void)hipStreamDestroy(st->aux_stream);
  if (st->own_stream && st->stream) (void)hipStreamDestroy(st->stream);
  delete st;
}

extern "C" tgx_status tgx_state_sync(tgx_state *st, tgx_error *err) try {
  bind_thread();
  if (!st) return fail(err, TGX_INVALID_ARGUMENT, "state is NULL");
  TGX_TRY(coalesce_flush(st, err));  // batches tgx_update has only noted so far
  if (st->device_ready) {
    TGX_TRY(distinct_resolve_all(st, err));  // the caller may release its DEVICE batches after this call
    HIP_TRY(hipStreamSynchronize(st->stream));
  }
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

extern "C" tgx_status tgx_state_reset(const tgx_plan *plan, tgx_state *st, tgx_error *err) try {
  bind_thread();
  if (!plan || !st || st->plan != plan) return fail(err, TGX_INVALID_ARGUMENT, "state does not belong to plan");
  coalesce_drop(st);
  st->keys_ready_recorded = false;
  st->passes = 0;
  if (st->device_ready) HIP_TRY(hipStreamSynchronize(st->stream));
  st->parked.clear();
  st->ptr_tables.clear();
  // host side back to the identity; device buffers are kept and re-zeroed (no hipFree / hipMalloc)
  st->batches = 0;
  st->col_types.assign(plan->n_columns_needed, 0);
  st->h_scan.assign(plan->scan.size(), scan_acc_identity());
  st->h_count.assign(plan->count.size(), CountAcc{0, 0});
  ComomentAcc z;
  memset(&z, 0, sizeof(z));
  st->h_como.assign(plan->como.size(), z);
  for (auto &d : st->distinct) {
    d.mode = DistinctMode::kUndecided;
    d.col_type = 0;
    d.base = 0;
    d.range = 0;
    d.wide = false;
    d.has_hint = false;
    d.speculative = false;
    d.outliers_possible = false;
    d.fp_staged = false;
    d.retained.clear();
    d.bitmap_words = 0;
    d.capacity = 0;  // buffers stay allocated; hash_ensure / the bitmap path clear them before use
    d.rows_upper_bound = 0;
    d.total_rows = 0;
    d.partitioned = false;
    d.h_total = d.h_non_null = d.h_distinct = d.h_twice = d.h_empty_rows = 0;
  }
  for (size_t i = 0; i < st->h_kll.size(); i++) {
    st->h_kll[i] = KllHost();
    st->h_kll[i].k = plan->kll[i].k;
  }
  kll_state_reset(st);
  regex_state_reset(st);
  spearman_state_reset(st);
  st->como_pivot_tries.clear();
  st->h_hll.assign(plan->hll.size(), std::vector<uint8_t>());
  st->hll_mode.assign(plan->hll.size(), 0);
  if (st->device_ready && st->d_hll.p)
    HIP_TRY(hipMemsetAsync(st->d_hll.p, 0, plan->hll.size() * (size_t)kHllRegisters, st->stream));
  if (st->device_ready) {
    if (!plan->scan.empty()) {
      HIP_TRY(hipMemcpyAsync(st->d_scan_acc.p, st->d_scan_identity.p, plan->scan.size() * sizeof(ScanAcc),
                             hipMemcpyDeviceToDevice, st->stream));
      HIP_TRY(hipMemsetAsync(st->d_pivots.p, 0, plan->scan.size() * sizeof(double), st->stream));
      HIP_TRY(hipMemsetAsync(st->d_pivot_set.p, 0, plan->scan.size() * sizeof(int32_t), st->stream));
    }
    if (!plan->count.empty())
      HIP_TRY(hipMemsetAsync(st->d_count_acc.p, 0, plan->count.size() * sizeof(CountAcc), st->stream));
    if (!plan->como.empty())
      HIP_TRY(hipMemsetAsync(st->d_como_acc.p, 0, plan->como.size() * sizeof(ComomentAcc), st->stream));
    if (st->d_distinct_counters.p)
      HIP_TRY(hipMemsetAsync(st->d_distinct_counters.p, 0,
                             st->distinct.size() * kNumDistinctCounters * sizeof(unsigned long long), st->stream));
  }
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

// ------------------------------------------------------------------------------------------------
// profiling
static void prof_begin(tgx_state *st, const char *name, uint64_t bytes, hipEvent_t *e0, hipEvent_t *e1) {
  *e0 = *e1 = nullptr;
  if (!st->profiling) return;
  if (hipEventCreate(e0) != hipSuccess || hipEventCreate(e1) != hipSuccess) {
    *e0 = *e1 = nullptr;
    return;
  }
  (void)hipEventRecord(*e0, st->stream);
  ProfileEntry &pe = st->profile[name];
  pe.pending_bytes.push_back(bytes);
}
static void prof_end(tgx_state *st, const char *name, hipEvent_t e0, hipEvent_t e1) {
  if (!st->profiling || !e0) return;
  (void)hipEventRecord(e1, st->stream);
  st->profile[name].pending.emplace_back(e0, e1);
}
struct ProfScope {
  tgx_state *st;
  const char *name;
  hipEvent_t e0, e1;
  ProfScope(tgx_state *s, const char *n, uint64_t bytes) : st(s), name(n) { prof_begin(s, n, bytes, &e0, &e1); }
  ~ProfScope() { prof_end(st, name, e0, e1); }
};

extern "C" tgx_status tgx_profile_enable(tgx_state *st, int32_t on) try {
  if (!st) return TGX_INVALID_ARGUMENT;
  st->profiling = on != 0;
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(nullptr);
}

static void prof_resolve(tgx_state *st) {
  for (auto &kv : st->profile) {
    ProfileEntry &pe = kv.second;
    for (size_t i = 0; i < pe.pending.size(); i++) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, pe.pending[i].first, pe.pending[i].second) == hipSuccess) {
        pe.total_ms += ms;
        pe.launches += 1;
        pe.bytes += i < pe.pending_bytes.size() ? pe.pending_bytes[i] : 0;
      }
      (void)hipEventDestroy(pe.pending[i].first);
      (void)hipEventDestroy(pe.pending[i].second);
    }
    pe.pending.clear();
    pe.pending_bytes.clear();
  }
}

extern "C" tgx_status tgx_profile_get(tgx_state *st, const char *kernel, double *total_ms,
                                      uint64_t *launches, uint64_t *algorithmic_bytes, tgx_error *err) try {
  bind_thread();
  if (!st || !kernel) return fail(err, TGX_INVALID_ARGUMENT, "state/kernel is NULL");
  if (strcmp(kernel, "coalesce") == 0) {  // flushes so far / batches that were coalesced into them (no timing)
    if (total_ms) *total_ms = 0.0;
    if (launches) *launches = st->coalesce.flushes;
    if (algorithmic_bytes) *algorithmic_bytes = st->coalesce.coalesced_batches;
    return TGX_OK;
  }
  TGX_TRY(coalesce_flush(st, err));
  if (st->device_ready) HIP_TRY(hipStreamSynchronize(st->stream));
  prof_resolve(st);
  auto it = st->profile.find(kernel);
  if (total_ms) *total_ms = it == st->profile.end() ? 0.0 : it->second.total_ms;
  if (launches) *launches = it == st->profile.end() ? 0 : it->second.launches;
  if (algorithmic_bytes) *algorithmic_bytes = it == st->profile.end() ? 0 : it->second.bytes;
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

extern "C" tgx_status tgx_profile_reset(tgx_state *st) try {
  bind_thread();
  if (!st) return TGX_INVALID_ARGUMENT;
  if (st->device_ready) (void)hipStreamSynchronize(st->stream);
  prof_resolve(st);
  st->profile.clear();
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(nullptr);
}

// ------------------------------------------------------------------------------------------------
// update
static bool is_numeric(int t) { return t == TGX_INT64 || t == TGX_FLOAT64; }
static bool is_numeric32(int t) { return t == TGX_INT32 || t == TGX_FLOAT32; }

// copies a HOST column's buffers to the device; `out` is the device view
constexpr size_t kArenaBytes = 8u << 20;        // pinned staging arena per state
constexpr size_t kArenaMaxBuffer = 256u << 10;  // buffers up to this size go through it

static bool is_string(int t) { return t == TGX_UTF8 || t == TGX_LARGE_UTF8; }
static bool is_any_string(int t) { return is_string(t) || t == TGX_UTF8_VIEW; }

// `widen32`: a TGX_INT32 / TGX_FLOAT32 column is needed as 8-byte values (DISTINCT, KLL, co-moments, Spearman); the
// scan alone reads 4-byte values as they are
static tgx_status stage_column(tgx_state *st, const tgx_column &c, tgx_column *out, tgx_error *err,
                               bool widen32 = true) {
  *out = c;
  if (c.type == TGX_DICT32_UTF8) {
    // the dictionary is a column of its own (and may live in a different memory space than the indices)
    st->dict_views.emplace_back();
    tgx_column *dv = &st->dict_views.back();
    if (c.dictionary->length == 0)
      *dv = *c.dictionary;
    else
      TGX_TRY(stage_column(st, *c.dictionary, dv, err));
    out->dictionary = dv;
  }
  if (c.mem != TGX_MEM_HOST && c.mem != TGX_MEM_DEVICE)
    return fail(err, TGX_INVALID_ARGUMENT, "unknown memory space %d", c.mem);
  auto stage = [&](const void *src, size_t bytes, const void **dst) -> tgx_status {
    *dst = nullptr;
    if (!src || bytes == 0) return TGX_OK;
    // HOST columns: tgx_update synchronises the stream before it returns, so the pinned arena is free again
    if (c.mem == TGX_MEM_HOST && bytes <= kArenaMaxBuffer) {
      const int k = st->arena_cur;
      if (!st->arena_host[k]) {
        HIP_TRY(hipHostMalloc(&st->arena_host[k], kArenaBytes, hipHostMallocDefault));
        HIP_TRY(st->arena_dev[k].reserve(kArenaBytes));
        HIP_TRY(hipEventCreateWithFlags(&st->arena_event[k], hipEventDisableTiming));
      }
      if (st->arena_busy[k]) {  // the update that used this arena two turns ago (almost always long done)
        HIP_TRY(hipEventSynchronize(st->arena_event[k]));
        st->arena_busy[k] = false;
      }
      const size_t at = (st->arena_used + 63) & ~(size_t)63;
      if (at + bytes + 16 <= kArenaBytes) {
        memcpy((char *)st->arena_host[k] + at, src, bytes);
        st->arena_used = at + bytes + 16;
        *dst = (const char *)st->arena_dev[k].p + at;
        return TGX_OK;
      }
    }
    if (c.mem == TGX_MEM_HOST) st->host_direct = true;
    if (st->staging_used == st->staging.size()) st->staging.emplace_back(new DevBuf());
    DevBuf *b = st->staging[st->staging_used++].get();
    HIP_TRY(b->reserve(bytes + 16));
    HIP_TRY(hipMemcpyAsync(b->p, src, bytes, hipMemcpyHostToDevice, st->stream));
    *dst = b->p;
    return TGX_OK;
  };
  if (c.type == TGX_UTF8_VIEW) {
    // the kernels read the data buffers through a DEVICE table of their (device) pointers
    st->ptr_tables.emplace_back((size_t)std::max(c.n_variadic, 1), nullptr);
    std::vector<const uint8_t *> &table = st->ptr_tables.back();
    for (int32_t k = 0; k < c.n_variadic; k++) {
      table[k] = c.variadic[k];
      if (c.mem == TGX_MEM_HOST) {
        const void *q = nullptr;
        TGX_TRY(stage(c.variadic[k], (size_t)c.variadic_sizes[k], &q));
        table[k] = (const uint8_t *)q;
      }
    }
    const void *dt = nullptr;
    TGX_TRY(stage(table.data(), table.size() * sizeof(void *), &dt));
    out->variadic = (const uint8_t *const *)dt;
  }
  if (c.mem == TGX_MEM_DEVICE && !(is_numeric32(c.type) && widen32)) return TGX_OK;
  // Only the window the batch views is copied: a sliced array (offset > 0 into big buffers) costs its own rows,
  // not everything before them.  The window starts at slot e0 = offset rounded down to 64 (keeps the validity
  // byte / word alignment the kernels like); the device view gets offset - e0 as its Arrow offset.
  const int64_t e0 = c.offset & ~(int64_t)63;
  const int64_t slots = c.offset - e0 + c.length;  // slots of the window
  const void *p = nullptr;
  if (is_numeric32(c.type)) {
    // 4-byte numerics (include/tgx.h): the window is widened to 8-byte values in a staging buffer on the device; the
    // kernels then see an Int64 / Float64 column.  DEVICE columns keep their validity bitmap and Arrow offset as they
    // are (only the values move: slot e0 of the source becomes slot 0 of the widened buffer, so the bitmap of a device
    // column is re-based by staging nothing and pointing at byte e0 / 8).
    const bool host = c.mem == TGX_MEM_HOST;
    const void *src = c.values ? (const uint8_t *)c.values + (size_t)e0 * 4 : nullptr;
    if (host) {
      TGX_TRY(stage(c.validity ? c.validity + (e0 >> 3) : nullptr, c.validity ? (size_t)((slots + 7) / 8) : 0, &p));
      out->validity = (const uint8_t *)p;
      TGX_TRY(stage(src, (size_t)slots * 4, &p));
      src = p;
    } else {
      out->validity = c.validity ? c.validity + (e0 >> 3) : nullptr;
    }
    out->offset = c.offset - e0;
    if (!widen32) {  // (a HOST column: the scan reads the staged 4-byte window)
      out->values = src;
      out->mem = TGX_MEM_DEVICE;
      return TGX_OK;
    }
    out->type = c.type == TGX_INT32 ? TGX_INT64 : TGX_FLOAT64;
    out->values = nullptr;
    if (src) {
      if (st->staging_used == st->staging.size()) st->staging.emplace_back(new DevBuf());
      DevBuf *w = st->staging[st->staging_used++].get();
      HIP_TRY(w->reserve((size_t)slots * 8 + 16));
      // launched by tgx_update once the pinned arena (small HOST buffers travel in it) has been uploaded
      st->pending_widen.push_back({src, w->p, slots, c.type == TGX_FLOAT32 ? 1 : 0});
      out->values = w->p;
    }
    return TGX_OK;
  }
  out->offset = c.offset - e0;
  TGX_TRY(stage(c.validity ? c.validity + (e0 >> 3) : nullptr, c.validity ? (size_t)((slots + 7) / 8) : 0, &p));
  out->validity = (const uint8_t *)p;
  if (is_numeric(c.type)) {
    TGX_TRY(stage(c.values ? (const uint8_t *)c.values + (size_t)e0 * 8 : nullptr, (size_t)slots * 8, &p));
    out->values = p;
  } else if (c.type == TGX_UTF8 || c.type == TGX_LARGE_UTF8) {
    const size_t ow = c.type == TGX_UTF8 ? 4 : 8;
    TGX_TRY(stage(c.offsets ? (const uint8_t *)c.offsets + (size_t)e0 * ow : nullptr, (size_t)(slots + 1) * ow, &p));
    out->offsets = p;
    int64_t first = 0, end = 0;
    if (c.offsets) {
      first = ow == 4 ? (int64_t)((const int32_t *)c.offsets)[c.offset] : ((const int64_t *)c.offsets)[c.offset];
      end = ow == 4 ? (int64_t)((const int32_t *)c.offsets)[c.offset + c.length]
                    : ((const int64_t *)c.offsets)[c.offset + c.length];
    }
    // value bytes [first, end) only; the device pointer is rebased so that the original offsets still index it
    // (16 bytes of slack in front: the pattern kernel stages 16-byte blocks by absolute address)
    const int64_t lead = first & 15;
    TGX_TRY(stage(c.data ? c.data + (first - lead) : nullptr, (size_t)(end - first + lead), &p));
    out->data = p ? (const uint8_t *)p - (first - lead) : nullptr;
  } else if (c.type == TGX_DICT32_UTF8) {
    TGX_TRY(stage(c.values ? (const uint8_t *)c.values + (size_t)e0 * 4 : nullptr, (size_t)slots * 4, &p));
    out->values = p;
  } else if (c.type == TGX_UTF8_VIEW) {
    TGX_TRY(stage(c.values ? (const uint8_t *)c.values + (size_t)e0 * 16 : nullptr, (size_t)slots * 16, &p));
    out->values = p;
  } else {
    return fail(err, TGX_UNSUPPORTED, "column type %d is not supported", c.type);
  }
  out->mem = TGX_MEM_DEVICE;
  return TGX_OK;
}

static void fill_scan_desc(const tgx_column &c, bool variance, const double *pivot, ScanColDesc *d) {
  d->values = c.values;
  d->validity = c.validity;
  d->offset = c.offset;
  d->length = c.length;
  d->is_float = c.type == TGX_FLOAT64 || c.type == TGX_FLOAT32;
  d->want_variance = variance ? 1 : 0;
  d->pivot = pivot;
  d->elem32 = is_numeric32(c.type) ? 1 : 0;
  d->skip_stats = 0;
  d->hll = nullptr;
  d->hll_regs = nullptr;
  const uintptr_t width = d->elem32 ? 4 : 8;  // a lane's pair of rows is one 2 x width load
  int64_t head = (64 - (c.offset & 63)) & 63;
  if (head > c.length) head = c.length;
  int64_t n_tiles = (c.length - head) / kTileRows;
  const uintptr_t vaddr = (uintptr_t)c.values + (uintptr_t)(c.offset + head) * width;
  const uintptr_t baddr = (uintptr_t)c.validity + (uintptr_t)((c.offset + head) >> 3);
  if ((vaddr & (2 * width - 1)) != 0 || (c.validity && (baddr & 7) != 0)) n_tiles = 0;  // per-lane path
  d->head = n_tiles > 0 ? head : 0;
  d->n_tiles = n_tiles;
}

static int scan_blocks_for(const ScanColDesc &d, int n_cols_in_launch, int per_cu = 8) {
  int64_t want;
  if (d.n_tiles > 0)
    want = (d.n_tiles + 4 * kWavesPerBlock - 1) / (4 * kWavesPerBlock);  // >= 4 tiles per wave
  else
    want = (d.length + kScanBlock * 8 - 1) / (kScanBlock * 8);
  // (4 .. 12 workgroups per CU all measured 20.7-22.5 ms on the 1 G x 16 scan: HBM-bound, not occupancy-bound)
  int cap = std::max(32, (g_ctx.n_cu * per_cu) / std::max(1, n_cols_in_launch));
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (int)want;
}

static uint64_t next_pow2(uint64_t x) {
  uint64_t p = 1;
  while (p < x) p <<= 1;
  return p;
}

static tgx_status distinct_read_counters(tgx_state *st, DistinctState &ds, unsigned long long *out,
                                         tgx_error *err) {
  memset(out, 0, kNumDistinctCounters * sizeof(unsigned long long));
  if (!ds.counters.p) return TGX_OK;
  HIP_TRY(hipMemcpyAsync(out, ds.counters.p, kNumDistinctCounters * sizeof(unsigned long long),
                         hipMemcpyDeviceToHost, st->stream));
  HIP_TRY(hipStreamSynchronize(st->stream));
  return TGX_OK;
}

static HashSetView hash_view(const DistinctState &ds) {
  HashSetView v;
  v.keys = ds.keys.as<uint64_t>();
  v.dup = ds.dup.as<uint32_t>();
  v.mask = ds.capacity - 1;
  return v;
}
static BitmapView bitmap_view(const DistinctState &ds) {
  BitmapView v;
  v.seen = ds.seen.as<uint32_t>();
  v.twice = ds.twice.as<uint32_t>();
  v.base = ds.base;
  v.range = ds.range;
  return v;
}

// allocate an empty table of `capacity` slots into (keys, dup)
static tgx_status hash_alloc(tgx_state *st, DevBuf &keys, DevBuf &dup, uint64_t capacity, bool mult,
                             bool wide, tgx_error *err) {
  const size_t slot_bytes = wide ? 16 : 8;
  HIP_TRY(keys.reserve(capacity * slot_bytes));
  HIP_TRY(hipMemsetAsync(keys.p, 0xFF, capacity * slot_bytes, st->stream));
  if (mult) {
    HIP_TRY(dup.reserve((capacity / 32 + 1) * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(dup.p, 0, (capacity / 32 + 1) * sizeof(uint32_t), st->stream));
  }
  return TGX_OK;
}

// make sure the hash table can take `incoming` more keys at load factor <= 0.5
static tgx_status hash_ensure(tgx_state *st, DistinctState &ds, bool mult, uint64_t incoming,
                              tgx_error *err) {
  if (ds.capacity == 0) {
    uint64_t want = std::max<uint64_t>(incoming, g_ctx.distinct_hint);
    ds.capacity = next_pow2(std::max<uint64_t>(2 * want, 1024));
    TGX_TRY(hash_alloc(st, ds.keys, ds.dup, ds.capacity, mult, ds.wide, err));
    ds.rows_upper_bound = 0;
  }
  if (2 * (ds.rows_upper_bound + incoming) <= ds.capacity) {
    ds.rows_upper_bound += incoming;
    return TGX_OK;
  }
  // the bound says it might not fit: read the real key count
  unsigned long long c[kNumDistinctCounters];
  TGX_TRY(distinct_read_counters(st, ds, c, err));
  uint64_t actual = c[kCntDistinct];
  if (2 * (actual + incoming) <= ds.capacity) {
    ds.rows_upper_bound = actual + incoming;
    return TGX_OK;
  }
  uint64_t new_cap = next_pow2(2 * (actual + incoming));
  DevBuf nk, nd;
  TGX_TRY(hash_alloc(st, nk, nd, new_cap, mult, ds.wide, err));
  HashSetView src = hash_view(ds);
  HashSetView dst{nk.as<uint64_t>(), nd.as<uint32_t>(), new_cap - 1};
  // re-insertion recounts distinct / twice: zero those two counters first
  HIP_TRY(hipMemsetAsync(ds.counters.p, 0, 2 * sizeof(unsigned long long), st->stream));
  if (ds.wide)
    launch_hash_rehash128(src, dst, mult ? 1 : 0, ds.counters.as<unsigned long long>(), st->stream);
  else
    launch_hash_rehash(src, dst, mult ? 1 : 0, ds.counters.as<unsigned long long>(), st->stream);
  HIP_TRY(hipStreamSynchronize(st->stream));
  std::swap(ds.keys.p, nk.p);
  std::swap(ds.keys.cap, nk.cap);
  std::swap(ds.dup.p, nd.p);
  std::swap(ds.dup.cap, nd.cap);
  ds.capacity = new_cap;
  ds.rows_upper_bound = actual + incoming;
  return TGX_OK;
}

static tgx_status bitmap_to_hash(tgx_state *st, DistinctState &ds, bool mult, uint64_t incoming,
                                 tgx_error *err) {
  unsigned long long c[kNumDistinctCounters];
  TGX_TRY(distinct_read_counters(st, ds, c, err));
  uint64_t actual = c[kCntDistinct];
  ds.capacity = 0;
  ds.rows_upper_bound = 0;
  uint64_t want = std::max<uint64_t>(actual + incoming, g_ctx.distinct_hint);
  ds.capacity = next_pow2(std::max<uint64_t>(2 * want, 1024));
  TGX_TRY(hash_alloc(st, ds.keys, ds.dup, ds.capacity, mult, ds.wide, err));
  HIP_TRY(hipMemsetAsync(ds.counters.p, 0, 2 * sizeof(unsigned long long), st->stream));
  launch_bitmap_to_hash(bitmap_view(ds), hash_view(ds), mult ? 1 : 0, ds.counters.as<unsigned long long>(),
                        st->stream);
  HIP_TRY(hipStreamSynchronize(st->stream));
  // (the bitmap's buffers stay with the state: a step that is reset and repeated would free and allocate them each
  //  time, and hipFree waits for the whole device)
  ds.mode = DistinctMode::kHash;
  ds.rows_upper_bound = actual;
  return TGX_OK;
}

// COUNT(DISTINCT (a, b, ...)): every row's tuple goes into the 128-bit fingerprint set (kernels/distinct128.hip)
// the kernels' view of a tuple of columns; cols[k] = the k-th component
static tgx_status tuple_desc_of(const std::vector<const tgx_column *> &cols, bool mult, TupleDesc *d, tgx_error *err) {
  memset(d, 0, sizeof(*d));
  d->n_cols = (int32_t)cols.size();
  d->want_multiplicity = mult ? 1 : 0;
  d->length = cols[0]->length;
  for (size_t k = 0; k < cols.size(); k++) {
    const tgx_column &c = *cols[k];
    TupleCol &tc = d->cols[k];
    tc.validity = c.validity;
    tc.offset = c.offset;
    if (is_numeric(c.type)) {
      tc.kind = 0;
      tc.values = c.values;
    } else if (c.type == TGX_UTF8 || c.type == TGX_LARGE_UTF8) {
      tc.kind = c.type == TGX_UTF8 ? 1 : 2;
      tc.offsets = c.offsets;
      tc.data = c.data;
    } else if (c.type == TGX_UTF8_VIEW) {
      tc.kind = 3;
      tc.values = c.values;
      tc.buffers = c.variadic;
    } else {
      return fail(err, TGX_UNSUPPORTED, "DISTINCT over a tuple: column type %d is not supported", c.type);
    }
  }
  return TGX_OK;
}

static bool fp_lists_fit_rows(int64_t rows);
static tgx_status fp_lists_tuple_update(tgx_state *st, size_t slot, const TupleDesc &d,
                                        const std::vector<const tgx_column *> &cols, tgx_error *err);

// COUNT(DISTINCT (a, b, ...)): every row's tuple goes into the 128-bit fingerprint set (kernels/distinct128.hip)
static tgx_status distinct_tuple_update(tgx_state *st, size_t slot, const tgx_column *dev, tgx_error *err,
                                        const tgx_column *orig = nullptr) {
  const DistinctTask &task = st->plan->distinct[slot];
  DistinctState &ds = st->distinct[slot];
  std::vector<const tgx_column *> cols;
  bool any_view = false;  // a component whose device view lives in per-update scratch: never retained (no lists)
  for (int c2 : task.tuple) {
    cols.push_back(&dev[c2]);
    any_view |= dev[c2].type == TGX_UTF8_VIEW || (orig && is_numeric32(orig[c2].type));
  }
  TupleDesc d;
  TGX_TRY(tuple_desc_of(cols, task.multiplicity, &d, err));
  ds.col_type = TGX_UTF8;  // a 128-bit fingerprint set, like a string column's
  ds.total_rows += d.length;
  if (d.length == 0) return TGX_OK;
  if (ds.fp_staged) TGX_TRY(distinct_resolve(st, slot, err));  // a second batch: the table takes over
  // the first big batch: through the partitioned lists (views read their buffers through a table staged per update)
  if (ds.mode == DistinctMode::kUndecided && !any_view && fp_lists_fit_rows(d.length))
    return fp_lists_tuple_update(st, slot, d, cols, err);
  if (ds.mode == DistinctMode::kUndecided) {
    ds.mode = DistinctMode::kHash;
    ds.wide = true;
  }
  TGX_TRY(hash_ensure(st, ds, task.multiplicity, (uint64_t)d.length, err));
  ProfScope ps(st, "distinct", 0);
  launch_distinct_tuple(d, hash_view(ds), ds.counters.as<unsigned long long>(), st->stream);
  return TGX_OK;
}

namespace {
struct NumericPrep {
  bool prepared = false;
  bool partitioned = false;  // the batch goes through partition_kernel / bucket_apply_kernel
  uint32_t sub_bits = 0;
  bool key16 = false;
  uint64_t n_buckets = 0;
};
}  // namespace
static tgx_status distinct_prepare_numeric(tgx_state *st, size_t slot, const tgx_column &c, NumericPrep *prep,
                                           tgx_error *err);
static tgx_status distinct_run_numeric(tgx_state *st, size_t slot, const tgx_column &c, const NumericPrep &prep,
                                       int stats_slot, tgx_error *err, const tgx_column *orig = nullptr);

// ---- big Utf8 batches: partitioned fingerprint lists (kernels/distinct128.hip, fp_*) ----
// records a list is sized for when `rows` values are spread over `lists` lists: the mean, twelve standard deviations
// (values that repeat widen the spread) and a floor
static uint64_t fp_list_cap(int64_t rows, uint64_t lists) {
  const double mean = (double)rows / (double)lists;
  return ((uint64_t)(mean + 12.0 * std::sqrt(mean) + 64.0) + 15) & ~15ull;
}
static bool fp_lists_fit_rows(int64_t rows) {
  // TGX_FP_LISTS_MIN_ROWS: smallest batch that takes this path (tests lower it; a huge value turns the path off)
  int64_t min_rows = kFpMinRows;
  if (const char *e = getenv("TGX_FP_LISTS_MIN_ROWS")) min_rows = std::max<int64_t>(1, atoll(e));
  return rows >= min_rows && fp_list_cap(rows, (uint64_t)kFpFan * kFpFan) <= kFpListMax;
}
static bool fp_lists_fit(const tgx_column &c) {
  return (is_any_string(c.type) || c.type == TGX_INT64 || c.type == TGX_FLOAT64) && fp_lists_fit_rows(c.length);
}
static void fp_views(const DistinctState &ds, FpLists *l1, FpLists *l2) {
  l1->recs = ds.fp_level1.as<uint64_t>();
  l1->offered = ds.fp_offered.as<uint32_t>();
  l1->cap = ds.fp_cap1;
  l2->recs = ds.fp_level2.as<uint64_t>();
  l2->offered = ds.fp_offered.as<uint32_t>() + kFpXcds * kFpFan;
  l2->cap = ds.fp_cap2;
}
// sizes and clears the two levels of lists for a batch of `rows` records of `rec_bytes` bytes
static tgx_status fp_lists_prepare(tgx_state *st, DistinctState &ds, int64_t rows, size_t rec_bytes, tgx_error *err) {
  constexpr uint64_t kLists2 = (uint64_t)kFpFan * kFpFan;
  constexpr uint64_t kLists1 = (uint64_t)kFpXcds * kFpFan;
  ds.fp_cap1 = fp_list_cap(rows, kLists1);
  ds.fp_cap2 = fp_list_cap(rows, kLists2);
  HIP_TRY(ds.fp_level1.reserve(kLists1 * ds.fp_cap1 * rec_bytes));
  HIP_TRY(ds.fp_level2.reserve(kLists2 * ds.fp_cap2 * rec_bytes));
  HIP_TRY(ds.fp_offered.reserve((kLists1 + kLists2) * sizeof(uint32_t)));
  HIP_TRY(ds.fp_per_list.reserve(kLists2 * sizeof(uint2)));
  HIP_TRY(hipMemsetAsync(ds.fp_offered.p, 0, (kLists1 + kLists2) * sizeof(uint32_t), st->stream));
  return TGX_OK;
}

static tgx_status fp_lists_update(tgx_state *st, size_t slot, const tgx_column &c, tgx_error *err) {
  DistinctState &ds = st->distinct[slot];
  const bool mult = st->plan->distinct[slot].multiplicity;
  TGX_TRY(fp_lists_prepare(st, ds, c.length, 16, err));
  FpLists l1, l2;
  fp_views(ds, &l1, &l2);
  ProfScope ps(st, "distinct", 0), ps_lists(st, "distinct_lists", 0);
  unsigned long long *counters = ds.counters.as<unsigned long long>();
  tgx_column kept = c;
  if (c.type == TGX_UTF8_VIEW) {
    // the table of data-buffer pointers the kernels read through is staged per update: the retained view gets a copy
    const size_t bytes = (size_t)std::max(c.n_variadic, 1) * sizeof(void *);
    HIP_TRY(ds.fp_buffers.reserve(bytes));
    if (c.n_variadic > 0)
      HIP_TRY(hipMemcpyAsync(ds.fp_buffers.p, c.variadic, (size_t)c.n_variadic * sizeof(void *), hipMemcpyDeviceToDevice,
                             st->stream));
    kept.variadic = (const uint8_t *const *)ds.fp_buffers.p;
    launch_fp_partition_views(c.values, kept.variadic, c.validity, c.offset, c.length, l1, counters, st->stream);
  } else {
    launch_fp_partition_strings(c.offsets, c.data, c.validity, c.offset, c.length, c.type == TGX_LARGE_UTF8, l1,
                                counters, st->stream);
  }
  launch_fp_partition_lists(l1, l2, counters, st->stream);
  launch_fp_count(l2, mult ? 1 : 0, ds.fp_per_list.as<uint2>(), l1.offered, counters, st->stream);
  ds.mode = DistinctMode::kHash;
  ds.wide = true;
  ds.capacity = 0;  // no table yet
  ds.rows_upper_bound = 0;
  ds.fp_staged = true;
  ds.retained.push_back(kept);  // (a DEVICE view, or a staged one looked at before the update returns)
  return TGX_OK;
}

// the same for the first big batch of a tuple task: its components are retained in tuple order
static tgx_status fp_lists_tuple_update(tgx_state *st, size_t slot, const TupleDesc &d,
                                        const std::vector<const tgx_column *> &cols, tgx_error *err) {
  DistinctState &ds = st->distinct[slot];
  const bool mult = st->plan->distinct[slot].multiplicity;
  TGX_TRY(fp_lists_prepare(st, ds, d.length, 16, err));
  FpLists l1, l2;
  fp_views(ds, &l1, &l2);
  ProfScope ps(st, "distinct", 0), ps_lists(st, "distinct_lists", 0);
  unsigned long long *counters = ds.counters.as<unsigned long long>();
  launch_fp_partition_tuples(d, l1, counters, st->stream);
  launch_fp_partition_lists(l1, l2, counters, st->stream);
  launch_fp_count(l2, mult ? 1 : 0, ds.fp_per_list.as<uint2>(), nullptr, counters, st->stream);  // (valid rows: level 1)
  ds.mode = DistinctMode::kHash;
  ds.wide = true;
  ds.capacity = 0;  // no table yet
  ds.rows_upper_bound = 0;
  ds.fp_staged = true;
  for (const tgx_column *c : cols) ds.retained.push_back(*c);
  return TGX_OK;
}

// `orig`: the caller's own view of the column when `c` is a per-update staging copy of a DEVICE column (a 4-byte
// numeric column widened for this pass): what a key set retains for a later repair must outlive the update
static tgx_status distinct_update(tgx_state *st, size_t slot, const tgx_column &c, tgx_error *err,
                                  const std::vector<DictGather> *gathers = nullptr, const NumericPrep *ready = nullptr,
                                  int stats_slot = -1, const tgx_column *orig = nullptr) {
  const DistinctTask &task = st->plan->distinct[slot];
  DistinctState &ds = st->distinct[slot];
  const bool mult = task.multiplicity;
  if (is_any_string(c.type)) {
    // values are reduced to 128-bit fingerprints on the fly (kernels/distinct128.hip)
    ds.col_type = c.type;
    ds.total_rows += c.length;
    if (c.length == 0) return TGX_OK;
    if (ds.fp_staged) TGX_TRY(distinct_resolve(st, slot, err));  // a second batch: the table takes over
    if (ds.mode == DistinctMode::kUndecided && fp_lists_fit(c)) return fp_lists_update(st, slot, c, err);
    if (ds.mode == DistinctMode::kUndecided) {
      ds.mode = DistinctMode::kHash;
      ds.wide = true;
    }
    TGX_TRY(hash_ensure(st, ds, mult, (uint64_t)c.length, err));
    ProfScope ps(st, "distinct", 0);
    const bool view = c.type == TGX_UTF8_VIEW;
    launch_distinct_utf8(c.offsets, c.data, view ? c.values : nullptr, view ? c.variadic : nullptr, c.validity,
                         c.offset, c.length, c.type == TGX_LARGE_UTF8, mult ? 1 : 0, hash_view(ds),
                         ds.counters.as<unsigned long long>(), st->stream);
    return TGX_OK;
  }
  if (c.type == TGX_DICT32_UTF8) {
    // string work once per dictionary entry: count references per entry, then insert the fingerprints of the
    // referenced entries -- identical set contents to the plain Utf8 path (kernels/dict.hip)
    const tgx_column &dict = *c.dictionary;
    ds.col_type = c.type;
    ds.total_rows += c.length;
    if (c.length == 0 || dict.length == 0) return TGX_OK;
    if (ds.mode == DistinctMode::kUndecided) {
      ds.mode = DistinctMode::kHash;
      ds.wide = true;
    }
    TGX_TRY(hash_ensure(st, ds, mult, (uint64_t)std::min<int64_t>(c.length, dict.length), err));
    const size_t uw = dict_usage_words(dict.length);
    HIP_TRY(ds.dict_usage.reserve(2 * uw * 4 + 16));
    const size_t scratch = dict_usage_scratch_bytes(c.length, dict.length, mult ? 1 : 0, g_ctx.n_cu);
    if (scratch)
      HIP_TRY(ds.dict_scratch.reserve(scratch));
    else
      HIP_TRY(hipMemsetAsync(ds.dict_usage.p, 0, 2 * uw * 4, st->stream));  // the global-atomics path accumulates
    uint32_t *u_seen = ds.dict_usage.as<uint32_t>(), *u_twice = u_seen + uw;
    ProfScope ps(st, "distinct", 0);
    if (gathers && !gathers->empty() && scratch) {
      // the column's pattern / length checks ride on this pass: the indices are read once
      const uint8_t *hits[4];
      unsigned long long *pc[4];
      int32_t niv[4];
      const int k = (int)std::min<size_t>(gathers->size(), 4);
      for (int i = 0; i < k; i++) {
        hits[i] = (*gathers)[i].hits;
        pc[i] = (*gathers)[i].counters;
        niv[i] = (*gathers)[i].null_is_valid;
      }
      launch_dict_usage_fused((const int32_t *)c.values, c.validity, c.offset, c.length, dict.length, mult ? 1 : 0, k,
                              hits, pc, niv, u_seen, u_twice, ds.dict_scratch.as<uint32_t>(),
                              ds.counters.as<unsigned long long>(), g_ctx.n_cu, st->stream);
    } else {
      launch_dict_usage((const int32_t *)c.values, c.validity, c.offset, c.length, dict.validity, dict.offset,
                        dict.length, mult ? 1 : 0, u_seen, u_twice, ds.dict_scratch.as<uint32_t>(),
                        ds.counters.as<unsigned long long>(), g_ctx.n_cu, st->stream);
    }
    launch_dict_insert(dict.offsets, dict.data, dict.validity, dict.offset, dict.length,
                       dict.type == TGX_LARGE_UTF8, mult ? 1 : 0, u_seen, u_twice, hash_view(ds),
                       ds.counters.as<unsigned long long>(), st->stream);
    return TGX_OK;
  }
  if (!is_numeric(c.type))
    return fail(err, TGX_UNSUPPORTED, "DISTINCT on column type %d is not supported yet", c.type);
  ds.col_type = c.type;
  ds.total_rows += c.length;
  if (c.length == 0) return TGX_OK;
  NumericPrep prep;
  if (ready && ready->prepared)
    prep = *ready;  // decided before the scan of this batch was queued (tgx_update)
  else
    TGX_TRY(distinct_prepare_numeric(st, slot, c, &prep, err));
  return distinct_run_numeric(st, slot, c, prep, stats_slot, err, orig);
}

// How the keys of one batch of an Int64 / Float64 column enter the set.
//   * a set that is a range bitmap takes every batch as it is: keys outside its range are counted, never inserted,
//     and repaired when the host next looks at the state (distinct_resolve) -- no batch waits for its own MIN / MAX;
//   * an undecided Int64 set first SAMPLES the batch (<= 2^16 values, evenly spread; the whole batch when it is
//     smaller): dense value range -> bitmap over the sampled range plus slack, else hash set.  A declared range
//     (tgx_distinct_range_hint) replaces the sample, and then keys outside it are an error, not repaired.
static void bitmap_shape(const DistinctState &ds, int64_t length, bool mult, uint32_t *sub_bits_out, bool *key16_out,
                         uint64_t *n_buckets_out, bool *partitioned_out) {
  // slices of 2^sub_bits keys: as many buckets as fit the phase-1 histogram (<= 2048 targeted) so every CU has lists
  // to replay (<= 1024 targeted: longer runs per tile); one slice (two with multiplicity) must fit 128 KiB of LDS
  uint32_t sub_bits = 14;
  while (sub_bits < (mult ? 19u : 20u) && ((ds.range + (1ull << sub_bits) - 1) >> sub_bits) > 1024) sub_bits++;
  // ranges up to 2048 x 2^16 values (134 M): buckets of <= 2^16 keys make a list entry 2 bytes instead of 4 --
  // half the list traffic for more, shorter runs (not with multiplicity: run padding repeats keys)
  bool key16 = false;
  if (!mult && sub_bits > 16) {
    uint32_t s16 = 14;
    while (s16 < 16 && ((ds.range + (1ull << s16) - 1) >> s16) > kMaxPartitions) s16++;
    if (((ds.range + (1ull << s16) - 1) >> s16) <= kMaxPartitions) {
      key16 = true;
      sub_bits = s16;
    }
  } else if (!mult) {
    key16 = true;  // sub_bits <= 16 already
  }
  const uint64_t n_buckets = (ds.range + (1ull << sub_bits) - 1) >> sub_bits;
  uint64_t cap_slots = (uint64_t)length / std::max<uint64_t>(n_buckets, 1);
  cap_slots = cap_slots + cap_slots / 4 + 16 * (((uint64_t)length >> 15) + 1) + 4096;
  *sub_bits_out = sub_bits;
  *key16_out = key16;
  *n_buckets_out = n_buckets;
  // (TGX_PARTITION_MIN_ROWS: the differential tester sends small batches through the partitioned pass as well)
  const char *min_env = getenv("TGX_PARTITION_MIN_ROWS");
  const int64_t min_rows = min_env ? std::max<int64_t>(1, atoll(min_env)) : (int64_t)1 << 20;
  *partitioned_out = length >= min_rows && n_buckets <= kMaxPartitions && (uint64_t)length * 64 >= ds.range &&
                     cap_slots < (1ull << 32) - 64;
}

static tgx_status pinned_readback(tgx_state *st, size_t bytes, tgx_error *err) {
  if (bytes <= st->h_pinned_cap) return TGX_OK;
  if (st->h_pinned) (void)hipHostFree(st->h_pinned);
  st->h_pinned = nullptr;
  st->h_pinned_cap = 0;
  const size_t want = std::max<size_t>(bytes + bytes / 2, 4096);
  HIP_TRY(hipHostMalloc(&st->h_pinned, want, hipHostMallocDefault));
  st->h_pinned_cap = want;
  return TGX_OK;
}

// does this batch of an undecided Int64 key set get its range from a sample? (see distinct_prepare_numeric)
static bool distinct_wants_sample(const DistinctState &ds, const tgx_column &c) {
  return ds.mode == DistinctMode::kUndecided && c.type == TGX_INT64 && !ds.has_hint && !ds.batch_range_known &&
         c.length >= (1 << 16);
}
// ... or the exact MIN / MAX of a coalesced flush whose key windows were DEVICE memory?  While the key set is undecided,
// or a bitmap no batch can have left outliers under: the flush then lays the bitmap out / grows it like a HOST flush
// (a stream of DEVICE batches of growing ids stays on the bitmap instead of going through the repair, flush after flush)
static bool distinct_wants_exact_range(const DistinctState &ds, const tgx_column &c) {
  if (!ds.flush_device_keys || c.type != TGX_INT64 || ds.has_hint || ds.batch_range_known || c.length < (1 << 16))
    return false;
  return ds.mode == DistinctMode::kUndecided ||
         (ds.mode == DistinctMode::kBitmap && ds.speculative && !ds.partitioned && !ds.outliers_possible);
}

// The samples of ALL key columns of the batch, queued together and read back with ONE wait: a read-back costs the
// stream's latency (~50 us) whatever its size -- two key columns sampled one after the other were 6 % of a
// 100 M-row step.
static tgx_status distinct_sample_all(tgx_state *st, const tgx_column *dev, tgx_error *err) {
  const tgx_plan *plan = st->plan;
  std::vector<size_t> who;
  for (size_t q = 0; q < plan->distinct.size(); q++) {
    const DistinctTask &t = plan->distinct[q];
    st->distinct[q].sample_ready = false;
    if (!t.tuple.empty() || !is_numeric(dev[t.column].type) || dev[t.column].length == 0) continue;
    bool lane = false;  // (the HyperLogLog lane has the column: its approx_only key set stays idle)
    for (size_t h = 0; h < plan->hll.size(); h++)
      lane |= plan->hll[h].distinct_slot == (int)q && st->hll_mode[h] == 1;
    if (lane) continue;
    if (distinct_wants_sample(st->distinct[q], dev[t.column]) || distinct_wants_exact_range(st->distinct[q], dev[t.column]))
      who.push_back(q);
  }
  if (who.empty()) return TGX_OK;
  TGX_TRY(pinned_readback(st, who.size() * sizeof(DistinctSample), err));
  DistinctSample *got = (DistinctSample *)st->h_pinned;
  for (size_t k = 0; k < who.size(); k++) {
    DistinctState &ds = st->distinct[who[k]];
    const tgx_column &c = dev[plan->distinct[who[k]].column];
    DistinctColDesc d;
    d.values = c.values;
    d.validity = c.validity;
    d.offset = c.offset;
    d.length = c.length;
    d.want_multiplicity = 0;
    d.pad = distinct_wants_exact_range(ds, c) ? 1 : 0;  // every row, not a sample
    HIP_TRY(ds.sample.reserve(sizeof(DistinctSample)));
    launch_distinct_init(ds.sample.as<DistinctSample>(), nullptr, st->stream);
    launch_distinct_sample(d, ds.sample.as<DistinctSample>(), st->stream);
    HIP_TRY(hipMemcpyAsync(&got[k], ds.sample.p, sizeof(DistinctSample), hipMemcpyDeviceToHost, st->stream));
  }
  HIP_TRY(hipStreamSynchronize(st->stream));  // (the stream holds nothing but the samples when a step starts)
  for (size_t k = 0; k < who.size(); k++) {
    DistinctState &ds = st->distinct[who[k]];
    if (distinct_wants_exact_range(ds, dev[plan->distinct[who[k]].column])) {
      if (got[k].count) {  // the flush's range, as if the host had seen the values
        ds.batch_range_known = true;
        ds.batch_lo = got[k].min_v;
        ds.batch_hi = got[k].max_v;
      }
      continue;
    }
    ds.sample_host = got[k];
    ds.sample_ready = true;
  }
  return TGX_OK;
}

// Extends a sampled-range bitmap so that it covers [lo, hi] as well: whole 2^20-bit slices are added below and / or
// above (the old words move by whole slices, a device copy), generously in the direction of growth -- at least the
// old range again -- so that a key column that keeps growing costs O(log) extensions.  Only while the range stays as
// dense as a bitmap must be (16 bits per row seen, below 2^34 values); otherwise the keys stay outliers for the repair.
static tgx_status bitmap_grow(tgx_state *st, DistinctState &ds, bool mult, int64_t lo, int64_t hi, int64_t incoming,
                              tgx_error *err) {
  const uint64_t old_top = (uint64_t)ds.base + (ds.range - 1);  // (as unsigned offsets from INT64_MIN they are ordered)
  auto u = [](int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; };
  const bool below = u(lo) < u(ds.base), above = u(hi) > u((int64_t)old_top);
  if (!below && !above) return TGX_OK;
  constexpr uint64_t kSlice = 1ull << 20;
  uint64_t add_below = 0, add_above = 0;
  if (below) {
    const uint64_t need = u(ds.base) - u(lo);
    add_below = (std::max(need, ds.range) + kSlice - 1) / kSlice * kSlice;
    if (add_below > u(ds.base)) add_below = u(ds.base) / kSlice * kSlice;  // (not below INT64_MIN)
    if (add_below < need) return TGX_OK;
  }
  if (above) {
    const uint64_t need = u(hi) - u((int64_t)old_top);
    add_above = std::max(need, ds.range);
    const uint64_t room = 0xFFFFFFFFFFFFFFFFull - u((int64_t)old_top);
    if (add_above > room) add_above = room;
    if (add_above < need) return TGX_OK;
  }
  // (the three terms can add up to exactly 2^64 -- a flush from INT64_MIN to INT64_MAX -- and wrap to a "range" of 0
  //  that passes every density test: sum them with the carry)
  auto sum3 = [](uint64_t a, uint64_t b, uint64_t c, uint64_t *out) {
    uint64_t t = 0;
    return !__builtin_add_overflow(a, b, &t) && !__builtin_add_overflow(t, c, out);
  };
  uint64_t new_range = 0;
  const bool fits = sum3(ds.range, add_below, add_above, &new_range);
  const uint64_t rows_seen = (uint64_t)std::max<int64_t>(ds.total_rows + incoming, 1);
  if (!fits || new_range >= (1ull << 34) || new_range / 16 > std::max<uint64_t>(rows_seen, g_ctx.distinct_hint)) {
    // too sparse for a bitmap once extended that far: take what the batch needs and no more, if that is dense enough
    add_below = below ? ((u(ds.base) - u(lo)) + kSlice - 1) / kSlice * kSlice : 0;
    add_above = above ? u(hi) - u((int64_t)old_top) : 0;
    uint64_t tight = 0;
    if (!sum3(ds.range, add_below, add_above, &tight)) return TGX_OK;
    if (tight >= (1ull << 34) || tight / 16 > std::max<uint64_t>(rows_seen, g_ctx.distinct_hint)) return TGX_OK;
  }
  const uint64_t range = ds.range + add_below + add_above;
  const size_t old_words = (size_t)ds.bitmap_words;
  const size_t words = (size_t)(((range + kSlice - 1) >> 20) << 15) + 4;
  const size_t shift_words = (size_t)(add_below >> 5);
  auto regrow = [&](DevBuf &buf) -> tgx_status {
    DevBuf bigger;
    HIP_TRY(bigger.reserve(words * 4));
    HIP_TRY(hipMemsetAsync(bigger.p, 0, words * 4, st->stream));
    HIP_TRY(hipMemcpyAsync((uint32_t *)bigger.p + shift_words, buf.p, old_words * 4, hipMemcpyDeviceToDevice, st->stream));
    // the old words are still being copied: the old buffer is parked until the stream is next drained (freeing it
    // here would mean waiting for the flush's upload, and hipFree waits for the whole device)
    st->parked.emplace_back(std::move(buf));
    buf = std::move(bigger);
    return TGX_OK;
  };
  TGX_TRY(regrow(ds.seen));
  if (mult) TGX_TRY(regrow(ds.twice));
  ds.base = (int64_t)((uint64_t)ds.base - add_below);
  ds.range = range;
  ds.bitmap_words = words - 4;
  return TGX_OK;
}

static tgx_status distinct_prepare_numeric(tgx_state *st, size_t slot, const tgx_column &c, NumericPrep *prep,
                                           tgx_error *err) {
  const DistinctTask &task = st->plan->distinct[slot];
  DistinctState &ds = st->distinct[slot];
  const bool mult = task.multiplicity;
  prep->prepared = true;
  prep->partitioned = false;
  if (ds.mode == DistinctMode::kUndecided) {
    bool have_range = false;
    int64_t lo = 0, hi = 0;
    if (c.type == TGX_INT64 && ds.has_hint) {
      have_range = true;  // the caller vouches for [lo, hi]; keys outside it are counted and reported
      lo = ds.hint_lo;
      hi = ds.hint_hi;
    } else if (c.type == TGX_INT64 && ds.batch_range_known) {
      have_range = true;  // a coalesced flush of HOST windows: the host saw every value on its way into the arena
      lo = ds.batch_lo;
      hi = ds.batch_hi;
    } else if (distinct_wants_sample(ds, c)) {
      // (a stream of small batches -- DataFusion hands out 8192 rows at a time -- goes straight to the hash set:
      // its inserts need no range, and the read-back of a sample would cost one stream synchronisation per batch)
      DistinctSample got;
      if (ds.sample_ready) {  // tgx_update has read the samples of all key columns at once
        got = ds.sample_host;
        ds.sample_ready = false;
      } else {
        DistinctColDesc d;
        d.values = c.values;
        d.validity = c.validity;
        d.offset = c.offset;
        d.length = c.length;
        d.want_multiplicity = 0;
        d.pad = 0;
        HIP_TRY(ds.sample.reserve(sizeof(DistinctSample)));
        launch_distinct_init(ds.sample.as<DistinctSample>(), nullptr, st->stream);
        launch_distinct_sample(d, ds.sample.as<DistinctSample>(), st->stream);
        HIP_TRY(hipMemcpyAsync(&got, ds.sample.p, sizeof(got), hipMemcpyDeviceToHost, st->stream));
        HIP_TRY(hipStreamSynchronize(st->stream));
      }
      if (got.count == 0) return TGX_OK;          // nothing valid among the sampled rows: decide on a later batch
      have_range = true;
      lo = got.min_v;
      hi = got.max_v;
    }
    bool use_bitmap = false;
    if (c.type == TGX_INT64 && have_range) {
      // unsigned width of [lo, hi]; bitmap when it is at most 16 bits per expected row and <= 2^34
      uint64_t width = (uint64_t)hi - (uint64_t)lo;
      uint64_t expect = std::max<uint64_t>((uint64_t)c.length, g_ctx.distinct_hint);
      if (width < (1ull << 34) && width / 16 <= expect) {
        // a sampled range is widened by 1/64 on either side: the extremes of 2^16 evenly spread values of a column
        // without heavy tails lie within ~width / 2^16 of the column's, i.e. a thousand times closer; what still
        // falls outside is repaired (distinct_resolve).  More slack costs buckets: at 1/4 the 1 G-value id column of
        // the bench needed 1431 slices instead of 985 and the 100 M-value one lost its 2-byte list entries.
        uint64_t slack = ds.has_hint ? 0 : std::min<uint64_t>(width / 64 + 4096, 1ull << 30);
        int64_t base = (lo < INT64_MIN + (int64_t)slack) ? INT64_MIN : lo - (int64_t)slack;
        uint64_t top = (hi > INT64_MAX - (int64_t)slack) ? (uint64_t)INT64_MAX : (uint64_t)(hi + (int64_t)slack);
        ds.base = base;
        ds.range = top - (uint64_t)base + 1;
        use_bitmap = true;
      }
    }
    if (use_bitmap) {
      // whole 2^20-bit slices, so the partitioned path can move slices through LDS
      size_t words = (size_t)(((ds.range + (1u << 20) - 1) >> 20) << 15) + 4;
      ds.bitmap_words = words - 4;
      if (ds.seen.cap < words * 4 && ds.spare_seen.cap >= words * 4) std::swap(ds.seen, ds.spare_seen);
      if (ds.twice.cap < words * 4 && ds.spare_twice.cap >= words * 4) std::swap(ds.twice, ds.spare_twice);
      HIP_TRY(ds.seen.reserve(words * 4));
      HIP_TRY(hipMemsetAsync(ds.seen.p, 0, words * 4, st->stream));
      if (mult) {
        HIP_TRY(ds.twice.reserve(words * 4));
        HIP_TRY(hipMemsetAsync(ds.twice.p, 0, words * 4, st->stream));
      }
      ds.mode = DistinctMode::kBitmap;
      ds.speculative = !ds.has_hint;
    } else {
      ds.mode = DistinctMode::kHash;
    }
  }
  // a later batch whose range the host knows and the bitmap does not cover (ids that grow from batch to batch): the
  // bitmap grows instead of counting the batch's keys as outliers and repairing them through the hash set afterwards
  if (ds.mode == DistinctMode::kBitmap && ds.speculative && !ds.partitioned && ds.batch_range_known &&
      !ds.outliers_possible)
    TGX_TRY(bitmap_grow(st, ds, mult, ds.batch_lo, ds.batch_hi, c.length, err));
  if (ds.mode == DistinctMode::kBitmap && ds.speculative) {
    // can this batch leave keys outside the range?  Not when the host saw every value and the bitmap covers them.
    auto u = [](int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; };
    const bool covered = ds.batch_range_known && u(ds.batch_lo) >= u(ds.base) &&
                         u(ds.batch_hi) - u(ds.base) < ds.range;
    if (!covered) ds.outliers_possible = true;
  }
  if (ds.mode == DistinctMode::kBitmap) {
    bitmap_shape(ds, c.length, mult, &prep->sub_bits, &prep->key16, &prep->n_buckets, &prep->partitioned);
    if (ds.partitioned) prep->partitioned = false;  // an owned slice after tgx_allreduce: plain inserts only
  }
  return TGX_OK;
}

// `stats_slot` >= 0: the partition pass also produces the column's COUNT / MIN / MAX / SUM into that scan slot (the
// numeric scan has skipped the column)
static tgx_status distinct_run_numeric(tgx_state *st, size_t slot, const tgx_column &c, const NumericPrep &prep,
                                       int stats_slot, tgx_error *err, const tgx_column *orig) {
  const DistinctTask &task = st->plan->distinct[slot];
  // what a later repair walks again: never a view into the update's staging scratch (the next update reuses it) --
  // a widened DEVICE Int32 / Float32 column is retained as the caller's 4-byte column and widened again at the
  // repair (retained_numeric_view); staged copies of HOST batches are resolved before tgx_update returns
  const tgx_column &keep = (orig && orig->mem == TGX_MEM_DEVICE && is_numeric32(orig->type)) ? *orig : c;
  DistinctState &ds = st->distinct[slot];
  const bool mult = task.multiplicity;
  DistinctColDesc d;
  d.values = c.values;
  d.validity = c.validity;
  d.offset = c.offset;
  d.length = c.length;
  d.want_multiplicity = mult ? 1 : 0;
  d.pad = 0;
  const uint64_t bytes = (uint64_t)c.length * 8 + (c.validity ? (uint64_t)(c.length + 7) / 8 : 0);
  if (ds.mode == DistinctMode::kUndecided) return TGX_OK;  // nothing valid seen yet
  if (ds.mode == DistinctMode::kBitmap) {
    if (ds.speculative) ds.retained.push_back(keep);  // (a DEVICE view, or a staged one resolved before the update returns)
    if (prep.partitioned) {
      // big batch over a dense range: bucket the keys and replay them against LDS-resident slices
      PartitionParams pp;
      memset(&pp, 0, sizeof(pp));
      pp.values = c.values;
      pp.validity = c.validity;
      pp.offset = c.offset;
      pp.length = c.length;
      pp.base = ds.base;
      pp.range = ds.range;
      pp.sub_bits = prep.sub_bits;
      pp.n_buckets = (uint32_t)prep.n_buckets;
      // runs are padded to 16 slots per (tile, bucket): budget the average load + 25 % + the padding
      const uint64_t tiles = ((uint64_t)c.length + kPartitionTile - 1) / kPartitionTile;
      // the buckets the batch can touch: all of them, unless the host knows the batch's own value range (a coalesced
      // flush of HOST windows) -- a flush of ids that grow lands in a few slices of a bitmap that has grown with the
      // stream, and lists sized for an even spread over ALL slices would overflow into the spill path
      pp.bucket0 = 0;
      pp.n_lists = pp.n_buckets;
      if (ds.batch_range_known) {
        auto u = [](int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; };
        const uint64_t ub = u(ds.base);
        const uint64_t rlo = u(ds.batch_lo) > ub ? u(ds.batch_lo) - ub : 0;
        uint64_t rhi = u(ds.batch_hi) > ub ? u(ds.batch_hi) - ub : 0;
        rhi = std::min(rhi, ds.range - 1);
        if (rlo <= rhi) {
          pp.bucket0 = (uint32_t)(rlo >> pp.sub_bits);
          pp.n_lists = (uint32_t)(rhi >> pp.sub_bits) - pp.bucket0 + 1;
        }
      }
      uint64_t cap = (uint64_t)c.length / pp.n_lists;
      cap = cap + cap / 4 + (prep.key16 ? 32 : 16) * tiles + 4096;
      pp.cap = prep.key16 ? (cap + 31) & ~31ull : (cap + 15) & ~15ull;
      if (pp.cap >= (1ull << 32) - 64) return fail(err, TGX_INTERNAL, "distinct: list capacity out of range");
      pp.want_multiplicity = mult ? 1 : 0;
      pp.key16 = prep.key16 ? 1 : 0;
      static const bool no_probe = getenv("TGX_NO_CLUSTERED_PROBE") && atoi(getenv("TGX_NO_CLUSTERED_PROBE")) != 0;
      pp.probe = no_probe ? 0 : 1;
      HIP_TRY(ds.lists.reserve((uint64_t)pp.n_lists * pp.cap * (prep.key16 ? sizeof(uint16_t) : sizeof(uint32_t))));
      HIP_TRY(ds.cursors.reserve((2 * pp.n_buckets + 1) * sizeof(unsigned long long)));  // (+ the probe's flag)
      pp.lists = ds.lists.as<uint32_t>();
      pp.cursors = ds.cursors.as<unsigned long long>();
      pp.seen = ds.seen.as<uint32_t>();
      pp.twice = mult ? ds.twice.as<uint32_t>() : nullptr;
      const int grid = partition_grid(c.length, g_ctx.n_cu);
      if (stats_slot >= 0) {
        HIP_TRY(ds.stat_partials.reserve((size_t)(grid + 1) * sizeof(ScanPartial)));
        HIP_TRY(ds.outlier_stats.reserve(sizeof(OutlierStats)));
        pp.stats = ds.stat_partials.as<ScanPartial>();
        pp.outliers = ds.outlier_stats.as<OutlierStats>();
      }
      unsigned long long *cnt = ds.counters.as<unsigned long long>();
      static_assert(kCntDistinct == 0 && kCntTwice == 1, "partition_init_kernel clears the two totals together");
      // cursors, limits, the outliers' aggregates and the totals phase 2 recomputes from the slices: one launch
      launch_partition_init(pp, cnt + kCntDistinct, st->stream);
      {
        ProfScope ps(st, "distinct", bytes);
        launch_partition(pp, cnt, g_ctx.n_cu, st->stream);
        HIP_TRY(launch_bucket_apply(pp, cnt, st->stream));
      }
      if (stats_slot >= 0) {
        ScanLaunch L;
        memset(&L, 0, sizeof(L));
        L.cols[0].length = c.length;
        L.cols[0].is_float = 0;
        L.acc_index[0] = stats_slot;
        launch_partition_outlier_stats(pp.outliers, pp.stats, grid, st->stream);
        launch_scan_reduce_only(L, 1, grid + 1, pp.stats, st->d_scan_acc.as<ScanAcc>(), st->stream);
      }
    } else {
      ProfScope ps(st, "distinct", bytes);
      launch_distinct_bitmap(d, bitmap_view(ds), ds.counters.as<unsigned long long>(), st->stream);
    }
  } else {
    if (ds.fp_staged) TGX_TRY(distinct_resolve(st, slot, err));  // a second batch: the table takes over
    if (ds.capacity == 0 && fp_lists_fit(c)) {
      // the first big batch of a key set without a dense range: mixed keys through partitioned lists, deduplicated
      // in LDS (kernels/distinct.hip, key_*) -- no global atomic per key; the lists are the set until the table is needed
      TGX_TRY(fp_lists_prepare(st, ds, c.length, 8, err));
      FpLists l1, l2;
      fp_views(ds, &l1, &l2);
      ProfScope ps(st, "distinct", bytes), ps_lists(st, "distinct_lists", 0);
      launch_key_lists(d, l1, l2, mult ? 1 : 0, ds.fp_per_list.as<uint2>(), ds.counters.as<unsigned long long>(),
                       st->stream);
      ds.fp_staged = true;
      ds.retained.push_back(keep);  // (a DEVICE view, or a staged one looked at before the update returns)
      return TGX_OK;
    }
    TGX_TRY(hash_ensure(st, ds, mult, (uint64_t)c.length, err));
    ProfScope ps(st, "distinct", bytes);
    launch_distinct_hash(d, hash_view(ds), ds.counters.as<unsigned long long>(), st->stream);
  }
  return TGX_OK;
}

// A retained DEVICE Int32 / Float32 column (distinct_run_numeric keeps the caller's 4-byte view, not the update's
// widened scratch) as the Int64 / Float64 view the repair kernels read: widened again into `tmp`, exactly as
// stage_column did for the update (window from slot offset & ~63, validity re-based by bytes).
static tgx_status retained_numeric_view(tgx_state *st, const tgx_column &col, std::vector<std::unique_ptr<DevBuf>> &tmp,
                                        tgx_column *out, tgx_error *err) {
  *out = col;
  if (!is_numeric32(col.type)) return TGX_OK;
  const int64_t e0 = col.offset & ~(int64_t)63;
  const int64_t slots = col.offset - e0 + col.length;
  tmp.emplace_back(new DevBuf());
  DevBuf *w = tmp.back().get();
  HIP_TRY(w->reserve((size_t)slots * 8 + 16));
  launch_widen32((const uint8_t *)col.values + (size_t)e0 * 4, w->p, slots, col.type == TGX_FLOAT32 ? 1 : 0, g_ctx.n_cu,
                 st->stream);
  out->type = col.type == TGX_INT32 ? TGX_INT64 : TGX_FLOAT64;
  out->values = w->p;
  out->validity = col.validity ? col.validity + (e0 >> 3) : nullptr;
  out->offset = col.offset - e0;
  return TGX_OK;
}

// The host is about to look at the key set (counts, export, exchange, merge) or the caller may release the batches:
// keys that fell outside a sampled range are brought in now.  The bitmap moves into a hash set and the retained
// batches are walked once more for their outliers only (disjoint from the bitmap's keys, so multiplicities stay right).
tgx_status tgx::distinct_resolve(tgx_state *st, size_t slot, tgx_error *err) {
  DistinctState &ds = st->distinct[slot];
  std::vector<std::unique_ptr<DevBuf>> widened;  // freed on the way out: every path below drains the stream first
  if (ds.fp_staged && st->device_ready) {
    // Utf8 fingerprint lists: into the table -- or, if a list overflowed, the batch again, through the table
    const bool mult = st->plan->distinct[slot].multiplicity;
    unsigned long long c[kNumDistinctCounters];
    TGX_TRY(distinct_read_counters(st, ds, c, err));
    ds.fp_staged = false;
    if (c[kCntOutOfRange] != 0) {
      if (ds.retained.empty())
        return fail(err, TGX_INTERNAL, "distinct: overflowed fingerprint lists and no batch to redo");
      HIP_TRY(hipMemsetAsync(ds.counters.p, 0, kNumDistinctCounters * sizeof(unsigned long long), st->stream));
      const DistinctTask &task = st->plan->distinct[slot];
      if (!task.tuple.empty()) {  // the retained columns are the tuple's components, in order
        std::vector<const tgx_column *> cols;
        for (const tgx_column &col : ds.retained) cols.push_back(&col);
        TupleDesc d;
        TGX_TRY(tuple_desc_of(cols, mult, &d, err));
        TGX_TRY(hash_ensure(st, ds, mult, (uint64_t)d.length, err));
        launch_distinct_tuple(d, hash_view(ds), ds.counters.as<unsigned long long>(), st->stream);
      } else
      for (const tgx_column &kept : ds.retained) {
        tgx_column col;
        TGX_TRY(retained_numeric_view(st, kept, widened, &col, err));
        TGX_TRY(hash_ensure(st, ds, mult, (uint64_t)col.length, err));
        if (!ds.wide) {  // a numeric key column
          DistinctColDesc d;
          d.values = col.values;
          d.validity = col.validity;
          d.offset = col.offset;
          d.length = col.length;
          d.want_multiplicity = mult ? 1 : 0;
          d.pad = 0;
          launch_distinct_hash(d, hash_view(ds), ds.counters.as<unsigned long long>(), st->stream);
          continue;
        }
        const bool view = col.type == TGX_UTF8_VIEW;
        launch_distinct_utf8(col.offsets, col.data, view ? col.values : nullptr, view ? col.variadic : nullptr,
                             col.validity, col.offset, col.length, col.type == TGX_LARGE_UTF8, mult ? 1 : 0,
                             hash_view(ds), ds.counters.as<unsigned long long>(), st->stream);
      }
    } else {
      FpLists l1, l2;
      fp_views(ds, &l1, &l2);
      TGX_TRY(hash_ensure(st, ds, mult, c[kCntDistinct], err));
      if (ds.wide)
        launch_fp_insert(l2, hash_view(ds), mult ? 1 : 0, st->stream);
      else
        launch_key_insert(l2, hash_view(ds), mult ? 1 : 0, st->stream);
    }
    HIP_TRY(hipStreamSynchronize(st->stream));
    ds.retained.clear();
    return TGX_OK;
  }
  if (!ds.speculative || ds.retained.empty() || !st->device_ready) {
    ds.retained.clear();
    return TGX_OK;
  }
  const bool mult = st->plan->distinct[slot].multiplicity;
  unsigned long long c[kNumDistinctCounters];
  TGX_TRY(distinct_read_counters(st, ds, c, err));
  const uint64_t n_out = c[kCntOutOfRange];
  if (n_out == 0 || ds.mode != DistinctMode::kBitmap) {
    ds.retained.clear();
    ds.outliers_possible = false;  // (the counters have just said so)
    return TGX_OK;
  }
  const int64_t old_base = ds.base;
  const uint64_t old_range = ds.range;
  TGX_TRY(bitmap_to_hash(st, ds, mult, n_out, err));
  TGX_TRY(hash_ensure(st, ds, mult, n_out, err));
  for (const tgx_column &kept : ds.retained) {
    tgx_column col;
    TGX_TRY(retained_numeric_view(st, kept, widened, &col, err));
    DistinctColDesc d;
    d.values = col.values;
    d.validity = col.validity;
    d.offset = col.offset;
    d.length = col.length;
    d.want_multiplicity = mult ? 1 : 0;
    d.pad = 0;
    launch_distinct_outliers(d, old_base, old_range, hash_view(ds), ds.counters.as<unsigned long long>(), st->stream);
  }
  HIP_TRY(hipMemsetAsync(ds.counters.as<unsigned long long>() + kCntOutOfRange, 0, sizeof(unsigned long long), st->stream));
  HIP_TRY(hipStreamSynchronize(st->stream));
  ds.retained.clear();
  ds.speculative = false;
  return TGX_OK;
}

tgx_status tgx::distinct_resolve_all(tgx_state *st, tgx_error *err) {
  // one read-back for all tasks (a synchronisation each would cost a step with several uniqueness checks more than
  // the checks themselves at small sizes); only a task that really has outliers goes through the repair
  bool pending = false;
  for (auto &ds : st->distinct) pending |= (ds.speculative || ds.fp_staged) && !ds.retained.empty();
  if (!pending || !st->device_ready || !st->d_distinct_counters.p) {
    for (auto &ds : st->distinct) ds.retained.clear();
    return TGX_OK;
  }
  std::vector<unsigned long long> all(st->distinct.size() * kNumDistinctCounters);
  TGX_TRY(pinned_readback(st, all.size() * sizeof(unsigned long long), err));
  HIP_TRY(hipMemcpyAsync(st->h_pinned, st->d_distinct_counters.p, all.size() * sizeof(unsigned long long),
                         hipMemcpyDeviceToHost, st->stream));
  HIP_TRY(hipStreamSynchronize(st->stream));
  memcpy(all.data(), st->h_pinned, all.size() * sizeof(unsigned long long));
  for (size_t k = 0; k < st->distinct.size(); k++) {
    DistinctState &ds = st->distinct[k];
    if ((ds.speculative || ds.fp_staged) && !ds.retained.empty() &&
        all[k * kNumDistinctCounters + kCntOutOfRange] != 0)
      TGX_TRY(distinct_resolve(st, k, err));
    else {
      ds.retained.clear();
      ds.outliers_possible = false;  // (the counters have just said so)
    }
  }
  return TGX_OK;
}

// The pivots of the pairs of one launch (kernels/comoments.hip, como_pivot_kernel): picked from the first batches that
// bring rows -- the kernel leaves a pair alone once rows have been folded into it; after a few batches nothing is
// launched any more (a stream of 8192-row batches must not pay a launch per batch for a decision long taken).
static tgx_status como_pivots(tgx_state *st, const ComomentLaunch &L, int n_pairs, tgx_error *err) {
  (void)err;
  bool want = false;
  if (st->como_pivot_tries.size() < st->plan->como.size()) st->como_pivot_tries.assign(st->plan->como.size(), 0);
  for (int k = 0; k < n_pairs; k++)
    if (L.pairs[k].length > 0 && st->como_pivot_tries[L.acc_index[k]] < 4) {
      st->como_pivot_tries[L.acc_index[k]]++;
      want = true;
    }
  if (want) launch_como_pivot(L, n_pairs, st->d_como_acc.as<ComomentAcc>(), st->stream);
  return TGX_OK;
}

// checks one batch's column views against the plan and the state (types, row counts, required buffers); *nrows_out =
// the batch's rows.  Nothing is allocated here: it runs once per 8192-row batch.
namespace {
struct BatchTraits {
  bool any_host = false, any_utf8 = false;
  bool coalescible = true;  // every used column is of a kind the segment gather takes (kernels/gather.hip)
};
}  // namespace

static tgx_status update_validate(const tgx_plan *plan, tgx_state *st, const tgx_column *columns, size_t n_columns,
                                  int64_t *nrows_out, BatchTraits *traits, tgx_error *err) {
  (void)n_columns;
  const std::vector<char> &used = plan->used, &reads_values = plan->reads_values;
  int64_t nrows = -1;
  for (int i = 0; i < plan->n_columns_needed; i++) {
    if (!used[i]) continue;
    const tgx_column &c = columns[i];
    if (c.length < 0 || c.offset < 0) return fail(err, TGX_INVALID_ARGUMENT, "column %d: negative length/offset", i);
    if (nrows < 0) nrows = c.length;
    if (c.length != nrows)
      return fail(err, TGX_INVALID_ARGUMENT, "column %d has %lld rows, expected %lld", i, (long long)c.length,
                  (long long)nrows);
    if (c.type < TGX_INT64 || c.type > TGX_FLOAT32) return fail(err, TGX_INVALID_ARGUMENT, "column %d: unknown type %d", i, c.type);
    if (c.mem != TGX_MEM_HOST && c.mem != TGX_MEM_DEVICE)
      return fail(err, TGX_INVALID_ARGUMENT, "column %d: unknown memory space %d", i, c.mem);
    if (st->col_types[i] == 0) st->col_types[i] = c.type;
    if (st->col_types[i] != c.type)
      return fail(err, TGX_INVALID_ARGUMENT, "column %d changed type between batches (%d -> %d)", i,
                  st->col_types[i], c.type);
    const bool host = c.mem == TGX_MEM_HOST;
    traits->any_host |= host;
    traits->any_utf8 |= c.type == TGX_UTF8;
    // string windows need their first / last offsets (Utf8View: the stretches its views point into; dictionaries:
    // theirs) on the host: HOST batches only (what DataFusion streams); DEVICE strings keep the immediate path
    traits->coalescible &= is_numeric(c.type) || is_numeric32(c.type) || (is_string(c.type) && host) ||
                           (c.type == TGX_UTF8_VIEW && host) ||
                           (c.type == TGX_DICT32_UTF8 && host && c.dictionary && c.dictionary->mem == TGX_MEM_HOST);
    if (c.length > 0) {
      if ((is_numeric(c.type) || is_numeric32(c.type)) && reads_values[i] && !c.values)
        return fail(err, TGX_INVALID_ARGUMENT, "column %d: values is NULL", i);
      if ((c.type == TGX_UTF8 || c.type == TGX_LARGE_UTF8) && !c.offsets)
        return fail(err, TGX_INVALID_ARGUMENT, "column %d: offsets is NULL", i);
    }
    if (c.type == TGX_UTF8_VIEW && c.length > 0) {
      if (!c.values) return fail(err, TGX_INVALID_ARGUMENT, "column %d: views (values) is NULL", i);
      if (c.n_variadic < 0 || (c.n_variadic > 0 && !c.variadic))
        return fail(err, TGX_INVALID_ARGUMENT, "column %d: malformed variadic buffer list", i);
      if (c.mem == TGX_MEM_HOST && c.n_variadic > 0 && !c.variadic_sizes)
        return fail(err, TGX_INVALID_ARGUMENT, "column %d: HOST Utf8View columns need variadic_sizes", i);
    }
    if (c.type == TGX_DICT32_UTF8) {
      const tgx_column *dc = c.dictionary;
      if (!dc || !is_string(dc->type))
        return fail(err, TGX_INVALID_ARGUMENT, "column %d: a Dictionary<Int32, Utf8> column needs a Utf8/LargeUtf8 dictionary", i);
      if (dc->length < 0 || dc->offset < 0 || (dc->length > 0 && !dc->offsets))
        return fail(err, TGX_INVALID_ARGUMENT, "column %d: malformed dictionary", i);
      if (c.length > 0 && reads_values[i] && !c.values)
        return fail(err, TGX_INVALID_ARGUMENT, "column %d: indices (values) is NULL", i);
    }
  }
  *nrows_out = nrows < 0 ? 0 : nrows;
  return TGX_OK;
}

static tgx_status update_impl(const tgx_plan *plan, tgx_state *st, const tgx_column *columns, int64_t nrows,
                              tgx_error *err);
static tgx_status coalesce_append(const tgx_plan *plan, tgx_state *st, const tgx_column *columns, int64_t nrows,
                                  const BatchTraits &traits, bool *taken, tgx_error *err);
constexpr int64_t kCoalesceMaxRows = 1 << 16;        // batches up to this many rows are coalesced

extern "C" tgx_status tgx_update(const tgx_plan *plan, tgx_state *st, const tgx_column *columns,
                                 size_t n_columns, tgx_error *err) try {
  if (!plan || !st || st->plan != plan) return fail(err, TGX_INVALID_ARGUMENT, "state does not belong to plan");
  if ((int)n_columns < plan->n_columns_needed)
    return fail(err, TGX_INVALID_ARGUMENT, "plan reads column %d but only %zu columns were passed",
                plan->n_columns_needed - 1, n_columns);
  if (n_columns > 0 && !columns) return fail(err, TGX_INVALID_ARGUMENT, "columns is NULL");
  TGX_TRY(need_device(err));
  int64_t nrows = 0;
  BatchTraits traits;
  TGX_TRY(update_validate(plan, st, columns, n_columns, &nrows, &traits, err));
  if (nrows == 0) {  // an empty RecordBatch (streams interleave them): nothing to note, nothing to flush for
    st->batches++;
    return TGX_OK;
  }
  // a small batch is only noted (kernels/gather.hip): no launch, no synchronisation per 8192-row batch
  const Coalescer &co = st->coalesce;
  if (traits.coalescible && nrows > 0 && nrows <= kCoalesceMaxRows && !co.disabled && !co.flushing) {
    bool taken = false;
    TGX_TRY(coalesce_append(plan, st, columns, nrows, traits, &taken, err));
    if (taken) return TGX_OK;
  }
  bind_thread();  // (the noted-only path above makes no HIP call: it binds where it does, in the arena set-up and the flush)
  TGX_TRY(coalesce_flush(st, err));  // batches stay in order
  return update_impl(plan, st, columns, nrows, err);
} catch (...) {
  return tgx::abi_exception(err);
}

// one batch through the fused pass: device views of its columns, then every kernel of the plan
static tgx_status update_impl(const tgx_plan *plan, tgx_state *st, const tgx_column *columns, int64_t nrows,
                              tgx_error *err) {
  TGX_TRY(state_init_device(st, err));
  const std::vector<char> &used = plan->used;

  // device views of every used column
  st->staging_used = 0;
  if (st->arena_used) (void)hipStreamSynchronize(st->stream);  // an update that failed half way left it in use
  st->arena_used = 0;
  st->host_direct = false;
  st->dict_views.clear();
  st->pending_widen.clear();
  std::vector<tgx_column> dev(plan->n_columns_needed);
  // 4-byte numeric columns are widened to 8-byte values only for the passes that need them so
  const std::vector<char> &needs_wide = plan->needs_wide;
  bool any_host = false;
  for (int i = 0; i < plan->n_columns_needed; i++) {
    if (!used[i]) continue;
    if (columns[i].mem == TGX_MEM_HOST && columns[i].length > 0) any_host = true;
    if (columns[i].length == 0) {
      dev[i] = columns[i];
      if (is_numeric32(dev[i].type) && needs_wide[i]) dev[i].type = dev[i].type == TGX_INT32 ? TGX_INT64 : TGX_FLOAT64;
      continue;
    }
    TGX_TRY(stage_column(st, columns[i], &dev[i], err, needs_wide[i] != 0));
  }
  const bool arena_in_use = st->arena_used != 0;
  if (arena_in_use)
    HIP_TRY(hipMemcpyAsync(st->arena_dev[st->arena_cur].p, st->arena_host[st->arena_cur], st->arena_used,
                           hipMemcpyHostToDevice, st->stream));
  for (const auto &w : st->pending_widen) launch_widen32(w.src, w.dst, w.n, w.is_float, g_ctx.n_cu, st->stream);
  st->pending_widen.clear();

  if (nrows > 0) {
    // ---- what rides on the numeric scan of this batch (kernels/scan.hip) ----
    // A COMOMENTS pair whose columns are plain 8-byte numerics is scanned by ONE workgroup per tile pair
    // (scan_pair_kernel): both columns' own aggregates and the co-moments from one read.  A KLL task of a batch big
    // enough to be sampled hands its sampler to the scan of its column (scan_kll_kernel / the pair kernel).  So a
    // suite with range, quantile and correlation checks on the same columns reads them once.
    const int n_plan_cols = plan->n_columns_needed;
    std::vector<int> kll_on_col(n_plan_cols, -1), pair_of_col(n_plan_cols, -1);
    std::vector<char> kll_fused(plan->kll.size(), 0), como_fused(plan->como.size(), 0);
    auto scan_slot_of = [&](int col) -> int {
      for (size_t q = 0; q < plan->scan.size(); q++)
        if (plan->scan[q].column == col) return (int)q;
      return -1;
    };
    auto plain8 = [&](int col) { return is_numeric(dev[col].type) && dev[col].values != nullptr; };
    struct FusedPair {
      int como, x, y;
      ScanColDesc dx, dy;
    };
    std::vector<FusedPair> fused_pairs;
    // ---- APPROX_DISTINCT: the HyperLogLog lane of the scan for numeric columns; the exact key set elsewhere ----
    // (decided by the first batch's column type; a column whose scan also carries variance lanes keeps the exact set)
    std::vector<int> hll_on_col(n_plan_cols, -1);
    std::vector<char> distinct_is_idle(plan->distinct.size(), 0);
    for (size_t q = 0; q < plan->hll.size(); q++) {
      const HllTask &t = plan->hll[q];
      const int type = dev[t.column].type;
      const bool numeric = is_numeric(type) || is_numeric32(type);
      const bool lane = numeric && plan->distinct[t.distinct_slot].approx_only && !plan->scan[t.scan_slot].variance;
      if (st->hll_mode[q] == 0) st->hll_mode[q] = lane ? 1 : 2;
      if (st->hll_mode[q] == 1) {
        if (dev[t.column].values) hll_on_col[t.column] = (int)q;
        distinct_is_idle[t.distinct_slot] = 1;  // (its approx_only key set has nothing to do)
      }
    }
    auto distinct_idle = [&](size_t q) { return distinct_is_idle[q] != 0; };
    auto has_hll = [&](int col) { return hll_on_col[col] >= 0; };  // its own scan launch: fuses with nothing else
    if (nrows >= (1 << 20)) {
      for (size_t q = 0; q < plan->kll.size(); q++) {
        const int col = plan->kll[q].column;
        if (plain8(col) && !has_hll(col) && kll_on_col[col] < 0 && kll_scan_eligible(nrows)) {
          kll_on_col[col] = (int)q;
          kll_fused[q] = 1;
        }
      }
      for (size_t q = 0; q < plan->como.size() && fused_pairs.size() < (size_t)kMaxPairsPerLaunch; q++) {
        const int x = plan->como[q].col_x, y = plan->como[q].col_y;
        if (x == y || !plain8(x) || !plain8(y) || pair_of_col[x] >= 0 || pair_of_col[y] >= 0) continue;
        if (has_hll(x) || has_hll(y)) continue;
        const int sx = scan_slot_of(x), sy = scan_slot_of(y);
        if ((sx >= 0 && plan->scan[sx].variance) || (sy >= 0 && plan->scan[sy].variance)) continue;
        FusedPair fp;
        fp.como = (int)q;
        fp.x = x;
        fp.y = y;
        fill_scan_desc(dev[x], false, nullptr, &fp.dx);
        fill_scan_desc(dev[y], false, nullptr, &fp.dy);
        if (fp.dx.head != fp.dy.head || fp.dx.n_tiles != fp.dy.n_tiles) continue;  // tiles must line up
        pair_of_col[x] = pair_of_col[y] = (int)fused_pairs.size();
        como_fused[q] = 1;
        fused_pairs.push_back(fp);
      }
    }
    // waves of a fused launch and the most rows one of them sees (sizes the sampler's buffers)
    auto fused_blocks = [&](const ScanColDesc &d, int n_tasks) -> int {
      const int64_t units = d.n_tiles > 0 ? d.n_tiles : (d.length + 63) / 64;
      int64_t want = (units + 4 * kWavesPerBlock - 1) / (4 * kWavesPerBlock);
      const int64_t cap = std::max(32, (g_ctx.n_cu * 3) / std::max(1, n_tasks));
      return (int)std::max<int64_t>(1, std::min(want, cap));
    };
    auto rows_per_wave = [&](const ScanColDesc &d, int blocks) -> int64_t {
      const int64_t waves = (int64_t)blocks * kWavesPerBlock;
      if (d.n_tiles > 0) return (d.n_tiles + waves - 1) / waves * kTileRows;
      return ((d.length + 63) / 64 + waves - 1) / waves * 64;
    };
    // ---- exact uniqueness over dense Int64 keys takes the column's range aggregates along (kernels/distinct.hip,
    // partition_kernel<.., STATS>): such a column is not scanned at all -- it crosses HBM once for MIN / MAX / SUM /
    // COUNT and COUNT(DISTINCT) together.  Decided here, before the scan is queued, from a sample of the batch.
    std::vector<NumericPrep> dprep(plan->distinct.size());
    std::vector<int> stats_by_partition(plan->scan.size(), -1);
    TGX_TRY(distinct_sample_all(st, dev.data(), err));
    for (size_t q = 0; q < plan->distinct.size(); q++) {
      const DistinctTask &t = plan->distinct[q];
      if (!t.tuple.empty() || !is_numeric(dev[t.column].type) || dev[t.column].length == 0 || distinct_idle(q)) continue;
      TGX_TRY(distinct_prepare_numeric(st, q, dev[t.column], &dprep[q], err));
      if (dprep[q].partitioned && dev[t.column].type == TGX_INT64 && t.scan_slot >= 0 &&
          !plan->scan[t.scan_slot].variance && pair_of_col[t.column] < 0 && kll_on_col[t.column] < 0)
        stats_by_partition[t.scan_slot] = (int)q;
    }
    // ---- the key columns' uniqueness passes go FIRST (numeric keys: everything they need is decided) and an event
    // marks their end: across ranks the exchange of the key sets (tgx_allreduce) can then run on a second stream
    // while the scan of the other columns below is still running (the >= 6x target is set by the exchange)
    std::vector<char> distinct_done(plan->distinct.size(), 0);
    for (size_t q = 0; q < plan->distinct.size(); q++) {
      const DistinctTask &t = plan->distinct[q];
      if (!t.tuple.empty() || distinct_idle(q) || !is_numeric(dev[t.column].type) || dev[t.column].length == 0) continue;
      const int stats_slot = (t.scan_slot >= 0 && stats_by_partition[t.scan_slot] == (int)q) ? t.scan_slot : -1;
      TGX_TRY(distinct_update(st, q, dev[t.column], err, nullptr, &dprep[q], stats_slot, &columns[t.column]));
      distinct_done[q] = 1;
    }
    if (!st->keys_ready) HIP_TRY(hipEventCreateWithFlags(&st->keys_ready, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(st->keys_ready, st->stream));
    // (the event stands for the key sets only when every key set of the plan was touched before it: string,
    //  dictionary and tuple sets are updated further down, behind the scan)
    bool all_early = true;
    for (size_t q = 0; q < plan->distinct.size(); q++) all_early &= distinct_done[q] || distinct_idle(q);
    // ... and for the running MIN / MAX the facts round reads (allreduce.cpp tightens a bitmap's range with them) only
    // when they, too, were produced before it: a key column whose aggregates come from the scan below -- a pass that is
    // not partitioned, variance / pair / KLL / HLL lanes on the key column -- would have its ScanAcc read on the second
    // stream while the scan is still writing it
    for (size_t q = 0; q < plan->distinct.size(); q++) {
      const DistinctTask &t = plan->distinct[q];
      if (!distinct_done[q] || t.scan_slot < 0 || st->distinct[q].has_hint) continue;
      all_early &= stats_by_partition[t.scan_slot] == (int)q;
    }
    st->keys_ready_recorded = all_early && (st->passes == 0 || st->keys_ready_recorded);
    st->passes++;
    // (a state that takes part in exchanges leaves two workgroup slots per CU to the second stream's kernels: the
    //  scan is HBM-bound from 4 workgroups per CU upwards)
    const int scan_per_cu = st->exchange_expected ? 6 : 8;
    // ---- numeric scan: all columns of the batch in launches of <= kMaxColsPerLaunch ----
    {
      std::vector<ScanColDesc> descs, kll_descs, hll_descs;
      std::vector<int32_t> index, kll_index, hll_index;
      std::vector<int> kll_slots;
      for (size_t s = 0; s < plan->scan.size(); s++) {
        const tgx_column &c = dev[plan->scan[s].column];
        if (!is_numeric(c.type) && !is_numeric32(c.type)) {
          // a scan task that only exists for DISTINCT's range decision does not apply to strings
          bool needed_by_stats = false;
          for (size_t i = 0; i < plan->specs.size(); i++)
            if (plan->specs[i].kind == TGX_CHECK_NUMERIC_STATS && plan->bind[i].slot == (int)s) needed_by_stats = true;
          if (needed_by_stats)
            return fail(err, TGX_INVALID_ARGUMENT, "NUMERIC_STATS on non-numeric column %d (type %d)",
                        plan->scan[s].column, c.type);
          continue;
        }
        if (pair_of_col[plan->scan[s].column] >= 0) continue;  // scanned with its partner below
        if (stats_by_partition[s] >= 0) continue;                // its DISTINCT pass brings the aggregates
        if (kll_on_col[plan->scan[s].column] < 0 && hll_on_col[plan->scan[s].column] < 0) {
          // a scan that only feeds DISTINCT's range decision is not needed once the range is declared
          bool bound = false, all_hinted = true, any_distinct = false;
          for (size_t i = 0; i < plan->specs.size(); i++) {
            const SpecBinding &b = plan->bind[i];
            if (b.slot == (int)s && (b.kind == TGX_CHECK_NUMERIC_STATS || (b.kind == TGX_CHECK_COUNT && b.count_src == Source::kScan)))
              bound = true;
          }
          for (size_t dd = 0; dd < plan->distinct.size(); dd++)
            if (plan->distinct[dd].scan_slot == (int)s) {
              any_distinct = true;
              all_hinted &= st->distinct[dd].has_hint;
            }
          if (!bound && any_distinct && all_hinted) continue;
        }
        ScanColDesc d;
        fill_scan_desc(c, plan->scan[s].variance, st->d_pivots.as<double>() + s, &d);
        if (hll_on_col[plan->scan[s].column] >= 0 && !plan->scan[s].variance) {
          d.skip_stats = plan->scan[s].stats_needed ? 0 : 1;
          d.hll_regs = st->d_hll.as<uint8_t>() + (size_t)hll_on_col[plan->scan[s].column] * kHllRegisters;
          hll_descs.push_back(d);
          hll_index.push_back((int32_t)s);
          hll_on_col[plan->scan[s].column] = -2;  // taken
          continue;
        }
        if (kll_on_col[plan->scan[s].column] >= 0) {
          kll_descs.push_back(d);
          kll_index.push_back((int32_t)s);
          kll_slots.push_back(kll_on_col[plan->scan[s].column]);
        } else {
          descs.push_back(d);
          index.push_back((int32_t)s);
        }
      }
      // sampled columns without a scan task of their own (a KLL check alone): scanned all the same, their column
      // aggregates are dropped (acc_index -1)
      for (size_t q = 0; q < plan->kll.size(); q++) {
        const int col = plan->kll[q].column;
        if (!kll_fused[q] || pair_of_col[col] >= 0 || scan_slot_of(col) >= 0) continue;
        ScanColDesc d;
        fill_scan_desc(dev[col], false, nullptr, &d);
        kll_descs.push_back(d);
        kll_index.push_back(-1);
        kll_slots.push_back((int)q);
      }
      // launches of <= kMaxColsPerLaunch columns; descriptors travel in the kernel arguments
      for (size_t c0 = 0; c0 < descs.size(); c0 += kMaxColsPerLaunch) {
        const int n = (int)std::min<size_t>(kMaxColsPerLaunch, descs.size() - c0);
        ScanLaunch L;
        memset(&L, 0, sizeof(L));
        int blocks = 1;
        uint64_t chunk_bytes = 0;
        bool chunk_var = false;
        for (int k = 0; k < n; k++) {
          L.cols[k] = descs[c0 + k];
          L.acc_index[k] = index[c0 + k];
          blocks = std::max(blocks, scan_blocks_for(L.cols[k], n, scan_per_cu));
          chunk_bytes += (uint64_t)L.cols[k].length * (L.cols[k].elem32 ? 4 : 8) +
                         (L.cols[k].validity ? (uint64_t)(L.cols[k].length + 7) / 8 : 0);
          chunk_var |= L.cols[k].want_variance != 0;
        }
        HIP_TRY(st->d_scan_partials.reserve((size_t)n * blocks * sizeof(ScanPartial)));
        if (chunk_var) launch_scan_pivot(L, n, st->d_pivots.as<double>(), st->d_pivot_set.as<int32_t>(), st->stream);
        {
          ProfScope ps(st, "scan", chunk_bytes);
          launch_scan_main_only(L, n, blocks, st->d_scan_partials.as<ScanPartial>(), st->d_scan_acc.as<ScanAcc>(),
                                st->stream);
        }
        if (blocks > 1)  // one workgroup per column folds into the running state itself (small batches: one launch)
          launch_scan_reduce_only(L, n, blocks, st->d_scan_partials.as<ScanPartial>(), st->d_scan_acc.as<ScanAcc>(),
                                  st->stream);
      }
      // columns whose KLL samplers ride on the scan: columns of one launch share the sampling level (same rows)
      for (size_t c0 = 0; c0 < kll_descs.size(); c0 += kMaxColsPerLaunch) {
        const int n = (int)std::min<size_t>(kMaxColsPerLaunch, kll_descs.size() - c0);
        ScanLaunch L;
        memset(&L, 0, sizeof(L));
        int blocks = 1;
        uint64_t chunk_bytes = 0;
        bool chunk_var = false;
        for (int k = 0; k < n; k++) {
          L.cols[k] = kll_descs[c0 + k];
          L.acc_index[k] = kll_index[c0 + k];
          blocks = std::max(blocks, fused_blocks(L.cols[k], n));
          chunk_bytes += (uint64_t)L.cols[k].length * 8 + (L.cols[k].validity ? (uint64_t)(L.cols[k].length + 7) / 8 : 0);
          chunk_var |= L.cols[k].want_variance != 0;
        }
        size_t lds = 0;
        for (int k = 0; k < n; k++) {
          TGX_TRY(kll_scan_prepare(st, (size_t)kll_slots[c0 + k], nrows, blocks * kWavesPerBlock,
                                   rows_per_wave(L.cols[k], blocks), &L.cols[k].kll, err));
          lds = std::max(lds, (size_t)kWavesPerBlock * (((size_t)1 << L.cols[k].kll.top) + kTileRows) * sizeof(double));
        }
        HIP_TRY(st->d_scan_partials.reserve((size_t)n * blocks * sizeof(ScanPartial)));
        if (chunk_var) launch_scan_pivot(L, n, st->d_pivots.as<double>(), st->d_pivot_set.as<int32_t>(), st->stream);
        {
          ProfScope ps(st, "scan", chunk_bytes);
          launch_scan_kll(L, n, blocks, lds, st->d_scan_partials.as<ScanPartial>(), st->stream);
        }
        launch_scan_reduce_only(L, n, blocks, st->d_scan_partials.as<ScanPartial>(), st->d_scan_acc.as<ScanAcc>(),
                                st->stream);
      }
      // columns with the HyperLogLog lane on (APPROX_DISTINCT)
      for (size_t c0 = 0; c0 < hll_descs.size(); c0 += kMaxColsPerLaunch) {
        const int n = (int)std::min<size_t>(kMaxColsPerLaunch, hll_descs.size() - c0);
        ScanLaunch L;
        memset(&L, 0, sizeof(L));
        int blocks = 1;
        uint64_t chunk_bytes = 0;
        for (int k = 0; k < n; k++) {
          L.cols[k] = hll_descs[c0 + k];
          L.acc_index[k] = hll_index[c0 + k];
          blocks = std::max(blocks, scan_blocks_for(L.cols[k], n));
          chunk_bytes += (uint64_t)L.cols[k].length * (L.cols[k].elem32 ? 4 : 8) +
                         (L.cols[k].validity ? (uint64_t)(L.cols[k].length + 7) / 8 : 0);
        }
        HIP_TRY(st->d_scan_partials.reserve((size_t)n * blocks * sizeof(ScanPartial)));
        HIP_TRY(st->d_hll_rows.reserve((size_t)n * blocks * kHllRegisters));
        for (int k = 0; k < n; k++) L.cols[k].hll = st->d_hll_rows.as<uint8_t>() + (size_t)k * blocks * kHllRegisters;
        {
          ProfScope ps(st, "scan", chunk_bytes), ps_hll(st, "scan_hll", chunk_bytes);
          launch_scan_hll(L, n, blocks, st->d_scan_partials.as<ScanPartial>(), st->stream);
        }
        launch_scan_reduce_only(L, n, blocks, st->d_scan_partials.as<ScanPartial>(), st->d_scan_acc.as<ScanAcc>(),
                                st->stream);
      }
      // COMOMENTS pairs: both columns and their co-moments from one read
      if (!fused_pairs.empty()) {
        const int n = (int)fused_pairs.size();
        // grouped by (x type, y type): one launch of the kernel instance of each combination (kernels/scan.hip)
        std::stable_sort(fused_pairs.begin(), fused_pairs.end(), [](const FusedPair &a, const FusedPair &b) {
          return 2 * a.dx.is_float + a.dy.is_float > 2 * b.dx.is_float + b.dy.is_float;
        });
        ScanPairLaunch PL;
        ScanLaunch RL;  // the same columns as the reduce kernel wants them: [2 k] = x, [2 k + 1] = y
        ComomentLaunch CL;
        memset(&PL, 0, sizeof(PL));
        memset(&RL, 0, sizeof(RL));
        memset(&CL, 0, sizeof(CL));
        int blocks = 1;
        uint64_t chunk_bytes = 0;
        for (int k = 0; k < n; k++) blocks = std::max(blocks, fused_blocks(fused_pairs[k].dx, n));
        size_t lds = 0;
        for (int k = 0; k < n; k++) {
          FusedPair &fp = fused_pairs[k];
          ScanPairDesc &P = PL.pairs[k];
          P.x = fp.dx;
          P.y = fp.dy;
          P.x_acc = scan_slot_of(fp.x);
          P.y_acc = scan_slot_of(fp.y);
          P.como_acc = fp.como;
          size_t rings = 0;
          for (int side = 0; side < 2; side++) {
            ScanColDesc &d = side ? P.y : P.x;
            const int col = side ? fp.y : fp.x;
            if (kll_on_col[col] >= 0) {
              TGX_TRY(kll_scan_prepare(st, (size_t)kll_on_col[col], nrows, blocks * kWavesPerBlock,
                                       rows_per_wave(d, blocks), &d.kll, err));
              rings += ((size_t)1 << d.kll.top) + kTileRows;
            }
            chunk_bytes += (uint64_t)d.length * 8 + (d.validity ? (uint64_t)(d.length + 7) / 8 : 0);
          }
          lds = std::max(lds, (size_t)kWavesPerBlock * rings * sizeof(double));
          RL.cols[2 * k] = P.x;
          RL.cols[2 * k + 1] = P.y;
          RL.acc_index[2 * k] = P.x_acc;
          RL.acc_index[2 * k + 1] = P.y_acc;
          const tgx_column &xc = dev[fp.x], &yc = dev[fp.y];
          CL.pairs[k].x = xc.values;
          CL.pairs[k].y = yc.values;
          CL.pairs[k].xv = xc.validity;
          CL.pairs[k].yv = yc.validity;
          CL.pairs[k].xoff = xc.offset;
          CL.pairs[k].yoff = yc.offset;
          CL.pairs[k].length = P.x.length;
          CL.pairs[k].x_is_float = xc.type == TGX_FLOAT64;
          CL.pairs[k].y_is_float = yc.type == TGX_FLOAT64;
          CL.acc_index[k] = fp.como;
        }
        TGX_TRY(como_pivots(st, CL, n, err));
        HIP_TRY(st->d_scan_partials.reserve((size_t)2 * n * blocks * sizeof(ScanPartial)));
        HIP_TRY(st->d_como_partials.reserve((size_t)n * blocks * comoments_partial_bytes()));
        {
          ProfScope ps(st, "scan", chunk_bytes);
          launch_scan_pairs(PL, n, blocks, lds, st->d_scan_partials.as<ScanPartial>(), st->d_como_partials.p,
                            st->d_como_acc.as<ComomentAcc>(), st->stream);
        }
        launch_scan_reduce_only(RL, 2 * n, blocks, st->d_scan_partials.as<ScanPartial>(), st->d_scan_acc.as<ScanAcc>(),
                                st->stream);
        launch_comoments_reduce(CL, n, blocks, st->d_como_partials.p, st->d_como_acc.as<ComomentAcc>(), st->stream);
      }
    }
    // ---- validity-only columns ----
    {
      std::vector<CountColDesc> descs;
      std::vector<int32_t> index;
      uint64_t bytes = 0;
      int64_t max_words = 0;
      for (size_t s = 0; s < plan->count.size(); s++) {
        const tgx_column &c = dev[plan->count[s].column];
        if (c.type == TGX_DICT32_UTF8 && c.dictionary->validity && c.dictionary->length > 0) {
          // a row whose dictionary VALUE is NULL is a NULL row (Arrow's logical nulls): count through the indices
          launch_dict_count((const int32_t *)c.values, c.validity, c.offset, c.length, c.dictionary->validity,
                            c.dictionary->offset, c.dictionary->length, st->d_count_acc.as<CountAcc>() + s, g_ctx.n_cu,
                            st->stream);
          continue;
        }
        if (!c.validity) {  // no validity buffer: COUNT(col) = COUNT(*) = length, no kernel needed
          st->h_count[s].total += c.length;
          st->h_count[s].non_null += c.length;
          continue;
        }
        descs.push_back({c.validity, c.offset, c.length});
        index.push_back((int32_t)s);
        bytes += (uint64_t)(c.length + 7) / 8;
        max_words = std::max<int64_t>(max_words, (c.length + 63) / 64 + 1);
      }
      for (size_t c0 = 0; c0 < descs.size(); c0 += kMaxColsPerLaunch) {
        const int n = (int)std::min<size_t>(kMaxColsPerLaunch, descs.size() - c0);
        CountLaunch L;
        memset(&L, 0, sizeof(L));
        uint64_t chunk_bytes = 0;
        for (int k = 0; k < n; k++) {
          L.cols[k] = descs[c0 + k];
          L.acc_index[k] = index[c0 + k];
          chunk_bytes += (uint64_t)(L.cols[k].length + 7) / 8;
        }
        int blocks = (int)std::min<int64_t>(std::max<int64_t>(1, (max_words + 256 * 4 - 1) / (256 * 4)),
                                            std::max(8, (g_ctx.n_cu * 8) / n));
        HIP_TRY(st->d_count_blocks.reserve((size_t)n * blocks * sizeof(unsigned long long)));
        ProfScope ps(st, "count", chunk_bytes);
        launch_count(L, n, blocks, st->d_count_blocks.as<unsigned long long>(), st->d_count_acc.as<CountAcc>(), st->stream);
      }
      (void)bytes;
    }
    // ---- co-moments ----
    if (!plan->como.empty()) {
      std::vector<ComomentColDesc> descs;
      std::vector<int32_t> index;
      uint64_t bytes = 0;
      for (size_t s = 0; s < plan->como.size(); s++) {
        if (como_fused[s]) continue;  // rode on the scan of its columns
        const tgx_column &x = dev[plan->como[s].col_x], &y = dev[plan->como[s].col_y];
        if (!is_numeric(x.type) || !is_numeric(y.type))
          return fail(err, TGX_INVALID_ARGUMENT, "COMOMENTS needs numeric columns (%d, %d)", x.type, y.type);
        ComomentColDesc d;
        d.x = x.values;
        d.y = y.values;
        d.xv = x.validity;
        d.yv = y.validity;
        d.xoff = x.offset;
        d.yoff = y.offset;
        d.length = x.length;
        d.x_is_float = x.type == TGX_FLOAT64;
        d.y_is_float = y.type == TGX_FLOAT64;
        descs.push_back(d);
        index.push_back((int32_t)s);
        bytes += (uint64_t)x.length * 16 + (x.validity ? (uint64_t)(x.length + 7) / 8 : 0) +
                 (y.validity ? (uint64_t)(y.length + 7) / 8 : 0);
      }
      for (size_t c0 = 0; c0 < descs.size(); c0 += kMaxColsPerLaunch) {
        const int n = (int)std::min<size_t>(kMaxColsPerLaunch, descs.size() - c0);
        ComomentLaunch L;
        memset(&L, 0, sizeof(L));
        for (int k = 0; k < n; k++) {
          L.pairs[k] = descs[c0 + k];
          L.acc_index[k] = index[c0 + k];
        }
        int blocks = (int)std::min<int64_t>(std::max<int64_t>(1, (nrows + 256 * 16 - 1) / (256 * 16)),
                                            std::max(32, (g_ctx.n_cu * 12) / n));  // 4/5/6/8/12 per CU: 6.8/6.4/6.2/6.5/6.0 ms (2 pairs, 1 G rows)
        HIP_TRY(st->d_como_partials.reserve((size_t)n * blocks * comoments_partial_bytes()));
        TGX_TRY(como_pivots(st, L, n, err));
        ProfScope ps(st, "comoments", bytes * n / std::max<size_t>(descs.size(), 1));
        launch_comoments(L, n, blocks, st->d_como_partials.p, st->d_como_acc.as<ComomentAcc>(), st->stream);
      }
    }
    // ---- exact distinct ----
    // dictionary columns with a DISTINCT check and pattern / length checks: the patterns are matched on the
    // dictionary ENTRIES first (regex_update), their per-row gathers then ride on the DISTINCT pass
    DictFuse fuse;
    for (size_t s = 0; s < plan->distinct.size(); s++) {
      const DistinctTask &t = plan->distinct[s];
      if (!t.tuple.empty()) continue;
      const tgx_column &c = dev[t.column];
      if (c.type != TGX_DICT32_UTF8 || c.length == 0 || c.dictionary->length == 0 || c.dictionary->validity) continue;
      if (dict_usage_scratch_bytes(c.length, c.dictionary->length, t.multiplicity ? 1 : 0, g_ctx.n_cu) == 0) continue;
      const int cap = dict_fuse_capacity(c.length, c.dictionary->length, t.multiplicity ? 1 : 0, g_ctx.n_cu);
      if (cap > 0 && !fuse.capacity.count(t.column)) fuse.capacity[t.column] = cap;
    }
    TGX_TRY(regex_update(st, dev.data(), err, &fuse));
    std::map<int, bool> fuse_done;
    for (size_t s = 0; s < plan->distinct.size(); s++)
      if (distinct_idle(s) || distinct_done[s]) {
        continue;
      } else if (plan->distinct[s].tuple.empty()) {
        const int col = plan->distinct[s].column;
        const std::vector<DictGather> *g = nullptr;
        auto it = fuse.by_column.find(col);
        if (it != fuse.by_column.end() && !fuse_done[col]) {
          g = &it->second;
          fuse_done[col] = true;
        }
        const int stats_slot = (plan->distinct[s].scan_slot >= 0 && stats_by_partition[plan->distinct[s].scan_slot] == (int)s)
                                   ? plan->distinct[s].scan_slot
                                   : -1;
        TGX_TRY(distinct_update(st, s, dev[col], err, g, &dprep[s], stats_slot, &columns[col]));
      } else {
        TGX_TRY(distinct_tuple_update(st, s, dev.data(), err, columns));
      }
    // (gathers handed out but not consumed -- cannot happen: every fusable column has exactly one DISTINCT task)
    // ---- KLL ----
    TGX_TRY(kll_scan_finish(st, err));  // sketches the picks the scan left for the tasks that rode on it
    for (size_t s = 0; s < plan->kll.size(); s++)
      if (!kll_fused[s]) TGX_TRY(kll_update(st, s, dev[plan->kll[s].column], err));
    // ---- Spearman: keep the pairs, rank at finalize ----
    TGX_TRY(spearman_update(st, dev.data(), err));
  }
  st->batches++;
  if (arena_in_use) {
    // the arena (and its device twin) are free again once everything this update queued has run
    HIP_TRY(hipEventRecord(st->arena_event[st->arena_cur], st->stream));
    st->arena_busy[st->arena_cur] = true;
    st->arena_cur ^= 1;
    st->arena_used = 0;
  }
  // HOST buffers copied straight from the caller's memory are borrowed only until tgx_update returns
  if (any_host && st->host_direct) HIP_TRY(hipStreamSynchronize(st->stream));
  // a sampled-range key set keeps views of its batches for a later repair: staged copies of HOST batches do not
  // live that long
  if (any_host) TGX_TRY(distinct_resolve_all(st, err));
  return TGX_OK;
}

// ------------------------------------------------------------------------------------------------
// coalescing: small batches are noted, gathered into one column per flush, then take the ordinary pass
// (internal.h, Coalescer; kernels/gather.hip).  Reference shape: DataFusion's `batch_size: 8192`
// (TG/core/context.rs:28-38) -- what `execute_stream()` hands a drop-in.
constexpr int64_t kCoalesceFlushRows = 4 << 20;      // pending rows that trigger a flush
constexpr size_t kCoalesceFlushBatches = 4096;       // pending batches that trigger a flush
constexpr size_t kCoalesceArenaMax = 128u << 20;     // pinned staging per arena turn (HOST batches)

// The copy of a HOST batch's windows into the pinned arena is the only per-row work tgx_update does for a coalesced
// batch, and one core moves about 27 GB/s: a helper thread takes half of every batch's bytes (the calling thread the
// other half), which is what brings a stream of 8192-row batches near the PCIe rate.  The helper spins for a short
// while after a job -- batches of a stream arrive every few microseconds, a condition-variable wake-up costs more
// than a batch -- and then sleeps.  TGX_COPY_THREADS=0 keeps every copy on the calling thread.
static void stream_copy(void *dst, const void *src, size_t bytes);
namespace {
typedef tgx::CoalesceCopy CopyJob;
// K workers (TGX_COPY_THREADS, default 3), each with its own job slot.  A caller CLAIMS the workers that are idle at
// that moment (states fed from several threads at once -- a state per DataFusion partition stream -- share the pool:
// the first version gave the whole pool to one caller at a time and let the others copy alone at a core's 27 GB/s),
// cuts its batch's copies into (claimed + 1) shares, posts one to every claimed worker through its slot (no queue, no
// lock: a worker spins on its own ticket for ~200 us after its last job, then sleeps), copies its own share and waits
// for the others.  tgx_shutdown stops and joins the workers.
class CopyPool {
 public:
  static constexpr int kMaxWorkers = 8;
  static CopyPool *get() {
    if (CopyPool *fast = fast_instance().load(std::memory_order_acquire)) return fast->n_workers_ > 0 ? fast : nullptr;
    std::lock_guard<std::mutex> lock(instance_mu());
    CopyPool *h = fast_instance().load(std::memory_order_acquire);
    if (!h) {
      const char *e = getenv("TGX_COPY_THREADS");
      int k = e ? atoi(e) : 3;
      const unsigned hw = std::thread::hardware_concurrency();
      if (hw && (unsigned)k + 1 > hw) k = hw > 1 ? (int)hw - 1 : 0;  // (the caller copies a share as well)
      if (k > kMaxWorkers) k = kMaxWorkers;
      h = new CopyPool(k < 0 ? 0 : k);
      fast_instance().store(h, std::memory_order_release);
    }
    return h->n_workers_ > 0 ? h : nullptr;
  }
  static void shutdown() {  // tgx_shutdown: no state is being fed any more
    CopyPool *h = nullptr;
    {
      std::lock_guard<std::mutex> lock(instance_mu());
      h = fast_instance().exchange(nullptr, std::memory_order_acq_rel);
    }
    if (!h) return;
    h->stop_.store(true, std::memory_order_seq_cst);
    for (int w = 0; w < h->n_workers_; w++) {
      std::lock_guard<std::mutex> lock(h->w_[w].mu);
      h->w_[w].cv.notify_all();
    }
    for (auto &t : h->threads_) t.join();
    delete h;
  }
  // the idle workers, now this caller's until release(): ids[0 .. return value)
  int claim(int *ids) {
    int n = 0;
    for (int w = 0; w < n_workers_; w++)
      if (!w_[w].busy.exchange(true, std::memory_order_acquire)) ids[n++] = w;
    return n;
  }
  void post(int w, const CopyJob *jobs, size_t n) {  // (claimed)
    Worker &k = w_[w];
    k.jobs = jobs;
    k.n = n;
    k.ticket = k.posted.load(std::memory_order_relaxed) + 1;
    // Sequentially consistent on both sides (this store / the load of `sleeping` here, the store of `sleeping` / the
    // load of `posted` in the worker's wait): with release / acquire alone the load below may pass the store above,
    // find the worker awake, and the worker -- about to sleep -- may still find nothing posted: nobody wakes it and
    // the caller spins for ever (seen once in a few thousand HOST streams).
    k.posted.store(k.ticket, std::memory_order_seq_cst);
    if (k.sleeping.load(std::memory_order_seq_cst)) {
      std::lock_guard<std::mutex> lock(k.mu);
      k.cv.notify_one();
    }
  }
  void wait_and_release(int w) {
    Worker &k = w_[w];
    while (k.done.load(std::memory_order_acquire) != k.ticket) pause_or_nop();
    k.busy.store(false, std::memory_order_release);
  }
  void release(int w) { w_[w].busy.store(false, std::memory_order_release); }

 private:
  struct Worker {
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<uint64_t> posted{0}, done{0};
    std::atomic<bool> sleeping{false}, busy{false};
    const CopyJob *jobs = nullptr;
    size_t n = 0;
    uint64_t ticket = 0;
  };
  static std::mutex &instance_mu() {
    static std::mutex m;
    return m;
  }
  static std::atomic<CopyPool *> &fast_instance() {
    static std::atomic<CopyPool *> h{nullptr};
    return h;
  }
  static void pause_or_nop() {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  explicit CopyPool(int k) : n_workers_(k) {
    for (int w = 0; w < k; w++) threads_.emplace_back([this, w] { run(w_[w]); });
  }
  void run(Worker &k) {
    uint64_t seen = 0;
    for (;;) {
      // spin for about 200 us, then sleep until the next post (or the end)
      int spins = 0;
      while (k.posted.load(std::memory_order_acquire) == seen) {
        if (stop_.load(std::memory_order_relaxed)) return;
        pause_or_nop();
        if (++spins > 20000) {
          std::unique_lock<std::mutex> lock(k.mu);
          k.sleeping.store(true, std::memory_order_seq_cst);
          k.cv.wait(lock, [&] { return k.posted.load(std::memory_order_seq_cst) != seen || stop_.load(std::memory_order_seq_cst); });
          k.sleeping.store(false, std::memory_order_seq_cst);
          spins = 0;
        }
      }
      seen = k.posted.load(std::memory_order_acquire);
      for (size_t q = 0; q < k.n; q++) stream_copy(k.jobs[q].dst, k.jobs[q].src, k.jobs[q].bytes);
      k.done.store(seen, std::memory_order_release);
    }
  }
  std::atomic<bool> stop_{false};
  Worker w_[kMaxWorkers];
  std::vector<std::thread> threads_;
  const int n_workers_;
};
}  // namespace
static void copy_pool_shutdown() { CopyPool::shutdown(); }

// a copy that does not pull the destination into the cache first (the arena is written once and read by the DMA
// engine): glibc's memcpy takes its streaming path only for copies of several MiB
static void stream_copy(void *dst, const void *src, size_t bytes) {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)  // (this file also passes through the device compiler)
  if (bytes >= 4096 && ((uintptr_t)dst & 31) == 0) {
    typedef long long v4 __attribute__((vector_size(32), aligned(1)));
    typedef long long v4a __attribute__((vector_size(32)));
    const size_t n32 = bytes / 32;
    const v4 *s = (const v4 *)src;
    v4a *d = (v4a *)dst;
    for (size_t i = 0; i < n32; i++) __builtin_nontemporal_store((v4a)s[i], d + i);
    __builtin_ia32_sfence();
    const size_t done = n32 * 32;
    if (done < bytes) memcpy((char *)dst + done, (const char *)src + done, bytes - done);
    return;
  }
#endif
  memcpy(dst, src, bytes);
}

// MIN / MAX of the non-NULL values of an Int64 window (row 0 = bit `bit0` of *validity).  Runs on the thread that
// notes the batch, once per key column and batch: the plain loop is compiled a second time for AVX2 (64-bit
// compares), taken when the CPU has it.
#define TGX_MINMAX_BODY                                                        \
  int64_t mn = *lo, mx = *hi;                                                  \
  if (!validity) {                                                             \
    for (int64_t i = 0; i < n; i++) {                                          \
      mn = v[i] < mn ? v[i] : mn;                                              \
      mx = v[i] > mx ? v[i] : mx;                                              \
    }                                                                          \
  } else {                                                                     \
    int64_t i = 0;                                                             \
    for (; i < n && ((bit0 + i) & 7); i++) {                                   \
      const int64_t b = bit0 + i;                                              \
      if ((validity[b >> 3] >> (b & 7)) & 1) {                                 \
        mn = v[i] < mn ? v[i] : mn;                                            \
        mx = v[i] > mx ? v[i] : mx;                                            \
      }                                                                        \
    }                                                                          \
    for (; i + 8 <= n; i += 8) { /* a validity byte at a time: all-valid bytes take the branch-free loop */ \
      const uint8_t m = validity[(bit0 + i) >> 3];                             \
      if (m == 0xFF) {                                                         \
        for (int k = 0; k < 8; k++) {                                          \
          mn = v[i + k] < mn ? v[i + k] : mn;                                  \
          mx = v[i + k] > mx ? v[i + k] : mx;                                  \
        }                                                                      \
      } else {                                                                 \
        for (int k = 0; k < 8; k++)                                            \
          if ((m >> k) & 1) {                                                  \
            mn = v[i + k] < mn ? v[i + k] : mn;                                \
            mx = v[i + k] > mx ? v[i + k] : mx;                                \
          }                                                                    \
      }                                                                        \
    }                                                                          \
    for (; i < n; i++) {                                                       \
      const int64_t b = bit0 + i;                                              \
      if ((validity[b >> 3] >> (b & 7)) & 1) {                                 \
        mn = v[i] < mn ? v[i] : mn;                                            \
        mx = v[i] > mx ? v[i] : mx;                                            \
      }                                                                        \
    }                                                                          \
  }                                                                            \
  *lo = mn;                                                                    \
  *hi = mx;
static void host_minmax_i64_plain(const int64_t *v, const uint8_t *validity, int64_t bit0, int64_t n, int64_t *lo,
                                  int64_t *hi) {
  TGX_MINMAX_BODY
}
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
__attribute__((target("avx2"))) static void host_minmax_i64_avx2(const int64_t *v, const uint8_t *validity, int64_t bit0,
                                                                 int64_t n, int64_t *lo, int64_t *hi) {
  TGX_MINMAX_BODY
}
#endif
static void host_minmax_i64(const int64_t *v, const uint8_t *validity, int64_t bit0, int64_t n, int64_t *lo, int64_t *hi) {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) return host_minmax_i64_avx2(v, validity, bit0, n, lo, hi);
#endif
  host_minmax_i64_plain(v, validity, bit0, n, lo, hi);
}

static tgx_status coalesce_arena_ready(tgx_state *st, tgx_error *err) {
  Coalescer &co = st->coalesce;
  bind_thread();
  const int k = co.arena_cur;
  if (co.arena_busy[k]) {  // the flush that used this arena two turns ago (long done)
    HIP_TRY(hipEventSynchronize(co.arena_event[k]));
    co.arena_busy[k] = false;
  }
  if (co.arena_cap[k] < co.arena_want) {
    if (co.arena_host[k]) (void)hipHostFree(co.arena_host[k]);
    co.arena_host[k] = nullptr;
    co.arena_cap[k] = 0;
    HIP_TRY(hipHostMalloc(&co.arena_host[k], co.arena_want, hipHostMallocDefault));
    HIP_TRY(co.arena_dev[k].reserve(co.arena_want));
    co.arena_cap[k] = co.arena_want;
  }
  if (!co.arena_event[k]) HIP_TRY(hipEventCreateWithFlags(&co.arena_event[k], hipEventDisableTiming));
  return TGX_OK;
}

// What a Utf8View / dictionary window needs beyond its fixed-width part, found before anything is noted:
//   views:        the stretches of the variadic buffers the window's long views point into (a Parquet page's buffer is
//                 shared by the batches cut from it: only what THIS window references is copied) -- one walk over the
//                 window's views, NULL rows skipped (their views may hold anything);
//   dictionaries: whether the window brings a dictionary the column has not noted yet (batches of one file share
//                 theirs: it is taken once per flush).
struct WindowPrep {
  bool ok = true;  // false: more than kGatherViewBufs buffers referenced -- the batch takes the immediate path
  int32_t vb_count = 0;
  int32_t vb_index[kGatherViewBufs];
  int64_t vb_min[kGatherViewBufs], vb_end[kGatherViewBufs];
  bool new_dict = false;
  int64_t dict_first = 0, dict_end = 0;  // value bytes of the new dictionary's window
};
static tgx_status coalesce_prepare_window(const tgx_column &c, int64_t nrows, const CoalesceColumn &cc, int col,
                                          WindowPrep *w, tgx_error *err) {
  if (c.type == TGX_UTF8_VIEW) {
    const int32_t *v = (const int32_t *)c.values + (size_t)c.offset * 4;
    for (int64_t i = 0; i < nrows; i++, v += 4) {
      const int32_t len = v[0];
      if (len <= 12) continue;
      if (c.validity && !((c.validity[(c.offset + i) >> 3] >> ((c.offset + i) & 7)) & 1)) continue;
      const int32_t b = v[2];
      const int64_t off = v[3], end = off + len;
      if (b < 0 || b >= c.n_variadic || off < 0 || end > c.variadic_sizes[b])
        return fail(err, TGX_INVALID_ARGUMENT, "column %d: a view of row %lld points outside its data buffers", col, (long long)i);
      int k = 0;
      while (k < w->vb_count && w->vb_index[k] != b) k++;
      if (k == w->vb_count) {
        if (k == kGatherViewBufs) {
          w->ok = false;
          return TGX_OK;
        }
        w->vb_index[k] = b;
        w->vb_min[k] = off;
        w->vb_end[k] = end;
        w->vb_count++;
      } else {
        w->vb_min[k] = std::min(w->vb_min[k], off);
        w->vb_end[k] = std::max(w->vb_end[k], end);
      }
    }
  } else if (c.type == TGX_DICT32_UTF8) {
    const tgx_column &d = *c.dictionary;
    const CoalesceDict *cd = cc.dict.get();
    w->new_dict = !cd || cd->segs.empty() || cd->last_offsets != d.offsets || cd->last_data != d.data ||
                  cd->last_validity != d.validity || cd->last_offset != d.offset || cd->last_length != d.length ||
                  cd->type != d.type;
    if (w->new_dict && d.length > 0) {
      const size_t ow = d.type == TGX_UTF8 ? 4 : 8;
      const uint8_t *o0 = (const uint8_t *)d.offsets + (size_t)d.offset * ow;
      w->dict_first = ow == 4 ? (int64_t)((const int32_t *)o0)[0] : ((const int64_t *)o0)[0];
      w->dict_end = ow == 4 ? (int64_t)((const int32_t *)o0)[d.length] : ((const int64_t *)o0)[d.length];
      if (w->dict_end < w->dict_first) return fail(err, TGX_INVALID_ARGUMENT, "column %d: dictionary offsets decrease", col);
    }
    if (cd && !cd->segs.empty() && cd->type != d.type) w->ok = false;  // (Utf8 and LargeUtf8 dictionaries in one flush)
  }
  return TGX_OK;
}

// bytes one batch's HOST windows take in the arena (each buffer padded to 64 bytes)
static size_t coalesce_host_bytes(const tgx_plan *plan, const tgx_column *columns, int64_t nrows,
                                  const std::vector<WindowPrep> &prep) {
  size_t total = 0;
  for (int i = 0; i < plan->n_columns_needed; i++) {
    if (!plan->used[i] || columns[i].mem != TGX_MEM_HOST) continue;
    const tgx_column &c = columns[i];
    if (c.validity) total += (size_t)(((c.offset & 7) + nrows + 7) >> 3) + 64;
    if (c.type == TGX_UTF8_VIEW) {
      total += (size_t)nrows * 16 + 64;
      for (int k = 0; k < prep[i].vb_count; k++) total += (size_t)(prep[i].vb_end[k] - prep[i].vb_min[k]) + 64;
      continue;
    }
    if (c.type == TGX_DICT32_UTF8) {
      total += (size_t)nrows * 4 + 64;
      if (prep[i].new_dict) {
        const tgx_column &d = *c.dictionary;
        total += (size_t)(d.length + 1) * (d.type == TGX_UTF8 ? 4 : 8) + 64 + (size_t)(prep[i].dict_end - prep[i].dict_first) + 64;
        if (d.validity) total += (size_t)(((d.offset & 7) + d.length + 7) >> 3) + 64;
      }
      continue;
    }
    if (is_string(c.type)) {
      const size_t ow = c.type == TGX_UTF8 ? 4 : 8;
      const int64_t first = ow == 4 ? (int64_t)((const int32_t *)c.offsets)[c.offset] : ((const int64_t *)c.offsets)[c.offset];
      const int64_t end = ow == 4 ? (int64_t)((const int32_t *)c.offsets)[c.offset + nrows]
                                  : ((const int64_t *)c.offsets)[c.offset + nrows];
      total += (size_t)(nrows + 1) * ow + 64 + (size_t)std::max<int64_t>(end - first, 0) + 64;
    } else if (plan->reads_values[i] || c.values) {
      total += (size_t)nrows * (is_numeric32(c.type) ? 4 : 8) + 64;
    }
  }
  return total;
}

static tgx_status coalesce_append(const tgx_plan *plan, tgx_state *st, const tgx_column *columns, int64_t nrows,
                                  const BatchTraits &traits, bool *taken, tgx_error *err) {
  Coalescer &co = st->coalesce;
  *taken = false;
  if (co.cols.size() != (size_t)plan->n_columns_needed) {
    co.cols.resize(plan->n_columns_needed);
    for (auto &cc : co.cols) cc.segs.reserve(kCoalesceFlushBatches);
  }
  const bool any_host = traits.any_host;
  // (a flush in here empties the pending lists -- the dictionaries' too: what the windows bring is then looked at again)
  std::vector<WindowPrep> prep(plan->n_columns_needed);
  for (int attempt = 0;; attempt++) {
    const uint64_t flushes_before = co.flushes;
    for (int i = 0; i < plan->n_columns_needed; i++) {
      prep[i] = WindowPrep();
      if (!plan->used[i] || (columns[i].type != TGX_UTF8_VIEW && columns[i].type != TGX_DICT32_UTF8)) continue;
      TGX_TRY(coalesce_prepare_window(columns[i], nrows, co.cols[i], i, &prep[i], err));
      if (!prep[i].ok) return TGX_OK;  // (not taken: the immediate path)
    }
    if (any_host) {
      const size_t need = coalesce_host_bytes(plan, columns, nrows, prep);
      if (need > kCoalesceArenaMax) return TGX_OK;  // (64 Ki rows of very long strings): the immediate path
      if (co.arena_used > 0 && co.arena_used + need > co.arena_cap[co.arena_cur]) {
        // the arena is full: flush, and ask for a bigger one next time (fewer, larger flushes)
        co.arena_want = std::min(kCoalesceArenaMax, std::max(co.arena_want * 2, need));
        TGX_TRY(coalesce_flush(st, err));
      }
      if (need > co.arena_want) co.arena_want = std::min(kCoalesceArenaMax, need + need / 2);
      if (co.arena_used == 0) TGX_TRY(coalesce_arena_ready(st, err));  // this arena turn's first HOST window
      if (co.arena_used + need > co.arena_cap[co.arena_cur]) return TGX_OK;  // (cannot happen after the above)
    }
    // a string column whose coalesced int32 offsets would pass 2^31: flush first
    for (int i = 0; traits.any_utf8 && i < plan->n_columns_needed; i++) {
      if (!plan->used[i] || columns[i].type != TGX_UTF8) continue;
      const tgx_column &c = columns[i];
      const int64_t bytes = (int64_t)((const int32_t *)c.offsets)[c.offset + nrows] - (int64_t)((const int32_t *)c.offsets)[c.offset];
      if (co.cols[i].data_bytes + bytes > 0x7FFFFF00LL) {
        TGX_TRY(coalesce_flush(st, err));
        if (any_host) TGX_TRY(coalesce_arena_ready(st, err));  // (strings are HOST windows: the arena has just turned)
        break;
      }
    }
    // ... and a Utf8View column's one coalesced data buffer (int32 offsets in the views) likewise
    for (int i = 0; i < plan->n_columns_needed; i++) {
      if (!plan->used[i] || columns[i].type != TGX_UTF8_VIEW) continue;
      int64_t bytes = 0;
      for (int k = 0; k < prep[i].vb_count; k++) bytes += prep[i].vb_end[k] - prep[i].vb_min[k] + 16;
      if (co.cols[i].data_bytes + bytes > 0x7FFFFF00LL) {
        TGX_TRY(coalesce_flush(st, err));
        if (any_host) TGX_TRY(coalesce_arena_ready(st, err));
        break;
      }
    }
    if (co.flushes == flushes_before) break;  // nothing was flushed: `prep` describes what is pending
    if (attempt >= 2) return TGX_OK;          // (cannot happen: after a flush nothing is pending)
  }
  // everything that can refuse the batch is checked BEFORE the first column notes its window: a batch is noted for all
  // columns or for none (a column with one segment more than its neighbours would make the next flush's gather write
  // past the coalesced buffers)
  for (int i = 0; i < plan->n_columns_needed; i++) {
    if (!plan->used[i] || !is_string(columns[i].type)) continue;
    const tgx_column &c = columns[i];
    const size_t ow = c.type == TGX_UTF8 ? 4 : 8;
    const uint8_t *o0 = (const uint8_t *)c.offsets + (size_t)c.offset * ow;
    const int64_t first = ow == 4 ? (int64_t)((const int32_t *)o0)[0] : ((const int64_t *)o0)[0];
    const int64_t end = ow == 4 ? (int64_t)((const int32_t *)o0)[nrows] : ((const int64_t *)o0)[nrows];
    if (end < first) return fail(err, TGX_INVALID_ARGUMENT, "column %d: offsets decrease", i);
  }
  struct Rollback {  // (a host allocation that throws while the windows are noted)
    Coalescer &co;
    std::vector<size_t> segs, dict_segs;
    std::vector<int64_t> data_bytes;
    size_t arena_used;
    bool armed = true;
    explicit Rollback(Coalescer &c) : co(c), arena_used(c.arena_used) {
      for (auto &cc : co.cols) {
        segs.push_back(cc.segs.size());
        dict_segs.push_back(cc.dict ? cc.dict->segs.size() : 0);
        data_bytes.push_back(cc.data_bytes);
      }
    }
    ~Rollback() {
      if (!armed) return;
      for (size_t i = 0; i < co.cols.size(); i++) {
        co.cols[i].segs.resize(segs[i]);
        co.cols[i].data_bytes = data_bytes[i];
        co.cols[i].range_known = false;  // (a MIN / MAX of rows that are not pending after all is only too wide)
        if (co.cols[i].dict && co.cols[i].dict->segs.size() > dict_segs[i]) {
          // (entries / data_bytes of the dropped dictionary stay counted: the buffers are only sized too generously;
          //  forgetting "the last dictionary" makes the next window bring its own again)
          co.cols[i].dict->segs.resize(dict_segs[i]);
          co.cols[i].dict->last_length = -1;
        }
      }
      co.arena_used = arena_used;
    }
  } rollback(co);
  // ... and a Utf8View column's one coalesced data buffer (int32 offsets in the views) likewise
  for (int i = 0; i < plan->n_columns_needed; i++) {
    if (!plan->used[i] || columns[i].type != TGX_UTF8_VIEW) continue;
    int64_t bytes = 0;
    for (int k = 0; k < prep[i].vb_count; k++) bytes += prep[i].vb_end[k] - prep[i].vb_min[k] + 16;
    if (co.cols[i].data_bytes + bytes > 0x7FFFFF00LL) {
      TGX_TRY(coalesce_flush(st, err));
      if (any_host) TGX_TRY(coalesce_arena_ready(st, err));
      // (the flush has emptied every column's dictionary list: what the windows bring is new again)
      for (int j = 0; j < plan->n_columns_needed; j++)
        if (plan->used[j] && columns[j].type == TGX_DICT32_UTF8) {
          prep[j] = WindowPrep();
          TGX_TRY(coalesce_prepare_window(columns[j], nrows, co.cols[j], j, &prep[j], err));
        }
      break;
    }
  }
  char *ah = any_host ? (char *)co.arena_host[co.arena_cur] : nullptr;
  const char *ad = any_host ? (const char *)co.arena_dev[co.arena_cur].p : nullptr;
  std::vector<CopyJob> &jobs = co.copy_jobs;
  jobs.clear();
  auto to_arena = [&](const void *src, size_t bytes) -> const void * {  // returns the DEVICE twin's address
    const size_t at = (co.arena_used + 63) & ~(size_t)63;
    jobs.push_back({ah + at, src, bytes});  // (copied below, half of the bytes by the helper thread)
    co.arena_used = at + bytes;
    return ad + at;
  };
  for (int i = 0; i < plan->n_columns_needed; i++) {
    if (!plan->used[i]) continue;
    const tgx_column &c = columns[i];
    CoalesceColumn &cc = co.cols[i];
    cc.type = c.type;
    CoalesceSegment sg;
    memset(&sg, 0, sizeof(sg));
    sg.length = nrows;
    const bool host = c.mem == TGX_MEM_HOST;
    if (c.validity) {
      const uint8_t *v0 = c.validity + (c.offset >> 3);
      sg.bit0 = c.offset & 7;
      sg.validity = host ? (const uint8_t *)to_arena(v0, (size_t)((sg.bit0 + nrows + 7) >> 3)) : v0;
      cc.any_validity = true;
    }
    if (c.type == TGX_UTF8_VIEW) {
      // the views as they are, and the stretches of the data buffers they point into (host is true: update_validate)
      sg.values = to_arena((const uint8_t *)c.values + (size_t)c.offset * 16, (size_t)nrows * 16);
      sg.vb_count = prep[i].vb_count;
      for (int k = 0; k < prep[i].vb_count; k++) {
        sg.vb_index[k] = prep[i].vb_index[k];
        sg.vb_min[k] = prep[i].vb_min[k];
        sg.vb_len[k] = prep[i].vb_end[k] - prep[i].vb_min[k];
        sg.vb_src[k] = (const uint8_t *)to_arena(c.variadic[prep[i].vb_index[k]] + prep[i].vb_min[k], (size_t)sg.vb_len[k]);
        sg.data_len += (sg.vb_len[k] + 15) & ~(int64_t)15;  // (every stretch lands 16-byte aligned)
      }
      cc.data_bytes += sg.data_len;
    } else if (c.type == TGX_DICT32_UTF8) {
      if (!cc.dict) cc.dict.reset(new CoalesceDict());
      CoalesceDict &cd = *cc.dict;
      const tgx_column &d = *c.dictionary;
      if (prep[i].new_dict) {
        CoalesceSegment ds;
        memset(&ds, 0, sizeof(ds));
        ds.length = d.length;
        cd.type = d.type;
        if (d.validity && d.length > 0) {
          ds.bit0 = d.offset & 7;
          ds.validity = (const uint8_t *)to_arena(d.validity + (d.offset >> 3), (size_t)((ds.bit0 + d.length + 7) >> 3));
          cd.any_validity = true;
        }
        if (d.length > 0) {
          const size_t ow = d.type == TGX_UTF8 ? 4 : 8;
          ds.data_first = prep[i].dict_first;
          ds.data_len = prep[i].dict_end - prep[i].dict_first;
          ds.values = to_arena((const uint8_t *)d.offsets + (size_t)d.offset * ow, (size_t)(d.length + 1) * ow);
          ds.data = (ds.data_len > 0 && d.data) ? (const uint8_t *)to_arena(d.data + ds.data_first, (size_t)ds.data_len) : nullptr;
        }
        cd.last_base = cd.entries;
        cd.entries += d.length;
        cd.data_bytes += ds.data_len;
        cd.last_offsets = d.offsets;
        cd.last_data = d.data;
        cd.last_validity = d.validity;
        cd.last_offset = d.offset;
        cd.last_length = d.length;
        cd.segs.push_back(ds);
      }
      sg.values = to_arena((const uint8_t *)c.values + (size_t)c.offset * 4, (size_t)nrows * 4);
      sg.index_shift = (int32_t)cd.last_base;
    } else if (is_string(c.type)) {
      const size_t ow = c.type == TGX_UTF8 ? 4 : 8;
      const uint8_t *o0 = (const uint8_t *)c.offsets + (size_t)c.offset * ow;
      const int64_t first = ow == 4 ? (int64_t)((const int32_t *)o0)[0] : ((const int64_t *)o0)[0];
      const int64_t end = ow == 4 ? (int64_t)((const int32_t *)o0)[nrows] : ((const int64_t *)o0)[nrows];
      sg.data_first = first;
      sg.data_len = end - first;
      sg.values = to_arena(o0, (size_t)(nrows + 1) * ow);
      sg.data = (sg.data_len > 0 && c.data) ? (const uint8_t *)to_arena(c.data + first, (size_t)sg.data_len) : nullptr;
      cc.data_bytes += sg.data_len;
    } else if (c.values) {
      const size_t ew = is_numeric32(c.type) ? 4 : 8;
      const uint8_t *v0 = (const uint8_t *)c.values + (size_t)c.offset * ew;
      sg.values = host ? to_arena(v0, (size_t)nrows * ew) : (const void *)v0;
      if (plan->key_column[i] && c.type == TGX_INT64) {
        if (host && cc.range_known)
          host_minmax_i64((const int64_t *)v0, c.validity ? c.validity + (c.offset >> 3) : nullptr, c.offset & 7, nrows,
                          &cc.range_lo, &cc.range_hi);
        else
          cc.range_known = false;
      }
    }
    cc.segs.push_back(sg);
  }
  if (!jobs.empty()) {
    size_t total = 0;
    for (const CopyJob &j : jobs) total += j.bytes;
    CopyPool *pool = total >= (64u << 10) ? CopyPool::get() : nullptr;
    int ids[CopyPool::kMaxWorkers];
    const int helpers = pool ? pool->claim(ids) : 0;  // (whoever is idle right now: other states may hold the rest)
    if (helpers == 0) {
      for (const CopyJob &j : jobs) stream_copy(j.dst, j.src, j.bytes);
    } else {
      // equal shares of the bytes (a job that straddles a boundary is cut at a multiple of 64 bytes): the first for
      // the caller, one for every claimed worker
      const int shares = helpers + 1;
      std::vector<CopyJob> &cut = co.copy_tail;  // all shares one behind the other; first[s] = where share s begins
      cut.clear();
      size_t first[CopyPool::kMaxWorkers + 2];
      const size_t per = (total / (size_t)shares + 63) & ~(size_t)63;
      size_t room = per;
      int share = 0;
      first[0] = 0;
      for (const CopyJob &j : jobs) {
        size_t at = 0;
        while (at < j.bytes) {
          if (room == 0 && share + 1 < shares) {
            first[++share] = cut.size();
            room = per;
          }
          size_t take = share + 1 < shares ? std::min(room, j.bytes - at) : j.bytes - at;
          if (take < j.bytes - at) take = std::min((take + 63) & ~(size_t)63, j.bytes - at);  // (cuts stay 64-byte aligned)
          cut.push_back({(char *)j.dst + at, (const char *)j.src + at, take});
          at += take;
          room -= std::min(room, take);
        }
      }
      while (share + 1 < shares) first[++share] = cut.size();
      first[shares] = cut.size();
      for (int w = 0; w < helpers; w++) pool->post(ids[w], cut.data() + first[w + 1], first[w + 2] - first[w + 1]);
      for (size_t q = first[0]; q < first[1]; q++) stream_copy(cut[q].dst, cut[q].src, cut[q].bytes);
      for (int w = 0; w < helpers; w++) pool->wait_and_release(ids[w]);
    }
  }
  rollback.armed = false;
  co.rows += nrows;
  co.batches += 1;
  co.coalesced_batches += 1;
  st->batches++;
  *taken = true;
  if (co.rows >= (co.flush_rows > 0 ? co.flush_rows : kCoalesceFlushRows) || co.batches >= kCoalesceFlushBatches)
    return coalesce_flush(st, err);
  return TGX_OK;
}

// Region set `set` is about to be overwritten: views retained into it (a sampled-range key set keeps its batches for
// a later repair) are dropped when the counters snapshot taken after the flush that filled it shows nothing to repair;
// otherwise the repair runs now.
static tgx_status coalesce_release_set(tgx_state *st, int set, tgx_error *err) {
  Coalescer &co = st->coalesce;
  bool any = false;
  for (auto &ds : st->distinct)
    for (size_t k = 0; k < ds.retained.size(); k++) any |= ds.retained.region_set[k] == set;
  if (!any) {
    co.snap_pending[set] = false;
    return TGX_OK;
  }
  bool repair = !co.snap_pending[set];
  if (co.snap_pending[set]) {
    HIP_TRY(hipEventSynchronize(co.snap_event[set]));  // (recorded two flushes ago)
    co.snap_pending[set] = false;
    const unsigned long long *snap = (const unsigned long long *)co.snap_host[set];
    for (size_t q = 0; q < st->distinct.size(); q++) {
      DistinctState &ds = st->distinct[q];
      bool tagged = false;
      for (size_t k = 0; k < ds.retained.size(); k++) tagged |= ds.retained.region_set[k] == set;
      if (!tagged) continue;
      if (snap[q * kNumDistinctCounters + kCntOutOfRange] != 0) {
        repair = true;
        continue;
      }
      // nothing outside the bitmap / no overflowed list as of the end of that flush: its batches hold nothing to repair
      size_t w = 0;
      for (size_t k = 0; k < ds.retained.size(); k++)
        if (ds.retained.region_set[k] != set) {
          ds.retained.cols[w] = ds.retained.cols[k];
          ds.retained.region_set[w++] = ds.retained.region_set[k];
        }
      ds.retained.cols.resize(w);
      ds.retained.region_set.resize(w);
    }
  }
  if (repair) TGX_TRY(distinct_resolve_all(st, err));
  return TGX_OK;
}

tgx_status tgx::coalesce_flush(tgx_state *st, tgx_error *err) {
  Coalescer &co = st->coalesce;
  if (co.rows == 0 || co.flushing) return TGX_OK;
  const tgx_plan *plan = st->plan;
  bind_thread();
  TGX_TRY(state_init_device(st, err));
  struct Guard {
    Coalescer &c;
    ~Guard() { c.flushing = false; }
  } guard{co};
  co.flushing = true;
  const int set = co.set_cur, ar = co.arena_cur;
  const int64_t rows = co.rows;
  TGX_TRY(coalesce_release_set(st, set, err));
  // the segment table: pinned, one turn per arena
  size_t n_segs = 0;
  for (int i = 0; i < plan->n_columns_needed; i++)
    if (plan->used[i]) n_segs += co.cols[i].segs.size() + (co.cols[i].dict ? co.cols[i].dict->segs.size() : 0);
  if (!co.arena_event[ar]) HIP_TRY(hipEventCreateWithFlags(&co.arena_event[ar], hipEventDisableTiming));
  if (co.arena_busy[ar]) {  // (DEVICE-only batches never went through coalesce_arena_ready)
    HIP_TRY(hipEventSynchronize(co.arena_event[ar]));
    co.arena_busy[ar] = false;
  }
  if (co.desc_cap[ar] < n_segs * sizeof(GatherSeg)) {
    if (co.desc_host[ar]) (void)hipHostFree(co.desc_host[ar]);
    co.desc_host[ar] = nullptr;
    co.desc_cap[ar] = 0;
    const size_t want = std::max<size_t>(2 * n_segs * sizeof(GatherSeg), 64u << 10);
    HIP_TRY(hipHostMalloc(&co.desc_host[ar], want, hipHostMallocDefault));
    co.desc_cap[ar] = want;
  }
  HIP_TRY(co.desc_dev[ar].reserve(co.desc_cap[ar]));
  GatherSeg *gs = (GatherSeg *)co.desc_host[ar];
  size_t g = 0;
  std::vector<tgx_column> views(plan->n_columns_needed);
  for (int i = 0; i < plan->n_columns_needed; i++) {
    tgx_column &v = views[i];
    memset(&v, 0, sizeof(v));
    if (!plan->used[i]) continue;
    CoalesceColumn &cc = co.cols[i];
    const bool str = is_string(cc.type), vw = cc.type == TGX_UTF8_VIEW, dct = cc.type == TGX_DICT32_UTF8;
    const size_t ew = str ? (cc.type == TGX_UTF8 ? 4 : 8) : vw ? 16 : (dct || is_numeric32(cc.type)) ? 4 : 8;
    bool has_values = false;
    for (const CoalesceSegment &sg : cc.segs) has_values |= sg.values != nullptr;
    if (has_values) HIP_TRY(cc.values[set].reserve((size_t)(rows + 1) * ew + 64));
    if (cc.any_validity) {
      const size_t vb = ((size_t)rows + 31) / 32 * 4 + 64;
      HIP_TRY(cc.validity[set].reserve(vb));
      HIP_TRY(hipMemsetAsync(cc.validity[set].p, 0, vb, st->stream));
    }
    if (str || vw) HIP_TRY(cc.data[set].reserve((size_t)cc.data_bytes + 64));
    int64_t row = 0, data_at = 0;
    for (const CoalesceSegment &sg : cc.segs) {
      GatherSeg &d = gs[g++];
      memset(&d, 0, sizeof(d));
      d.src_values = sg.values;
      d.src_validity = sg.validity;
      d.src_data = sg.data;
      d.dst_values = has_values ? cc.values[set].p : nullptr;
      d.dst_validity = cc.any_validity ? cc.validity[set].as<uint8_t>() : nullptr;
      d.dst_data = (str || vw) ? cc.data[set].as<uint8_t>() : nullptr;
      d.src_bit0 = sg.bit0;
      d.length = sg.length;
      d.dst_row = row;
      d.data_first = sg.data_first;
      d.data_base = data_at;
      d.data_len = sg.data ? sg.data_len : 0;
      d.elem_bytes = (int32_t)ew;
      d.kind = str ? (cc.type == TGX_UTF8 ? 1 : 2) : vw ? 3 : dct ? 4 : 0;
      if (vw) {  // the window's stretches one behind the other, each 16-byte aligned
        d.vb_count = sg.vb_count;
        int64_t at = data_at;
        for (int k = 0; k < sg.vb_count; k++) {
          d.vb_index[k] = sg.vb_index[k];
          d.vb_min[k] = sg.vb_min[k];
          d.vb_len[k] = sg.vb_len[k];
          d.vb_src[k] = sg.vb_src[k];
          d.vb_base[k] = at;
          at += (sg.vb_len[k] + 15) & ~(int64_t)15;
        }
      }
      d.index_shift = sg.index_shift;
      row += sg.length;
      data_at += sg.data_len;
    }
    v.type = cc.type;
    v.mem = TGX_MEM_DEVICE;
    v.length = rows;
    v.offset = 0;
    v.null_count = -1;
    v.validity = cc.any_validity ? cc.validity[set].as<uint8_t>() : nullptr;
    if (str) {
      v.offsets = cc.values[set].p;
      v.data = cc.data[set].as<uint8_t>();
    } else if (vw) {  // a Utf8View column with ONE data buffer
      v.values = cc.values[set].p;
      cc.view_buf[set] = cc.data[set].as<uint8_t>();
      v.variadic = &cc.view_buf[set];
      v.n_variadic = 1;
    } else {
      v.values = has_values ? cc.values[set].p : nullptr;
    }
    if (dct) {  // the windows' dictionaries, gathered like a Utf8 column of their own
      CoalesceDict &cd = *cc.dict;
      const size_t dw = cd.type == TGX_UTF8 ? 4 : 8;
      HIP_TRY(cd.values[set].reserve((size_t)(cd.entries + 1) * dw + 64));
      HIP_TRY(hipMemsetAsync(cd.values[set].p, 0, (size_t)(cd.entries + 1) * dw, st->stream));  // (an empty dictionary: offset 0)
      if (cd.any_validity) {
        const size_t vb = ((size_t)cd.entries + 31) / 32 * 4 + 64;
        HIP_TRY(cd.validity[set].reserve(vb));
        HIP_TRY(hipMemsetAsync(cd.validity[set].p, 0, vb, st->stream));
      }
      HIP_TRY(cd.data[set].reserve((size_t)cd.data_bytes + 64));
      int64_t drow = 0, dat = 0;
      for (const CoalesceSegment &sg : cd.segs) {
        GatherSeg &d = gs[g++];
        memset(&d, 0, sizeof(d));
        d.src_values = sg.values;
        d.src_validity = sg.validity;
        d.src_data = sg.data;
        d.dst_values = cd.values[set].p;
        d.dst_validity = cd.any_validity ? cd.validity[set].as<uint8_t>() : nullptr;
        d.dst_data = cd.data[set].as<uint8_t>();
        d.src_bit0 = sg.bit0;
        d.length = sg.length;
        d.dst_row = drow;
        d.data_first = sg.data_first;
        d.data_base = dat;
        d.data_len = sg.data ? sg.data_len : 0;
        d.elem_bytes = (int32_t)dw;
        d.kind = cd.type == TGX_UTF8 ? 1 : 2;
        drow += sg.length;
        dat += sg.data_len;
      }
      tgx_column &dv = cd.view[set];
      memset(&dv, 0, sizeof(dv));
      dv.type = cd.type;
      dv.mem = TGX_MEM_DEVICE;
      dv.length = cd.entries;
      dv.null_count = -1;
      dv.validity = cd.any_validity ? cd.validity[set].as<uint8_t>() : nullptr;
      dv.offsets = cd.values[set].p;
      dv.data = cd.data[set].as<uint8_t>();
      v.dictionary = &dv;
    }
  }
  if (co.arena_used)
    HIP_TRY(hipMemcpyAsync(co.arena_dev[ar].p, co.arena_host[ar], co.arena_used, hipMemcpyHostToDevice, st->stream));
  HIP_TRY(hipMemcpyAsync(co.desc_dev[ar].p, gs, g * sizeof(GatherSeg), hipMemcpyHostToDevice, st->stream));
  {
    ProfScope ps(st, "gather", 0);
    launch_gather_segments(co.desc_dev[ar].as<GatherSeg>(), (int)g, st->stream);
  }
  // the arena and the table are free again once the gather has run
  HIP_TRY(hipEventRecord(co.arena_event[ar], st->stream));
  co.arena_busy[ar] = true;
  co.arena_cur ^= 1;
  co.arena_used = 0;
  // Int64 key columns whose pending windows were all HOST: the flush's value range is known exactly
  for (size_t q = 0; q < plan->distinct.size(); q++) {
    const DistinctTask &t = plan->distinct[q];
    DistinctState &ds = st->distinct[q];
    ds.batch_range_known = false;
    if (!t.tuple.empty() || t.approx_only) continue;
    const CoalesceColumn &cc = co.cols[t.column];
    ds.flush_device_keys = false;
    if (cc.type == TGX_INT64 && cc.range_known && cc.range_lo <= cc.range_hi && !cc.segs.empty()) {
      ds.batch_range_known = true;
      ds.batch_lo = cc.range_lo;
      ds.batch_hi = cc.range_hi;
    } else if (cc.type == TGX_INT64 && !cc.range_known && !cc.segs.empty()) {
      ds.flush_device_keys = true;  // (DEVICE windows: the device will say, distinct_sample_all)
    }
  }
  // the pending list is empty from here on (update_impl may come back to tgx::coalesce_flush through a resolve)
  for (auto &cc : co.cols) {
    cc.segs.clear();
    cc.any_validity = false;
    cc.data_bytes = 0;
    cc.range_known = true;
    cc.range_lo = INT64_MAX;
    cc.range_hi = INT64_MIN;
    if (cc.dict) {
      cc.dict->segs.clear();
      cc.dict->any_validity = false;
      cc.dict->data_bytes = cc.dict->entries = 0;
      cc.dict->last_length = -1;
    }
  }
  const int64_t batches_of_flush = (int64_t)co.batches;
  co.rows = 0;
  co.batches = 0;
  co.set_cur ^= 1;
  co.flushes++;
  std::vector<size_t> kept_before(st->distinct.size());
  for (size_t q = 0; q < st->distinct.size(); q++) kept_before[q] = st->distinct[q].retained.size();
  st->batches -= batches_of_flush;  // update_impl counts the flush as one batch: keep the caller's count
  tgx_status rc = update_impl(plan, st, views.data(), rows, err);
  st->batches += batches_of_flush - 1;
  for (auto &ds : st->distinct) {
    ds.batch_range_known = false;
    ds.flush_device_keys = false;
  }
  // views the key sets kept of this flush point into region set `set` -- also when the pass failed half-way: a view
  // that kept the tag of "the caller's memory" would dangle once the set is used again
  for (size_t q = 0; q < st->distinct.size(); q++) {
    DistinctState::Retained &r = st->distinct[q].retained;
    for (size_t k = std::min(kept_before[q], r.size()); k < r.size(); k++) r.region_set[k] = (int8_t)set;
  }
  if (rc != TGX_OK) return rc;
  bool any_kept = false;
  for (auto &ds : st->distinct) any_kept |= !ds.retained.empty();
  if (any_kept && st->d_distinct_counters.p) {
    const size_t bytes = st->distinct.size() * kNumDistinctCounters * sizeof(unsigned long long);
    if (co.snap_cap[set] < bytes) {
      if (co.snap_host[set]) (void)hipHostFree(co.snap_host[set]);
      co.snap_host[set] = nullptr;
      HIP_TRY(hipHostMalloc(&co.snap_host[set], bytes + 256, hipHostMallocDefault));
      co.snap_cap[set] = bytes + 256;
    }
    if (!co.snap_event[set]) HIP_TRY(hipEventCreateWithFlags(&co.snap_event[set], hipEventDisableTiming));
    HIP_TRY(hipMemcpyAsync(co.snap_host[set], st->d_distinct_counters.p, bytes, hipMemcpyDeviceToHost, st->stream));
    HIP_TRY(hipEventRecord(co.snap_event[set], st->stream));
    co.snap_pending[set] = true;
  }
  return TGX_OK;
}

static void coalesce_drop(tgx_state *st) {  // reset / destroy: pending batches are forgotten
  Coalescer &co = st->coalesce;
  for (auto &cc : co.cols) {
    cc.segs.clear();
    cc.any_validity = false;
    cc.data_bytes = 0;
    cc.range_known = true;
    cc.range_lo = INT64_MAX;
    cc.range_hi = INT64_MIN;
    if (cc.dict) {
      cc.dict->segs.clear();
      cc.dict->any_validity = false;
      cc.dict->data_bytes = cc.dict->entries = 0;
      cc.dict->last_length = -1;
    }
  }
  co.rows = 0;
  co.batches = 0;
  co.arena_used = 0;
  co.snap_pending[0] = co.snap_pending[1] = false;
}

// ------------------------------------------------------------------------------------------------
// gather: the merged (device + host) view of a state
namespace {
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
}  // namespace

// `pre`: the counters of all tasks, already read back in one copy (gather); nullptr: read this task's now
static tgx_status distinct_totals(tgx_state *st, size_t slot, DistinctTotals *t, tgx_error *err,
                                  const unsigned long long *pre = nullptr) {
  DistinctState &ds = st->distinct[slot];
  unsigned long long c[kNumDistinctCounters];
  memset(c, 0, sizeof(c));
  if (pre)
    memcpy(c, pre + slot * kNumDistinctCounters, sizeof(c));
  else if (st->device_ready)
    TGX_TRY(distinct_read_counters(st, ds, c, err));
  if (c[kCntOutOfRange] != 0)
    return fail(err, TGX_INTERNAL, "distinct: %llu keys fell outside the range bitmap", c[kCntOutOfRange]);
  const uint64_t empty_rows = c[kCntEmptyRows] + ds.h_empty_rows;
  t->total = (uint64_t)ds.total_rows + ds.h_total;
  t->non_null = c[kCntValidRows] + ds.h_non_null;
  t->distinct = c[kCntDistinct] + ds.h_distinct + (empty_rows > 0 ? 1 : 0);
  t->twice = c[kCntTwice] + ds.h_twice + (empty_rows > 1 ? 1 : 0);
  t->empty_rows = empty_rows;
  return TGX_OK;
}

static tgx_status gather(tgx_state *st, Gathered *g, tgx_error *err) {
  const tgx_plan *plan = st->plan;
  TGX_TRY(coalesce_flush(st, err));  // batches tgx_update has only noted so far
  g->scan = st->h_scan;
  g->count = st->h_count;
  g->como = st->h_como;
  g->hll = st->h_hll;
  g->distinct.resize(plan->distinct.size());
  std::vector<unsigned long long> all;
  if (st->device_ready) {
    // every accumulator comes back with copies ordered on the state's OWN stream and one synchronisation of it: a
    // synchronous hipMemcpy (null stream) also waits for the other streams of the device, e.g. another state's scan
    // ... and into PINNED memory: the four copies are queued together (into pageable memory each is staged and
    // waited for before the next is issued: 25 us between two of them)
    std::vector<ScanAcc> d_scan(plan->scan.size());
    std::vector<CountAcc> d_count(plan->count.size());
    std::vector<ComomentAcc> d_como(plan->como.size());
    if (st->d_distinct_counters.p) all.resize(plan->distinct.size() * kNumDistinctCounters);
    const size_t b_scan = d_scan.size() * sizeof(ScanAcc), b_count = d_count.size() * sizeof(CountAcc),
                 b_como = d_como.size() * sizeof(ComomentAcc), b_all = all.size() * sizeof(unsigned long long);
    const size_t b_hll = st->d_hll.p ? plan->hll.size() * (size_t)kHllRegisters : 0;
    TGX_TRY(pinned_readback(st, b_scan + b_count + b_como + b_all + b_hll + 64, err));
    char *h = (char *)st->h_pinned;
    if (b_scan) HIP_TRY(hipMemcpyAsync(h, st->d_scan_acc.p, b_scan, hipMemcpyDeviceToHost, st->stream));
    if (b_count) HIP_TRY(hipMemcpyAsync(h + b_scan, st->d_count_acc.p, b_count, hipMemcpyDeviceToHost, st->stream));
    if (b_como)
      HIP_TRY(hipMemcpyAsync(h + b_scan + b_count, st->d_como_acc.p, b_como, hipMemcpyDeviceToHost, st->stream));
    if (b_all)
      HIP_TRY(hipMemcpyAsync(h + b_scan + b_count + b_como, st->d_distinct_counters.p, b_all, hipMemcpyDeviceToHost,
                             st->stream));
    if (b_hll)
      HIP_TRY(hipMemcpyAsync(h + b_scan + b_count + b_como + b_all, st->d_hll.p, b_hll, hipMemcpyDeviceToHost, st->stream));
    HIP_TRY(hipStreamSynchronize(st->stream));
    for (size_t q = 0; b_hll && q < plan->hll.size(); q++) {
      if (st->hll_mode[q] != 1) continue;
      const uint8_t *regs = (const uint8_t *)h + b_scan + b_count + b_como + b_all + q * (size_t)kHllRegisters;
      std::vector<uint8_t> &out = g->hll[q];
      if (out.empty()) {
        out.assign(regs, regs + kHllRegisters);
      } else {
        for (int r = 0; r < kHllRegisters; r++) out[r] = std::max(out[r], regs[r]);
      }
    }
    if (b_scan) memcpy(d_scan.data(), h, b_scan);
    if (b_count) memcpy(d_count.data(), h + b_scan, b_count);
    if (b_como) memcpy(d_como.data(), h + b_scan + b_count, b_como);
    if (b_all) memcpy(all.data(), h + b_scan + b_count + b_como, b_all);
    st->ptr_tables.clear();
    st->parked.clear();  // (the stream has just been drained)
    // keys that fell outside a sampled bitmap range (DistinctState::speculative): the counters just read say whether
    // any task has some -- only then is there a repair to run and its counters to read again
    bool repaired = false;
    for (size_t k = 0; k < st->distinct.size(); k++) {
      DistinctState &ds = st->distinct[k];
      if ((ds.speculative || ds.fp_staged) && !ds.retained.empty() && !all.empty() &&
          all[k * kNumDistinctCounters + kCntOutOfRange] != 0) {
        TGX_TRY(distinct_resolve(st, k, err));
        repaired = true;
      } else {
        ds.retained.clear();
        if (!all.empty()) ds.outliers_possible = false;  // (the counters have just said so)
      }
    }
    if (repaired) {
      HIP_TRY(hipMemcpyAsync(all.data(), st->d_distinct_counters.p, all.size() * sizeof(unsigned long long),
                             hipMemcpyDeviceToHost, st->stream));
      HIP_TRY(hipStreamSynchronize(st->stream));
    }
    for (size_t i = 0; i < d_scan.size(); i++) scan_acc_merge(g->scan[i], d_scan[i]);
    for (size_t i = 0; i < d_count.size(); i++) {
      g->count[i].total += d_count[i].total;
      g->count[i].non_null += d_count[i].non_null;
    }
    for (size_t i = 0; i < d_como.size(); i++) como_acc_merge(g->como[i], d_como[i]);
  }
  for (size_t i = 0; i < plan->distinct.size(); i++)
    TGX_TRY(distinct_totals(st, i, &g->distinct[i], err, all.empty() ? nullptr : all.data()));
  return TGX_OK;
}

static double key_to_double(int64_t k) {
  int64_t bits = f64_total_key(k);  // the transform is an involution
  double d;
  memcpy(&d, &bits, 8);
  return d;
}

static double i128_to_double(uint64_t lo, int64_t hi) {
  const __int128 v = (__int128)(((unsigned __int128)(uint64_t)hi << 64) | (unsigned __int128)lo);  // (no shift of a negative value)
  return (double)v;
}

static void fill_stats(const ScanAcc &a, bool variance, tgx_result *r) {
  r->is_float = a.is_float;
  r->total = a.total;
  r->non_null = a.non_null;
  r->has_value = a.non_null > 0;
  if (!r->has_value) {
    r->min_f = r->max_f = r->mean = NAN;
    r->var_samp = r->stddev_samp = NAN;
    return;
  }
  if (a.is_float) {
    r->min_f = key_to_double(a.min_k);
    r->max_f = key_to_double(a.max_k);
    r->sum_f = isfinite(a.sum) ? a.sum + a.comp : a.sum;
    r->sum_i = 0;
  } else {
    r->min_i = a.min_k;
    r->max_i = a.max_k;
    r->min_f = (double)a.min_k;
    r->max_f = (double)a.max_k;
    r->sum_i = (int64_t)a.sum_lo;  // SUM(Int64) wraps
    r->sum_f = i128_to_double(a.sum_lo, a.sum_hi);
  }
  r->mean = r->sum_f / (double)a.non_null;
  r->var_samp = r->stddev_samp = NAN;
  if (variance && a.var_n >= 2) {
    r->has_variance = 1;
    r->var_samp = a.var_m2 / (double)(a.var_n - 1);
    r->stddev_samp = sqrt(r->var_samp);
  }
}

// The cardinality estimate of a HyperLogLog sketch of 2^p registers whose ranks run 0 .. q + 1: Ertl, "New
// cardinality estimation algorithms for HyperLogLog sketches" (2017), algorithm 6 -- the estimator of DataFusion's
// APPROX_DISTINCT (datafusion-functions-aggregate 50.3.0, hyperloglog.rs `count`, there with p = 14, q = 50; here
// q = 32: the rank comes from a 32-bit word).
static double hll_sigma(double x) {
  if (x == 1.0) return INFINITY;
  double y = 1.0, z = x;
  for (;;) {
    x *= x;
    const double z0 = z;
    z += x * y;
    y += y;
    if (z0 == z) return z;
  }
}
static double hll_tau(double x) {
  if (x == 0.0 || x == 1.0) return 0.0;
  double y = 1.0, z = 1.0 - x;
  for (;;) {
    x = sqrt(x);
    const double z0 = z;
    y *= 0.5;
    z -= (1.0 - x) * (1.0 - x) * y;
    if (z0 == z) return z / 3.0;
  }
}
static uint64_t hll_estimate(const std::vector<uint8_t> &regs) {
  if (regs.empty()) return 0;
  constexpr int q = kHllMaxRank - 1;
  uint32_t hist[kHllMaxRank + 1] = {0};
  for (int r = 0; r < kHllRegisters; r++) hist[std::min<int>(regs[r], kHllMaxRank)]++;
  const double m = (double)kHllRegisters;
  double z = m * hll_tau((m - (double)hist[q + 1]) / m);
  for (int k = q; k >= 1; k--) z = 0.5 * (z + (double)hist[k]);
  z += m * hll_sigma((double)hist[0] / m);
  const double e = 0.5 / log(2.0) * m * m / z;
  return std::isfinite(e) ? (uint64_t)llround(e) : 0;
}

extern "C" tgx_status tgx_finalize(const tgx_plan *plan, tgx_state *st, tgx_result *results,
                                   size_t n_results, tgx_error *err) try {
  bind_thread();
  if (!plan || !st || st->plan != plan) return fail(err, TGX_INVALID_ARGUMENT, "state does not belong to plan");
  if (n_results < plan->specs.size() || (!results && !plan->specs.empty()))
    return fail(err, TGX_INVALID_ARGUMENT, "results has room for %zu of %zu specs", n_results, plan->specs.size());
  Gathered g;
  TGX_TRY(gather(st, &g, err));
  struct FetchScope {  // pattern / length counters: one readback for all tasks, dropped on every way out
    tgx_state *s;
    ~FetchScope() { regex_fetch_end(s); }
  } fetch_scope{st};
  if (st->regex) TGX_TRY(regex_fetch_begin(st, err));
  for (size_t i = 0; i < plan->specs.size(); i++) {
    tgx_result *r = &results[i];
    memset(r, 0, sizeof(*r));
    const SpecBinding &b = plan->bind[i];
    r->kind = b.kind;
    switch (b.kind) {
      case TGX_CHECK_COUNT:
        if (b.count_src == Source::kScan) {
          r->total = g.scan[b.slot].total;
          r->non_null = g.scan[b.slot].non_null;
          // a scan task that only exists for DISTINCT's range decision does not run on Utf8 columns:
          // there the distinct kernel has counted the non-NULL rows
          const int col = plan->scan[b.slot].column;
          if (g.scan[b.slot].total == 0)
            for (size_t d = 0; d < plan->distinct.size(); d++)
              if (plan->distinct[d].column == col && plan->distinct[d].tuple.empty() && g.distinct[d].total > 0) {
                r->total = (int64_t)g.distinct[d].total;
                r->non_null = (int64_t)g.distinct[d].non_null;
              }
        } else {
          r->total = g.count[b.slot].total;
          r->non_null = g.count[b.slot].non_null;
        }
        break;
      case TGX_CHECK_NUMERIC_STATS:
        fill_stats(g.scan[b.slot], plan->scan[b.slot].variance, r);
        break;
      case TGX_CHECK_DISTINCT: {
        const DistinctTotals &t = g.distinct[b.slot];
        r->total = (int64_t)t.total;
        r->non_null = (int64_t)t.non_null;
        r->distinct = (int64_t)t.distinct;
        // groups with cnt == 1 (GROUP BY cols ... WHERE cnt = 1, uniqueness.rs:670-680): keys seen exactly once, plus --
        // for a single column -- the NULL group when it has one row.  A tuple with NULL components is a key of its
        // own and already among them (with one such row the group was counted twice).
        // (only tracked when the spec asks for TGX_FLAG_MULTIPLICITY; 0 otherwise)
        const bool null_group = plan->distinct[b.slot].tuple.empty() && (t.total - t.non_null) == 1;
        r->groups_once = (plan->specs[i].flags & TGX_FLAG_MULTIPLICITY)
                             ? (int64_t)(t.distinct - t.twice) + (null_group ? 1 : 0)
                             : 0;
        break;
      }
      case TGX_CHECK_COMOMENTS: {
        const ComomentAcc &a = g.como[b.slot];
        r->total = a.total;
        r->non_null = a.n;
        // the raw sums the analyzer reports (TG/analyzers/advanced/correlation.rs:239-249): the sums about (0, 0)
        ComomentAcc raw = a;
        como_rebase(raw, 0.0, 0.0);
        r->sum_x = (double)como_sum(raw, 0);
        r->sum_y = (double)como_sum(raw, 1);
        r->sum_x2 = (double)como_sum(raw, 2);
        r->sum_y2 = (double)como_sum(raw, 3);
        r->sum_xy = (double)como_sum(raw, 4);
        // the centred moments CORR / COVAR_SAMP are made of (TG/constraints/correlation.rs:260-275: DataFusion's
        // online accumulators arrive at these, not at the raw sums): taken about the pivots, which lie near the data
        if (a.n > 0) {
          const xdouble n = (xdouble)a.n, s0 = como_sum(a, 0), s1 = como_sum(a, 1);
          const xdouble m2x = como_sum(a, 2) - s0 * s0 / n, m2y = como_sum(a, 3) - s1 * s1 / n;
          r->co_mean_x = (double)((xdouble)a.px + s0 / n);
          r->co_mean_y = (double)((xdouble)a.py + s1 / n);
          r->co_m2_x = m2x > 0 ? (double)m2x : (m2x == m2x ? 0.0 : (double)m2x);
          r->co_m2_y = m2y > 0 ? (double)m2y : (m2y == m2y ? 0.0 : (double)m2y);
          r->co_c_xy = (double)(como_sum(a, 4) - s0 * s1 / n);
        }
        break;
      }
      case TGX_CHECK_APPROX_DISTINCT: {
        const HllTask &t = plan->hll[b.slot];
        if (st->hll_mode[b.slot] == 1) {  // the HyperLogLog lane of the column's scan
          r->total = g.scan[t.scan_slot].total;
          r->non_null = g.scan[t.scan_slot].non_null;
          r->distinct = (int64_t)hll_estimate(g.hll[b.slot]);
        } else {  // the exact key set (string / dictionary columns, or a DISTINCT check of the same column)
          const DistinctTotals &d = g.distinct[t.distinct_slot];
          r->total = (int64_t)d.total;
          r->non_null = (int64_t)d.non_null;
          r->distinct = (int64_t)d.distinct;
        }
        break;
      }
      case TGX_CHECK_KLL:
        TGX_TRY(kll_fill_result(st, b.slot, r, err));
        break;
      case TGX_CHECK_LENGTH:
      case TGX_CHECK_REGEX_MATCH:
        TGX_TRY(regex_fill_result(st, b.slot, r, err));
        break;
      case TGX_CHECK_SPEARMAN:
        TGX_TRY(spearman_fill_result(st, b.slot, r, err));
        break;
      default:
        break;
    }
  }
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

// ------------------------------------------------------------------------------------------------
// distinct: export / import / merge
tgx_status tgx::distinct_export_impl(tgx_state *st, size_t slot, uint32_t world,
                                       const void **device_records, uint64_t *counts, tgx_error *err) {
  TGX_TRY(coalesce_flush(st, err));
  DistinctState &ds = st->distinct[slot];
  const bool mult = st->plan->distinct[slot].multiplicity;
  TGX_TRY(state_init_device(st, err));
  TGX_TRY(distinct_resolve(st, slot, err));
  std::vector<unsigned long long> h_counts(world, 0);
  unsigned long long c[kNumDistinctCounters];
  TGX_TRY(distinct_read_counters(st, ds, c, err));
  const uint64_t n_keys = c[kCntDistinct];
  const uint64_t empty_rows = c[kCntEmptyRows];
  HIP_TRY(ds.export_counts.reserve(2 * world * sizeof(unsigned long long)));
  unsigned long long *d_counts = ds.export_counts.as<unsigned long long>();
  unsigned long long *d_cursors = d_counts + world;
  HIP_TRY(hipMemsetAsync(d_counts, 0, 2 * world * sizeof(unsigned long long), st->stream));
  if (ds.mode == DistinctMode::kHash && ds.wide)
    launch_hash_export_count128(hash_view(ds), world, d_counts, st->stream);
  else if (ds.mode == DistinctMode::kHash)
    launch_hash_export_count(hash_view(ds), world, d_counts, st->stream);
  else if (ds.mode == DistinctMode::kBitmap)
    launch_bitmap_export_count(bitmap_view(ds), world, d_counts, st->stream);
  HIP_TRY(hipMemcpyAsync(h_counts.data(), d_counts, world * sizeof(unsigned long long), hipMemcpyDeviceToHost, st->stream));
  HIP_TRY(hipStreamSynchronize(st->stream));
  // the all-ones key lives in a side counter; it travels as one extra record to its owner
  uint32_t empty_owner = 0;
  if (empty_rows > 0) {
    // same owner function as the kernels (distinct.hip owner_of)
    uint64_t x = kEmptyKey ^ 0x9e3779b97f4a7c15ULL;
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; x ^= x >> 31;
    empty_owner = (uint32_t)((x >> 32) % world);
    h_counts[empty_owner] += 1;
  }
  std::vector<unsigned long long> starts(world, 0);
  uint64_t total = 0;
  for (uint32_t r = 0; r < world; r++) {
    starts[r] = total;
    total += h_counts[r];
  }
  (void)n_keys;
  const size_t rec_bytes = ds.wide ? sizeof(KeyRecord128) : sizeof(KeyRecord);
  HIP_TRY(ds.export_records.reserve(std::max<uint64_t>(total, 1) * rec_bytes));
  HIP_TRY(hipMemcpyAsync(d_cursors, starts.data(), world * sizeof(unsigned long long), hipMemcpyHostToDevice, st->stream));
  HIP_TRY(hipStreamSynchronize(st->stream));
  if (ds.mode == DistinctMode::kHash && ds.wide)
    launch_hash_export_scatter128(hash_view(ds), world, mult ? 1 : 0, d_cursors, ds.export_records.as<KeyRecord128>(),
                                  st->stream);
  else if (ds.mode == DistinctMode::kHash)
    launch_hash_export_scatter(hash_view(ds), world, mult ? 1 : 0, d_cursors, ds.export_records.as<KeyRecord>(), st->stream);
  else if (ds.mode == DistinctMode::kBitmap)
    launch_bitmap_export_scatter(bitmap_view(ds), world, mult ? 1 : 0, d_cursors, ds.export_records.as<KeyRecord>(), st->stream);
  if (empty_rows > 0) {
    KeyRecord rec{kEmptyKey, std::min<uint64_t>(empty_rows, 2)};
    uint64_t pos = starts[empty_owner] + h_counts[empty_owner] - 1;
    HIP_TRY(hipMemcpyAsync(ds.export_records.as<KeyRecord>() + pos, &rec, sizeof(rec), hipMemcpyHostToDevice, st->stream));
  }
  HIP_TRY(hipStreamSynchronize(st->stream));
  for (uint32_t r = 0; r < world; r++) counts[r] = h_counts[r];
  *device_records = ds.export_records.p;
  return TGX_OK;
}

extern "C" size_t tgx_distinct_record_bytes(const tgx_plan *plan, const tgx_state *st, size_t spec_index) {
  if (!plan || !st || st->plan != plan || spec_index >= plan->specs.size() ||
      plan->specs[spec_index].kind != TGX_CHECK_DISTINCT)
    return 0;
  const DistinctState &ds = st->distinct[plan->bind[spec_index].slot];
  const bool wide = ds.wide || is_any_string(ds.col_type) || ds.col_type == TGX_DICT32_UTF8;
  return wide ? sizeof(KeyRecord128) : sizeof(KeyRecord);
}

extern "C" tgx_status tgx_distinct_export(const tgx_plan *plan, tgx_state *st, size_t spec_index,
                                          uint32_t world, const void **device_records, uint64_t *counts,
                                          tgx_error *err) try {
  bind_thread();
  if (!plan || !st || st->plan != plan) return fail(err, TGX_INVALID_ARGUMENT, "state does not belong to plan");
  if (spec_index >= plan->specs.size() || plan->specs[spec_index].kind != TGX_CHECK_DISTINCT)
    return fail(err, TGX_INVALID_ARGUMENT, "spec %zu is not a DISTINCT check", spec_index);
  if (world == 0 || world > 256 || !device_records || !counts)
    return fail(err, TGX_INVALID_ARGUMENT, "bad arguments (world must be 1..256)");
  TGX_TRY(need_device(err));
  return distinct_export_impl(st, plan->bind[spec_index].slot, world, device_records, counts, err);
} catch (...) {
  return tgx::abi_exception(err);
}

// union `n` device records into the state's set (switching it to hash mode)
tgx_status tgx::distinct_import_records(tgx_state *st, size_t slot, const void *d_recs, uint64_t n, bool wide,
                                          tgx_error *err) {
  DistinctState &ds = st->distinct[slot];
  const bool mult = st->plan->distinct[slot].multiplicity;
  TGX_TRY(state_init_device(st, err));
  TGX_TRY(distinct_resolve(st, slot, err));
  if (ds.mode == DistinctMode::kBitmap) TGX_TRY(bitmap_to_hash(st, ds, mult, n, err));
  if (ds.mode == DistinctMode::kHash && ds.capacity > 0 && ds.wide != wide)
    return fail(err, TGX_INVALID_ARGUMENT, "distinct: cannot unite a Utf8 key set with a numeric one");
  ds.mode = DistinctMode::kHash;
  ds.wide = wide;
  TGX_TRY(hash_ensure(st, ds, mult, n, err));
  // the EMPTY stand-in's rows arrive through counters[2]; [5] is scratch
  if (wide)
    launch_hash_import128((const KeyRecord128 *)d_recs, n, hash_view(ds), mult ? 1 : 0,
                          ds.counters.as<unsigned long long>(), st->stream);
  else
    launch_hash_import((const KeyRecord *)d_recs, n, hash_view(ds), mult ? 1 : 0,
                       ds.counters.as<unsigned long long>(), st->stream);
  return TGX_OK;
}

extern "C" tgx_status tgx_distinct_import(const tgx_plan *plan, tgx_state *st, size_t spec_index,
                                          const void *device_records, uint64_t n_records, tgx_error *err) try {
  bind_thread();
  if (!plan || !st || st->plan != plan) return fail(err, TGX_INVALID_ARGUMENT, "state does not belong to plan");
  if (spec_index >= plan->specs.size() || plan->specs[spec_index].kind != TGX_CHECK_DISTINCT)
    return fail(err, TGX_INVALID_ARGUMENT, "spec %zu is not a DISTINCT check", spec_index);
  TGX_TRY(need_device(err));
  TGX_TRY(coalesce_flush(st, err));
  const size_t slot = plan->bind[spec_index].slot;
  DistinctState &ds = st->distinct[slot];
  TGX_TRY(state_init_device(st, err));
  // keep the row counts, replace the key set
  unsigned long long c[kNumDistinctCounters];
  TGX_TRY(distinct_read_counters(st, ds, c, err));
  const unsigned long long valid_rows = c[kCntValidRows];
  const bool wide = ds.wide || is_any_string(ds.col_type) || ds.col_type == TGX_DICT32_UTF8;
  ds.seen.release();
  ds.twice.release();
  ds.keys.release();
  ds.dup.release();
  ds.capacity = 0;
  ds.rows_upper_bound = 0;
  ds.mode = DistinctMode::kHash;
  unsigned long long zero[kNumDistinctCounters];
  memset(zero, 0, sizeof(zero));
  zero[kCntValidRows] = valid_rows;
  HIP_TRY(hipMemcpyAsync(ds.counters.p, zero, sizeof(zero), hipMemcpyHostToDevice, st->stream));
  HIP_TRY(hipStreamSynchronize(st->stream));  // `zero` is on this stack frame
  TGX_TRY(distinct_import_records(st, slot, device_records, n_records, wide, err));
  HIP_TRY(hipStreamSynchronize(st->stream));
  ds.partitioned = true;
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

static tgx_status distinct_slot_of(const tgx_plan *plan, tgx_state *st, size_t spec_index, size_t *slot,
                                   tgx_error *err) {
  if (!plan || !st || st->plan != plan) return fail(err, TGX_INVALID_ARGUMENT, "state does not belong to plan");
  if (spec_index >= plan->specs.size() || plan->specs[spec_index].kind != TGX_CHECK_DISTINCT)
    return fail(err, TGX_INVALID_ARGUMENT, "spec %zu is not a DISTINCT check", spec_index);
  *slot = (size_t)plan->bind[spec_index].slot;
  return TGX_OK;
}

extern "C" tgx_status tgx_distinct_range_hint(const tgx_plan *plan, tgx_state *st, size_t spec_index, int64_t lo,
                                              int64_t hi, tgx_error *err) try {
  size_t slot = 0;
  TGX_TRY(distinct_slot_of(plan, st, spec_index, &slot, err));
  TGX_TRY(coalesce_flush(st, err));
  DistinctState &ds = st->distinct[slot];
  if (ds.mode != DistinctMode::kUndecided)
    return fail(err, TGX_INVALID_ARGUMENT, "range hint must be given before the first batch (after tgx_state_reset)");
  if (hi < lo) return fail(err, TGX_INVALID_ARGUMENT, "range hint: hi < lo");
  ds.has_hint = true;
  ds.hint_lo = lo;
  ds.hint_hi = hi;
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

extern "C" tgx_status tgx_distinct_bitmap_view(const tgx_plan *plan, tgx_state *st, size_t spec_index, int64_t *base,
                                               uint64_t *n_words, const void **seen, const void **twice,
                                               tgx_error *err) try {
  bind_thread();
  size_t slot = 0;
  TGX_TRY(distinct_slot_of(plan, st, spec_index, &slot, err));
  TGX_TRY(coalesce_flush(st, err));
  TGX_TRY(distinct_resolve(st, slot, err));
  DistinctState &ds = st->distinct[slot];
  if (ds.mode != DistinctMode::kBitmap)
    return fail(err, TGX_UNSUPPORTED, "the key set is not a range bitmap; use tgx_distinct_export / _import");
  HIP_TRY(hipStreamSynchronize(st->stream));
  if (base) *base = ds.base;
  if (n_words) *n_words = ds.bitmap_words;
  if (seen) *seen = ds.seen.p;
  if (twice) *twice = st->plan->distinct[slot].multiplicity ? ds.twice.p : nullptr;
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

extern "C" tgx_status tgx_distinct_adopt_slices(const tgx_plan *plan, tgx_state *st, size_t spec_index,
                                                int64_t slice_base, const void *seen_slices,
                                                const void *twice_slices, uint32_t n_slices, uint64_t slice_words,
                                                uint64_t slice_stride_words, tgx_error *err) try {
  bind_thread();
  size_t slot = 0;
  TGX_TRY(distinct_slot_of(plan, st, spec_index, &slot, err));
  TGX_TRY(need_device(err));
  TGX_TRY(coalesce_flush(st, err));
  TGX_TRY(state_init_device(st, err));
  DistinctState &ds = st->distinct[slot];
  const bool mult = plan->distinct[slot].multiplicity;
  if (!seen_slices || n_slices == 0 || slice_words == 0) return fail(err, TGX_INVALID_ARGUMENT, "bad slice arguments");
  if (slice_stride_words == 0) slice_stride_words = slice_words;
  if (slice_stride_words < slice_words) return fail(err, TGX_INVALID_ARGUMENT, "slice stride shorter than a slice");
  if (mult && !twice_slices) return fail(err, TGX_INVALID_ARGUMENT, "this check needs the 'twice' slices too");
  if (ds.wide) return fail(err, TGX_INVALID_ARGUMENT, "Utf8 key sets have no range bitmap");
  unsigned long long c[kNumDistinctCounters];
  TGX_TRY(distinct_read_counters(st, ds, c, err));
  DevBuf &new_seen = ds.spare_seen, &new_twice = ds.spare_twice;
  HIP_TRY(new_seen.reserve(slice_words * 4 + 16));
  if (mult) HIP_TRY(new_twice.reserve(slice_words * 4 + 16));
  unsigned long long zero[kNumDistinctCounters];
  memset(zero, 0, sizeof(zero));
  zero[kCntValidRows] = c[kCntValidRows];
  zero[kCntOutOfRange] = c[kCntOutOfRange];
  HIP_TRY(hipMemcpyAsync(ds.counters.p, zero, sizeof(zero), hipMemcpyHostToDevice, st->stream));
  launch_bitmap_adopt((const uint32_t *)seen_slices, mult ? (const uint32_t *)twice_slices : nullptr, n_slices,
                      slice_words, slice_stride_words, new_seen.as<uint32_t>(), mult ? new_twice.as<uint32_t>() : nullptr,
                      ds.counters.as<unsigned long long>(), st->stream);
  HIP_TRY(hipStreamSynchronize(st->stream));
  std::swap(ds.seen, ds.spare_seen);  // the old bitmap stays around as the spare of the next round
  std::swap(ds.twice, ds.spare_twice);
  ds.capacity = 0;
  ds.mode = DistinctMode::kBitmap;
  ds.base = slice_base;
  ds.range = slice_words * 32;
  ds.bitmap_words = slice_words;
  ds.partitioned = true;
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

extern "C" tgx_status tgx_merge(const tgx_plan *plan, tgx_state *dst, tgx_state *const *srcs, size_t n_srcs,
                                tgx_error *err) try {
  bind_thread();
  if (!plan || !dst || dst->plan != plan) return fail(err, TGX_INVALID_ARGUMENT, "dst does not belong to plan");
  TGX_TRY(coalesce_flush(dst, err));
  {  // what can refuse a source is checked for ALL sources before dst takes anything of any of them
    std::vector<int> mode(dst->hll_mode.begin(), dst->hll_mode.end());
    for (size_t i = 0; i < n_srcs; i++) {
      tgx_state *src = srcs ? srcs[i] : nullptr;
      if (!src || src->plan != plan) return fail(err, TGX_INVALID_ARGUMENT, "src %zu does not belong to plan", i);
      if (src == dst) return fail(err, TGX_INVALID_ARGUMENT, "src %zu is dst", i);
      TGX_TRY(spearman_check_mergeable(src, err));
      TGX_TRY(coalesce_flush(src, err));  // (its noted batches decide which form an APPROX_DISTINCT task takes)
      for (size_t k = 0; k < plan->hll.size(); k++) {
        if (src->hll_mode[k] == 0) continue;
        if (mode[k] == 0) mode[k] = src->hll_mode[k];
        if (mode[k] != src->hll_mode[k])
          return fail(err, TGX_INVALID_ARGUMENT,
                      "APPROX_DISTINCT task %zu: one state holds registers, the other a key set (src %zu); nothing was merged",
                      k, i);
      }
    }
  }
  for (size_t i = 0; i < n_srcs; i++) {
    tgx_state *src = srcs[i];
    Gathered g;
    // distinct totals are handled set-wise below; gather the fixed-size parts
    {
      std::vector<DistinctState> hold;  // gather() reads distinct counters too; harmless
      TGX_TRY(gather(src, &g, err));
    }
    for (size_t k = 0; k < g.scan.size(); k++) scan_acc_merge(dst->h_scan[k], g.scan[k]);
    for (size_t k = 0; k < g.count.size(); k++) {
      dst->h_count[k].total += g.count[k].total;
      dst->h_count[k].non_null += g.count[k].non_null;
    }
    for (size_t k = 0; k < g.como.size(); k++) como_acc_merge(dst->h_como[k], g.como[k]);
    for (size_t k = 0; k < plan->hll.size(); k++) {
      if (src->hll_mode[k] == 0) continue;
      if (dst->hll_mode[k] == 0) dst->hll_mode[k] = src->hll_mode[k];
      if (dst->hll_mode[k] != src->hll_mode[k])
        return fail(err, TGX_INVALID_ARGUMENT, "APPROX_DISTINCT task %zu: one state holds registers, the other a key set", k);
      if (g.hll[k].empty()) continue;
      std::vector<uint8_t> &out = dst->h_hll[k];
      if (out.empty()) {
        out = g.hll[k];
      } else {
        for (int r = 0; r < kHllRegisters; r++) out[r] = std::max(out[r], g.hll[k][r]);
      }
    }
    for (size_t k = 0; k < plan->distinct.size(); k++) {
      DistinctState &s = src->distinct[k];
      DistinctState &d = dst->distinct[k];
      const bool src_has_set = s.mode == DistinctMode::kBitmap || s.mode == DistinctMode::kHash;
      if (s.partitioned || !src_has_set) {
        // owner-partitioned (or count-only) partial: key sets are disjoint by construction
        if (src_has_set && !s.partitioned)
          return fail(err, TGX_INTERNAL, "distinct merge: unexpected state");
        const DistinctTotals &t = g.distinct[k];
        // remove the +1 adjustments distinct_totals() made for the EMPTY stand-in: they are re-derived
        d.h_total += t.total;
        d.h_non_null += t.non_null;
        d.h_distinct += t.distinct - (t.empty_rows > 0 ? 1 : 0);
        d.h_twice += t.twice - (t.empty_rows > 1 ? 1 : 0);
        d.h_empty_rows += t.empty_rows;
        if (s.partitioned) d.partitioned = true;
      } else {
        // exact set union on the device
        TGX_TRY(need_device(err));
        const void *recs = nullptr;
        uint64_t cnt = 0;
        TGX_TRY(distinct_export_impl(src, k, 1, &recs, &cnt, err));
        TGX_TRY(state_init_device(dst, err));
        TGX_TRY(distinct_import_records(dst, k, recs, cnt, s.wide, err));
        HIP_TRY(hipStreamSynchronize(dst->stream));
        d.h_total += (uint64_t)s.total_rows + s.h_total;
        unsigned long long c[kNumDistinctCounters];
        TGX_TRY(distinct_read_counters(src, s, c, err));
        d.h_non_null += c[kCntValidRows] + s.h_non_null;
        d.h_distinct += s.h_distinct;
        d.h_twice += s.h_twice;
        d.h_empty_rows += s.h_empty_rows;
      }
    }
    TGX_TRY(kll_merge_states(dst, src, err));
    TGX_TRY(regex_merge_states(dst, src, err));
  }
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

// ------------------------------------------------------------------------------------------------
// wire form
namespace {
struct Writer {
  uint8_t *buf;
  size_t cap, len = 0;
  void put(const void *p, size_t n) {
    if (buf && len + n <= cap) memcpy(buf + len, p, n);
    len += n;
  }
  template <class T>
  void pod(const T &v) { put(&v, sizeof(T)); }
};
struct Reader {
  const uint8_t *buf;
  size_t len, pos = 0;
  bool ok = true;
  void get(void *p, size_t n) {
    if (pos + n > len) {
      ok = false;
      memset(p, 0, n);
      return;
    }
    memcpy(p, buf + pos, n);
    pos += n;
  }
  template <class T>
  T pod() {
    T v;
    get(&v, sizeof(T));
    return v;
  }
};
constexpr uint32_t kWireMagic = 0x53584754;  // "TGXS"
constexpr uint32_t kWireVersion = 2;  // 2: ComomentAcc carries its pivots
}  // namespace

extern "C" tgx_status tgx_state_serialize(const tgx_plan *plan, tgx_state *st, uint8_t *buf, size_t cap,
                                          size_t *len, tgx_error *err) try {
  bind_thread();
  if (!plan || !st || st->plan != plan || !len) return fail(err, TGX_INVALID_ARGUMENT, "bad arguments");
  TGX_TRY(spearman_check_mergeable(st, err));
  Gathered g;
  TGX_TRY(gather(st, &g, err));
  Writer w{buf, cap};
  w.pod(kWireMagic);
  w.pod(kWireVersion);
  w.pod((uint32_t)g.scan.size());
  w.pod((uint32_t)g.count.size());
  w.pod((uint32_t)g.como.size());
  w.pod((uint32_t)g.distinct.size());
  w.pod((uint32_t)plan->kll.size());
  w.pod((uint32_t)regex_num_tasks(plan));
  w.pod((uint32_t)plan->hll.size());
  for (auto &a : g.scan) w.pod(a);
  for (auto &a : g.count) w.pod(a);
  for (auto &a : g.como) w.pod(a);
  for (size_t k = 0; k < g.distinct.size(); k++) {
    DistinctState &ds = st->distinct[k];
    const DistinctTotals &t = g.distinct[k];
    const bool has_set = ds.mode == DistinctMode::kBitmap || ds.mode == DistinctMode::kHash;
    uint32_t partitioned = (ds.partitioned || !has_set) ? 1 : 0;
    w.pod(partitioned);
    w.pod((uint32_t)(ds.wide ? 1 : 0));
    w.pod(t);
    uint64_t n_records = 0;
    if (!partitioned) {
      // non-partitioned sets travel with their keys so the receiver can take an exact union
      const void *recs = nullptr;
      TGX_TRY(distinct_export_impl(st, k, 1, &recs, &n_records, err));
      w.pod(n_records);
      size_t bytes = (size_t)n_records * (ds.wide ? sizeof(KeyRecord128) : sizeof(KeyRecord));
      if (w.buf && w.len + bytes <= w.cap)
        HIP_TRY(hipMemcpy(w.buf + w.len, recs, bytes, hipMemcpyDeviceToHost));
      w.len += bytes;
    } else {
      w.pod(n_records);
    }
  }
  TGX_TRY(kll_serialize(st, &w.len, w.buf, w.cap, err));
  TGX_TRY(regex_serialize(st, &w.len, w.buf, w.cap, err));
  for (size_t k = 0; k < plan->hll.size(); k++) {  // { u32 mode, u32 has_registers; registers }
    w.pod((uint32_t)st->hll_mode[k]);
    w.pod((uint32_t)(g.hll[k].empty() ? 0 : 1));
    if (!g.hll[k].empty()) w.put(g.hll[k].data(), kHllRegisters);
  }
  *len = w.len;
  if (buf && w.len > cap) return fail(err, TGX_INVALID_ARGUMENT, "buffer too small: need %zu bytes", w.len);
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

extern "C" tgx_status tgx_state_deserialize(const tgx_plan *plan, const uint8_t *buf, size_t len,
                                            tgx_state **out, tgx_error *err) try {
  bind_thread();
  if (!plan || !buf || !out) return fail(err, TGX_INVALID_ARGUMENT, "bad arguments");
  *out = nullptr;
  Reader r{buf, len};
  if (r.pod<uint32_t>() != kWireMagic) return fail(err, TGX_INVALID_ARGUMENT, "not a tgx state blob");
  if (r.pod<uint32_t>() != kWireVersion) return fail(err, TGX_INVALID_ARGUMENT, "state blob version mismatch");
  uint32_t n_scan = r.pod<uint32_t>(), n_count = r.pod<uint32_t>(), n_como = r.pod<uint32_t>(),
           n_dist = r.pod<uint32_t>(), n_kll = r.pod<uint32_t>(), n_regex = r.pod<uint32_t>(), n_hll = r.pod<uint32_t>();
  if (n_scan != plan->scan.size() || n_count != plan->count.size() || n_como != plan->como.size() ||
      n_dist != plan->distinct.size() || n_kll != plan->kll.size() || n_regex != regex_num_tasks(plan) ||
      n_hll != plan->hll.size())
    return fail(err, TGX_INVALID_ARGUMENT, "state blob was produced by a different plan");
  std::unique_ptr<tgx_state, void (*)(tgx_state *)> st(new tgx_state(), tgx_state_destroy);
  state_init_host(st.get(), plan);
  for (auto &a : st->h_scan) a = r.pod<ScanAcc>();
  for (auto &a : st->h_count) a = r.pod<CountAcc>();
  for (auto &a : st->h_como) a = r.pod<ComomentAcc>();
  for (size_t k = 0; k < n_dist; k++) {
    DistinctState &ds = st->distinct[k];
    uint32_t partitioned = r.pod<uint32_t>();
    const bool wide = r.pod<uint32_t>() != 0;
    DistinctTotals t = r.pod<DistinctTotals>();
    uint64_t n_records = r.pod<uint64_t>();
    if (!r.ok) break;
    if (partitioned) {
      ds.partitioned = true;
      ds.h_total = t.total;
      ds.h_non_null = t.non_null;
      ds.h_distinct = t.distinct - (t.empty_rows > 0 ? 1 : 0);
      ds.h_twice = t.twice - (t.empty_rows > 1 ? 1 : 0);
      ds.h_empty_rows = t.empty_rows;
    } else {
      // rebuild the key set on the device from the records.  n_records comes from the blob: bound it by the bytes
      // that are really there BEFORE multiplying (a crafted count would wrap the product past the check)
      const size_t rec_bytes = wide ? sizeof(KeyRecord128) : sizeof(KeyRecord);
      if (n_records > (r.len - r.pos) / rec_bytes) {
        r.ok = false;
        break;
      }
      size_t bytes = (size_t)n_records * rec_bytes;
      tgx_status s = need_device(err);
      if (s != TGX_OK) return s;
      s = state_init_device(st.get(), err);
      if (s != TGX_OK) return s;
      DevBuf tmp;
      HIP_TRY(tmp.reserve(std::max<size_t>(bytes, 16)));
      HIP_TRY(hipMemcpy(tmp.p, r.buf + r.pos, bytes, hipMemcpyHostToDevice));
      r.pos += bytes;
      s = distinct_import_records(st.get(), k, tmp.p, n_records, wide, err);
      if (s != TGX_OK) return s;
      HIP_TRY(hipStreamSynchronize(st->stream));
      ds.h_total = t.total;
      ds.h_non_null = t.non_null;
    }
  }
  if (r.ok) {
    tgx_status s = kll_deserialize(st.get(), r.buf, r.len, &r.pos, err);
    if (s != TGX_OK) return s;
    s = regex_deserialize(st.get(), r.buf, r.len, &r.pos, err);
    if (s != TGX_OK) return s;
    for (size_t k = 0; k < plan->hll.size() && r.ok; k++) {
      const uint32_t mode = r.pod<uint32_t>(), has = r.pod<uint32_t>();
      if (mode > 2 || has > 1) return fail(err, TGX_INVALID_ARGUMENT, "malformed state blob (APPROX_DISTINCT task)");
      st->hll_mode[k] = (int)mode;
      if (has) {
        st->h_hll[k].resize(kHllRegisters);
        r.get(st->h_hll[k].data(), kHllRegisters);
      }
    }
  }
  if (!r.ok) return fail(err, TGX_INVALID_ARGUMENT, "truncated state blob");
  *out = st.release();
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}
