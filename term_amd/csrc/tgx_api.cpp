// tgx_api.cpp -- the C ABI of include/tgx.h: plans, states, batch updates, merges, results.
//
// Host-side only bookkeeping lives here; every per-row computation is a HIP kernel under
// kernels/.  There is no CPU fallback: without a gfx950 device the compute entry points fail.
// This file: errors, the device context, plans, states, profiling, gather + finalize.  The batch path is update.cpp,
// the key sets distinct_state.cpp, small-batch coalescing coalesce.cpp, state blobs wire.cpp (api_internal.h).
#include "api_internal.h"

#include <errno.h>
#include <sys/random.h>

Context g_ctx;

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
  tgx::dev_cache_set_live(true);
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

extern "C" tgx_status tgx_shutdown(void) try {
  copy_pool_shutdown();  // (the helper threads of the coalescing arenas' copies: stopped and joined)
  std::lock_guard<std::mutex> lock(g_ctx.mu);
  if (g_ctx.inited) {
    (void)hipSetDevice(g_ctx.device);
    tgx::dev_cache_set_live(false);  // (a state destroyed from here on frees its blocks: nobody would trim them)
    tgx::dev_cache_trim();  // (blocks of destroyed states; live states keep theirs)
  }
  g_ctx.inited = false;
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(nullptr);
}

extern "C" tgx_status tgx_trim(void) try {
  bind_thread();
  tgx::dev_cache_trim();
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(nullptr);
}

extern "C" tgx_status tgx_cache_stats_get(tgx_cache_stats *out) try {
  if (!out) return TGX_INVALID_ARGUMENT;
  tgx::dev_cache_stats(out);
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
// The fingerprint key: 16 bytes from the operating system (TGX_FINGERPRINT_KEY=<32 hex digits> for a reproducible run).
static tgx_status draw_fingerprint_key(tgx_plan *plan, tgx_error *err) {
  uint8_t raw[16];
  bool have = false;
  if (const char *e = getenv("TGX_FINGERPRINT_KEY")) {
    if (strlen(e) != 32) return fail(err, TGX_INVALID_ARGUMENT, "TGX_FINGERPRINT_KEY must be 32 hex digits");
    for (int i = 0; i < 16; i++) {
      unsigned v = 0;
      if (sscanf(e + 2 * i, "%2x", &v) != 1) return fail(err, TGX_INVALID_ARGUMENT, "TGX_FINGERPRINT_KEY must be 32 hex digits");
      raw[i] = (uint8_t)v;
    }
    have = true;
  }
  for (size_t got = 0; !have;) {
    const ssize_t n = getrandom(raw + got, sizeof(raw) - got, 0);
    if (n < 0) {
      if (errno == EINTR) continue;
      return fail(err, TGX_INTERNAL, "getrandom failed: %s", strerror(errno));
    }
    got += (size_t)n;
    have = got == sizeof(raw);
  }
  uint32_t k[4];
  memcpy(k, raw, 16);
  plan->fp_key = fp_key_expand(k);
  return TGX_OK;
}

extern "C" tgx_status tgx_plan_set_fingerprint_key(tgx_plan *plan, const uint8_t key[16], tgx_error *err) try {
  if (!plan || !key) return fail(err, TGX_INVALID_ARGUMENT, "plan/key is NULL");
  if (plan->fp_key_locked.load())
    return fail(err, TGX_INVALID_ARGUMENT, "the fingerprint key is fixed once a state of the plan exists");
  uint32_t k[4];
  memcpy(k, key, 16);
  plan->fp_key = fp_key_expand(k);
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

extern "C" tgx_status tgx_plan_get_fingerprint_key(const tgx_plan *plan, uint8_t key_out[16]) {
  if (!plan || !key_out) return TGX_INVALID_ARGUMENT;
  memcpy(key_out, plan->fp_key.k, 16);
  return TGX_OK;
}

extern "C" tgx_status tgx_plan_create(const tgx_check_spec *specs, size_t n_specs, tgx_plan **out,
                                      tgx_error *err) try {
  if (!out) return fail(err, TGX_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (n_specs > 0 && !specs) return fail(err, TGX_INVALID_ARGUMENT, "specs is NULL");
  std::unique_ptr<tgx_plan> plan(new tgx_plan());
  TGX_TRY(draw_fingerprint_key(plan.get(), err));
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
        if (sp.flags & TGX_FLAG_EXACT_KEYS) plan->distinct[slot].exact = true;
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
    P->stats_on.assign(P->n_columns_needed, 0);
    for (auto &t : P->distinct)
      if (t.tuple.empty() && !t.approx_only) P->key_column[t.column] = 1;
    // (by what was ASKED: a key column has a scan task of its own for the range decisions of its key set)
    for (size_t i = 0; i < n_specs; i++) {
      const tgx_check_spec &sp = P->specs[i];
      if (sp.kind == TGX_CHECK_NUMERIC_STATS || sp.kind == TGX_CHECK_KLL || sp.kind == TGX_CHECK_APPROX_DISTINCT)
        P->stats_on[sp.column] = 1;
      if (sp.kind == TGX_CHECK_COMOMENTS) P->stats_on[sp.column] = P->stats_on[sp.column2] = 1;
    }
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
      P->stats_on[i] |= sp_used[i];
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
ScanAcc scan_acc_identity() {
  ScanAcc a;
  memset(&a, 0, sizeof(a));
  a.min_k = INT64_MAX;
  a.max_k = INT64_MIN;
  return a;
}

void host_two_sum(double &s, double &c, double x) {
  double t = s + x;
  double bp = t - s;
  c += (s - (t - bp)) + (x - bp);
  s = t;
}

void scan_acc_merge(ScanAcc &a, const ScanAcc &b) {
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
xdouble como_sum(const ComomentAcc &a, int k) {
  return std::isfinite(a.s[k]) ? (xdouble)a.s[k] + (xdouble)a.c[k] : (xdouble)a.s[k];
}
void como_store(ComomentAcc &a, int k, xdouble v) {
  a.s[k] = (double)v;
  a.c[k] = std::isfinite(a.s[k]) ? (double)(v - (xdouble)a.s[k]) : 0.0;
}
// the same sums about (px, py): x - px = (x - b.px) + dx
void como_rebase(ComomentAcc &b, double px, double py) {
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
void como_acc_merge(ComomentAcc &a, const ComomentAcc &b_in) {
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

void state_init_host(tgx_state *st, const tgx_plan *plan) {
  st->plan = plan;
  plan->fp_key_locked.store(true);  // (every state comes through here: created, deserialized, unpacked by a reduction)
  st->col_types.assign(plan->n_columns_needed, 0);
  st->h_scan.assign(plan->scan.size(), scan_acc_identity());
  st->h_count.assign(plan->count.size(), CountAcc{0, 0});
  ComomentAcc z;
  memset(&z, 0, sizeof(z));
  st->h_como.assign(plan->como.size(), z);
  st->distinct.clear();
  st->distinct.resize(plan->distinct.size());
  for (size_t k = 0; k < plan->distinct.size(); k++) st->distinct[k].exact = plan->distinct[k].exact;
  st->h_kll.clear();
  st->h_kll.resize(plan->kll.size());
  for (size_t i = 0; i < plan->kll.size(); i++) st->h_kll[i].k = plan->kll[i].k;
  st->h_hll.assign(plan->hll.size(), std::vector<uint8_t>());
  st->hll_mode.assign(plan->hll.size(), 0);
  regex_state_init(st);
  kll_state_init(st);
  spearman_state_init(st);
}

// one launch for all the small accumulators of a state (fresh, or reset): the scan accumulators to their identities,
// everything else to zero (all sizes are multiples of four bytes)
static void state_reset_small(tgx_state *st) {
  const tgx_plan *plan = st->plan;
  StateResetArgs a;
  memset(&a, 0, sizeof(a));
  auto zero = [&](void *p, size_t bytes) {
    if (!p || bytes == 0) return;
    a.zero[a.n_zero] = p;
    a.zero_bytes[a.n_zero] = (uint32_t)bytes;
    a.n_zero++;
  };
  if (!plan->scan.empty()) {
    a.ident = st->d_scan_acc.as<ScanAcc>();
    a.n_ident = (uint32_t)plan->scan.size();
    zero(st->d_pivots.p, plan->scan.size() * sizeof(double));
    zero(st->d_pivot_set.p, plan->scan.size() * sizeof(int32_t));
  }
  if (!plan->count.empty()) zero(st->d_count_acc.p, plan->count.size() * sizeof(CountAcc));
  if (!plan->como.empty()) zero(st->d_como_acc.p, plan->como.size() * sizeof(ComomentAcc));
  zero(st->d_distinct_counters.p, st->distinct.size() * kNumDistinctCounters * sizeof(unsigned long long));
  if (a.n_zero || a.n_ident) launch_state_reset(a, st->stream);
}

tgx_status tgx::state_init_device(tgx_state *st, tgx_error *err) {
  if (st->device_ready) return TGX_OK;
  TGX_TRY(need_device(err));
  const tgx_plan *plan = st->plan;
  if (!st->stream) {
    HIP_TRY(stream_acquire(&st->stream, false));  // (from the library's pool: creating one costs ~0.4 ms)
    st->own_stream = true;
  }
  // the small accumulators: allocated, then brought to their identities by ONE launch on the state's own stream (a
  // synchronous copy of the identities, a device copy and five fills cost a fresh state ~0.1 ms of host time)
  if (!plan->scan.empty()) {
    HIP_TRY(st->d_scan_acc.reserve(plan->scan.size() * sizeof(ScanAcc)));
    HIP_TRY(st->d_pivots.reserve(plan->scan.size() * sizeof(double)));
    HIP_TRY(st->d_pivot_set.reserve(plan->scan.size() * sizeof(int32_t)));
  }
  if (!plan->count.empty()) HIP_TRY(st->d_count_acc.reserve(plan->count.size() * sizeof(CountAcc)));
  if (!plan->como.empty()) HIP_TRY(st->d_como_acc.reserve(plan->como.size() * sizeof(ComomentAcc)));
  if (!plan->hll.empty()) {
    HIP_TRY(st->d_hll.reserve(plan->hll.size() * (size_t)kHllRegisters));
    HIP_TRY(hipMemsetAsync(st->d_hll.p, 0, plan->hll.size() * (size_t)kHllRegisters, st->stream));
  }
  if (!st->distinct.empty()) {
    const size_t each = kNumDistinctCounters * sizeof(unsigned long long);
    HIP_TRY(st->d_distinct_counters.reserve(st->distinct.size() * each));
    for (size_t i = 0; i < st->distinct.size(); i++)
      st->distinct[i].counters.borrow((char *)st->d_distinct_counters.p + i * each, each);
  }
  state_reset_small(st);
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
  // one wait for the whole device -- what the first hipFree of the state's buffers used to do implicitly (and every
  // further one again) -- then its device and pinned blocks go back to the cache without waiting (devcache.cpp)
  // (a state that never touched the device -- the blobs tgx_allreduce unpacks and merges -- owns nothing to wait for)
  // (without the wait no scope: should such a state own a block after all, its release waits by itself)
  const bool waited = g_ctx.inited && (st->device_ready || st->stream);
  if (waited) (void)hipDeviceSynchronize();
  std::unique_ptr<tgx::QuiescedScope> quiesced(waited ? new tgx::QuiescedScope() : nullptr);
  for (auto &kv : st->profile)
    for (auto &ev : kv.second.pending) {
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
  regex_state_free(st);
  kll_state_free(st);
  spearman_state_free(st);
  coalesce_drop(st);  // (copy threads that are still filling an arena let go of it first)
  for (int k = 0; k < 2; k++) {
    if (st->arena_event[k]) (void)hipEventDestroy(st->arena_event[k]);
    pinned_free(st->arena_host[k], kArenaBytes);
    tgx::Coalescer &co = st->coalesce;
    if (co.arena_event[k]) (void)hipEventDestroy(co.arena_event[k]);
    if (co.upload_done[k]) (void)hipEventDestroy(co.upload_done[k]);
    if (co.snap_event[k]) (void)hipEventDestroy(co.snap_event[k]);
    pinned_free(co.arena_host[k], co.arena_cap[k]);
    pinned_free(co.desc_host[k], co.desc_cap[k]);
    pinned_free(co.snap_host[k], co.snap_cap[k]);
  }
  pinned_free(st->h_pinned, st->h_pinned_cap);
  if (st->keys_ready) (void)hipEventDestroy(st->keys_ready);
  if (st->aux_done) (void)hipEventDestroy(st->aux_done);
  // (idle: the device has been waited for above) back to the pool -- hipStreamDestroy costs ~0.5 ms
  if (st->aux_stream) stream_release(st->aux_stream, true);
  if (st->batch_in) (void)hipEventDestroy(st->batch_in);
  if (st->key_stream) stream_release(st->key_stream, false);
  if (st->coalesce.copy_stream) stream_release(st->coalesce.copy_stream, false);
  if (st->own_stream && st->stream) stream_release(st->stream, false);
  delete st;
}

extern "C" tgx_status tgx_state_pending(const tgx_state *st, uint64_t *batches, uint64_t *rows) {
  if (!st) return TGX_INVALID_ARGUMENT;
  if (batches) *batches = (uint64_t)st->coalesce.batches;
  if (rows) *rows = (uint64_t)st->coalesce.rows;
  return TGX_OK;
}

extern "C" tgx_status tgx_state_sync(tgx_state *st, tgx_error *err) try {
  bind_thread();
  if (!st) return fail(err, TGX_INVALID_ARGUMENT, "state is NULL");
  TGX_TRY(coalesce_flush(st, err));  // batches tgx_update has only noted so far
  if (st->device_ready) {
    TGX_TRY(distinct_resolve_all(st, err));  // the caller may release its DEVICE batches after this call
    TGX_TRY(spearman_resolve_all(st, err));
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
    d.fp_exact_lists = false;
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
  if (st->device_ready) state_reset_small(st);
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

// ------------------------------------------------------------------------------------------------
// profiling
void prof_begin(tgx_state *st, const char *name, uint64_t bytes, hipEvent_t *e0, hipEvent_t *e1) {
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
void prof_end(tgx_state *st, const char *name, hipEvent_t e0, hipEvent_t e1) {
  if (!st->profiling || !e0) return;
  (void)hipEventRecord(e1, st->stream);
  st->profile[name].pending.emplace_back(e0, e1);
}

extern "C" tgx_status tgx_profile_enable(tgx_state *st, int32_t on) try {
  if (!st) return TGX_INVALID_ARGUMENT;
  st->profiling = on != 0;
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(nullptr);
}

void prof_resolve(tgx_state *st) {
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
// gather: the merged (device + host) view of a state

// `pre`: the counters of all tasks, already read back in one copy (gather); nullptr: read this task's now
tgx_status distinct_totals(tgx_state *st, size_t slot, DistinctTotals *t, tgx_error *err,
                                  const unsigned long long *pre) {
  DistinctState &ds = st->distinct[slot];
  unsigned long long c[kNumDistinctCounters];
  memset(c, 0, sizeof(c));
  if (pre)
    memcpy(c, pre + slot * kNumDistinctCounters, sizeof(c));
  else if (st->device_ready)
    TGX_TRY(distinct_read_counters(st, ds, c, err));
  if (c[kCntOutOfRange] != 0)
    return fail(err, TGX_INTERNAL, "distinct: %llu keys fell outside the range bitmap", c[kCntOutOfRange]);
  if (c[kCntStoreFull] != 0)
    return fail(err, TGX_INTERNAL, "distinct: %llu keys found no room in the key store of an exact set", c[kCntStoreFull]);
  const uint64_t empty_rows = c[kCntEmptyRows] + ds.h_empty_rows;
  t->total = (uint64_t)ds.total_rows + ds.h_total;
  t->non_null = c[kCntValidRows] + ds.h_non_null;
  t->distinct = c[kCntDistinct] + ds.h_distinct + (empty_rows > 0 ? 1 : 0);
  t->twice = c[kCntTwice] + ds.h_twice + (empty_rows > 1 ? 1 : 0);
  t->empty_rows = empty_rows;
  return TGX_OK;
}

tgx_status gather(tgx_state *st, Gathered *g, tgx_error *err) {
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
      // (which form the key column's last partition pass took: the next pass launches only that one)
      if (!all.empty() && (all[k * kNumDistinctCounters + kCntForm] == 1 || all[k * kNumDistinctCounters + kCntForm] == 2))
        ds.remembered_form = (int32_t)all[k * kNumDistinctCounters + kCntForm];
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

double key_to_double(int64_t k) {
  int64_t bits = f64_total_key(k);  // the transform is an involution
  double d;
  memcpy(&d, &bits, 8);
  return d;
}

double i128_to_double(uint64_t lo, int64_t hi) {
  const __int128 v = (__int128)(((unsigned __int128)(uint64_t)hi << 64) | (unsigned __int128)lo);  // (no shift of a negative value)
  return (double)v;
}

void fill_stats(const ScanAcc &a, bool variance, tgx_result *r) {
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
double hll_sigma(double x) {
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
double hll_tau(double x) {
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
uint64_t hll_estimate(const std::vector<uint8_t> &regs) {
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

