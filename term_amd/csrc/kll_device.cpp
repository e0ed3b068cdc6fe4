// kll_device.cpp -- KLL tasks of a state: device sketching (kernels/kll.hip) + host-side merge/query.
#include "kll_device.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "kernels/kll_types.h"

namespace tgx {

void launch_kll_init(KllDeviceSketch *s, hipStream_t stream, uint32_t shift);
void launch_kll_jobs(const KllJob *jobs, int n_jobs, hipStream_t stream);
void launch_kll_meta(const KllWaveMeta *meta, int n_waves, KllDeviceSketch *state, hipStream_t stream);
uint32_t kll_top_for(int64_t rows);

namespace {
// what the scan leaves behind for a task whose sampler rode on it (kll_types.h, ScanKll)
struct KllScanBuffers {
  DevBuf picks, left, meta;
  int64_t n_picks = 0, n_left = 0;
  int n_waves = 0;
  uint32_t top = 0;
  bool pending = false;  // prepared for the batch being queued: kll_scan_finish sketches it
};

struct KllDeviceState {
  std::vector<DevBuf> sketch;  // running device sketch per task
  std::vector<char> dirty;     // device sketch holds data not yet folded into h_kll
  // running sketch of pre-sampled values (the picks of the fused scan): loose items weigh 2^hi_shift
  std::vector<DevBuf> sketch_hi;
  std::vector<uint32_t> hi_shift;
  std::vector<char> dirty_hi;
  std::vector<KllScanBuffers> scan;
  DevBuf scratch;              // per-workgroup sketches of the batch being processed
  uint64_t salt = 0x6b6c6c5f74677821ULL;
};

tgx_status kfail(tgx_error *err, tgx_status code, const char *fmt, ...) {
  if (err) {
    err->code = code;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err->msg, sizeof(err->msg), fmt, ap);
    va_end(ap);
  }
  return code;
}
#define KHIP(expr)                                                                                   \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      return kfail(err, e_ == hipErrorOutOfMemory ? TGX_OUT_OF_MEMORY : TGX_DEVICE_ERROR, "%s failed: %s", \
                   #expr, hipGetErrorString(e_));                                                    \
  } while (0)

KllDeviceState *dev(tgx_state *st) { return (KllDeviceState *)st->kll; }
}  // namespace

void kll_state_init(tgx_state *st) {
  if (st->kll) return;
  KllDeviceState *k = new KllDeviceState();
  k->sketch.resize(st->plan->kll.size());
  k->dirty.assign(st->plan->kll.size(), 0);
  k->sketch_hi.resize(st->plan->kll.size());
  k->hi_shift.assign(st->plan->kll.size(), 0);
  k->dirty_hi.assign(st->plan->kll.size(), 0);
  k->scan.resize(st->plan->kll.size());
  st->kll = k;
}

void kll_state_free(tgx_state *st) {
  delete dev(st);
  st->kll = nullptr;
}

void kll_state_reset(tgx_state *st) {
  KllDeviceState *k = dev(st);
  if (!k) return;
  for (size_t i = 0; i < k->sketch.size(); i++) {
    if (k->sketch[i].p) launch_kll_init(k->sketch[i].as<KllDeviceSketch>(), st->stream, 0);
    if (k->sketch_hi[i].p) launch_kll_init(k->sketch_hi[i].as<KllDeviceSketch>(), st->stream, k->hi_shift[i]);
    k->dirty[i] = k->dirty_hi[i] = 0;
    k->scan[i].pending = false;
  }
}

// groups / chunk of a build over `n` pre-sampled values: every 1024 of them are sorted by one workgroup, ~35 us a
// time, so the work is spread wide (8 Ki values per workgroup, at most 1024 workgroups)
static void build_shape(int64_t n, int64_t *groups, int64_t *chunk) {
  int64_t g = (n + 8191) / 8192;
  g = std::max<int64_t>(1, std::min<int64_t>(g, 1024));
  int64_t c = (n + g - 1) / g;
  c = (c + 4095) / 4096 * 4096;
  *groups = (n + c - 1) / c;
  *chunk = c;
}

bool kll_scan_eligible(int64_t rows) { return kll_top_for(rows) >= 1; }

tgx_status kll_scan_prepare(tgx_state *st, size_t slot, int64_t rows, int n_waves, int64_t max_rows_per_wave,
                            ScanKll *out, tgx_error *err) {
  KllDeviceState *k = dev(st);
  KllScanBuffers &b = k->scan[slot];
  // one level above the stand-alone kernel's: half the picks to sketch afterwards (the picks of a 1 G-row column are
  // 3.9 M values at top = 8), for a sampling error that stays an order below the sketch's own (kll.hip)
  const uint32_t top = std::min<uint32_t>(kll_top_for(rows) + 1, (uint32_t)kScanKllMaxTop);
  const int64_t cap = ((max_rows_per_wave + 1024) >> top) + 2;
  b.n_waves = n_waves;
  b.top = top;
  b.n_picks = (int64_t)n_waves * cap;
  b.n_left = (int64_t)n_waves << top;
  KHIP(b.picks.reserve((size_t)b.n_picks * 8 + 64));
  KHIP(b.left.reserve((size_t)b.n_left * 8 + 64));
  KHIP(b.meta.reserve((size_t)n_waves * sizeof(KllWaveMeta)));
  k->salt = k->salt * 6364136223846793005ULL + 1442695040888963407ULL;
  out->picks = b.picks.as<double>();
  out->left = b.left.as<double>();
  out->meta = b.meta.as<KllWaveMeta>();
  out->salt = k->salt ^ 0x7363616e5f6b6c6cULL;
  out->top = (int32_t)top;
  out->cap = (int32_t)cap;
  b.pending = true;
  return TGX_OK;
}

// the scan of the batch has been queued: sketch what it leaves -- the leftovers (weight 1) into the running sketch,
// the picks (weight 2^top) into the running sketch of that shift.  All tasks of the batch side by side.
tgx_status kll_scan_finish(tgx_state *st, tgx_error *err) {
  KllDeviceState *k = dev(st);
  if (!k) return TGX_OK;
  std::vector<size_t> slots;
  for (size_t i = 0; i < k->scan.size(); i++)
    if (k->scan[i].pending) slots.push_back(i);
  if (slots.empty()) return TGX_OK;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (st->profiling && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess)
    (void)hipEventRecord(e0, st->stream);
  std::vector<KllJob> jobs;
  std::vector<int64_t> scratch_at;
  int64_t scratch_total = 0;
  uint64_t bytes = 0;
  for (size_t slot : slots) {
    KllScanBuffers &b = k->scan[slot];
    b.pending = false;
    if (!k->sketch[slot].p) {
      KHIP(k->sketch[slot].reserve(sizeof(KllDeviceSketch)));
      launch_kll_init(k->sketch[slot].as<KllDeviceSketch>(), st->stream, 0);
    }
    if (k->sketch_hi[slot].p && k->dirty_hi[slot] && k->hi_shift[slot] != b.top) {
      // a batch of another size class: its picks weigh differently -- hand the old ones to the host first
      tgx_status s = kll_flush(st, err);
      if (s != TGX_OK) return s;
    }
    if (!k->sketch_hi[slot].p || k->hi_shift[slot] != b.top) {
      KHIP(k->sketch_hi[slot].reserve(sizeof(KllDeviceSketch)));
      k->hi_shift[slot] = b.top;
      launch_kll_init(k->sketch_hi[slot].as<KllDeviceSketch>(), st->stream, b.top);
    }
    for (int hi = 0; hi < 2; hi++) {
      KllJob job;
      memset(&job, 0, sizeof(job));
      int64_t groups, chunk;
      build_shape(hi ? b.n_picks : b.n_left, &groups, &chunk);
      job.d.is_float = 1;
      job.d.values = hi ? b.picks.p : b.left.p;
      job.d.length = hi ? b.n_picks : b.n_left;
      job.chunk = chunk;
      job.state = (hi ? k->sketch_hi[slot] : k->sketch[slot]).as<KllDeviceSketch>();
      job.salt = k->salt ^ (0x1111ULL * (uint64_t)(2 * slot + hi + 1));
      job.top = 0;                 // the values are the sample: none of them is dropped again
      job.shift = hi ? b.top : 0;
      job.groups = (int32_t)groups;
      scratch_at.push_back(scratch_total);
      scratch_total += groups;
      jobs.push_back(job);
      bytes += (uint64_t)job.d.length * 8;
    }
    k->dirty[slot] = k->dirty_hi[slot] = 1;
  }
  KHIP(k->scratch.reserve((size_t)scratch_total * sizeof(KllDeviceSketch)));
  for (size_t j = 0; j < jobs.size(); j++) jobs[j].sketches = k->scratch.as<KllDeviceSketch>() + scratch_at[j];
  launch_kll_jobs(jobs.data(), (int)jobs.size(), st->stream);
  for (size_t slot : slots) {
    KllScanBuffers &b = k->scan[slot];
    // (into both: either of them may end up without items -- no leftovers, or no complete group -- and an empty sketch
    // is skipped by the merge)
    launch_kll_meta(b.meta.as<KllWaveMeta>(), b.n_waves, k->sketch[slot].as<KllDeviceSketch>(), st->stream);
    launch_kll_meta(b.meta.as<KllWaveMeta>(), b.n_waves, k->sketch_hi[slot].as<KllDeviceSketch>(), st->stream);
  }
  if (st->profiling && e0 && e1) {
    (void)hipEventRecord(e1, st->stream);
    ProfileEntry &pe = st->profile["kll"];
    pe.pending.emplace_back(e0, e1);
    pe.pending_bytes.push_back(bytes);
  }
  return TGX_OK;
}

tgx_status kll_update(tgx_state *st, size_t slot, const tgx_column &c, tgx_error *err) {
  KllDeviceState *k = dev(st);
  if (c.type != TGX_INT64 && c.type != TGX_FLOAT64)
    return kfail(err, TGX_UNSUPPORTED, "KLL needs a numeric column (type %d)", c.type);
  if (c.length == 0) return TGX_OK;
  if (!k->sketch[slot].p) {
    KHIP(k->sketch[slot].reserve(sizeof(KllDeviceSketch)));
    launch_kll_init(k->sketch[slot].as<KllDeviceSketch>(), st->stream, 0);
  }
  // one workgroup per >= 64 Ki rows, at most 1024 of them (235 MiB of scratch sketches)
  int64_t groups = (c.length + 65535) / 65536;
  groups = std::max<int64_t>(1, std::min<int64_t>(groups, 1024));
  int64_t chunk = (c.length + groups - 1) / groups;
  chunk = (chunk + 4095) / 4096 * 4096;
  groups = (c.length + chunk - 1) / chunk;
  KHIP(k->scratch.reserve((size_t)groups * sizeof(KllDeviceSketch)));
  KllColDesc d;
  d.values = c.values;
  d.validity = c.validity;
  d.offset = c.offset;
  d.length = c.length;
  d.is_float = c.type == TGX_FLOAT64;
  d.pad = 0;
  k->salt = k->salt * 6364136223846793005ULL + 1442695040888963407ULL;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (st->profiling && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
    (void)hipEventRecord(e0, st->stream);
  }
  KllJob job;
  memset(&job, 0, sizeof(job));
  job.d = d;
  job.chunk = chunk;
  job.sketches = k->scratch.as<KllDeviceSketch>();
  job.state = k->sketch[slot].as<KllDeviceSketch>();
  job.salt = k->salt;
  job.top = kll_top_for(d.length);
  job.groups = (int32_t)groups;
  launch_kll_jobs(&job, 1, st->stream);
  if (st->profiling && e0 && e1) {
    (void)hipEventRecord(e1, st->stream);
    ProfileEntry &pe = st->profile["kll"];
    pe.pending.emplace_back(e0, e1);
    pe.pending_bytes.push_back((uint64_t)c.length * 8 + (c.validity ? (uint64_t)(c.length + 7) / 8 : 0));
  }
  k->dirty[slot] = 1;
  return TGX_OK;
}

tgx_status kll_flush(tgx_state *st, tgx_error *err) {
  KllDeviceState *k = dev(st);
  if (!k) return TGX_OK;
  for (size_t i = 0; i < k->sketch.size(); i++) {
    for (int hi = 0; hi < 2; hi++) {
      DevBuf &buf = hi ? k->sketch_hi[i] : k->sketch[i];
      char &dirty = hi ? k->dirty_hi[i] : k->dirty[i];
      if (!dirty) continue;
      std::vector<uint8_t> raw(sizeof(KllDeviceSketch));
      KHIP(hipMemcpyAsync(raw.data(), buf.p, raw.size(), hipMemcpyDeviceToHost, st->stream));
      KHIP(hipStreamSynchronize(st->stream));
      const KllDeviceSketch *s = (const KllDeviceSketch *)raw.data();
      KllHost part;
      part.k = st->h_kll[i].k;
      part.n = s->n;
      part.min_v = s->min_v;
      part.max_v = s->max_v;
      // loose items weigh 2^shift (a sketch of pre-sampled picks), the runs were stored at their final levels
      if (s->lv0_count) part.add_level_items(s->shift, s->lv0, s->lv0_count);
      for (int l = 1; l < kKllMaxLevels; l++)
        if ((s->level_mask >> l) & 1) part.add_level_items((size_t)l, s->runs[l], kKllRunItems);
      if (!st->h_kll[i].merge(part)) return kfail(err, TGX_INTERNAL, "KLL merge failed");
      launch_kll_init(buf.as<KllDeviceSketch>(), st->stream, hi ? k->hi_shift[i] : 0);
      dirty = 0;
    }
  }
  return TGX_OK;
}

tgx_status kll_fill_result(tgx_state *st, int slot, tgx_result *r, tgx_error *err) {
  tgx_status s = kll_flush(st, err);
  if (s != TGX_OK) return s;
  r->kll_n = st->h_kll[slot].n;
  r->non_null = (int64_t)st->h_kll[slot].n;
  return TGX_OK;
}

tgx_status kll_merge_states(tgx_state *dst, tgx_state *src, tgx_error *err) {
  tgx_status s = kll_flush(dst, err);
  if (s != TGX_OK) return s;
  s = kll_flush(src, err);
  if (s != TGX_OK) return s;
  for (size_t i = 0; i < dst->h_kll.size(); i++)
    if (!dst->h_kll[i].merge(src->h_kll[i]))
      return kfail(err, TGX_INVALID_ARGUMENT, "Cannot merge sketches with different k values: %u vs %u",
                   dst->h_kll[i].k, src->h_kll[i].k);
  return TGX_OK;
}

// wire: per task  u32 k, u32 n_levels, u64 n, f64 min, f64 max, then per level u32 count + items
tgx_status kll_serialize(tgx_state *st, size_t *len, uint8_t *buf, size_t cap, tgx_error *err) {
  tgx_status s = kll_flush(st, err);
  if (s != TGX_OK) return s;
  auto put = [&](const void *p, size_t n) {
    if (n && buf && *len + n <= cap) memcpy(buf + *len, p, n);  // (an empty level has no items: p may be null)
    *len += n;
  };
  for (auto &h : st->h_kll) {
    uint32_t k = h.k, nl = (uint32_t)h.levels.size();
    put(&k, 4);
    put(&nl, 4);
    put(&h.n, 8);
    put(&h.min_v, 8);
    put(&h.max_v, 8);
    for (auto &lv : h.levels) {
      uint32_t c = (uint32_t)lv.size();
      put(&c, 4);
      put(lv.data(), (size_t)c * 8);
    }
  }
  return TGX_OK;
}

tgx_status kll_deserialize(tgx_state *st, const uint8_t *buf, size_t len, size_t *pos, tgx_error *err) {
  auto get = [&](void *p, size_t n) -> bool {
    if (*pos + n > len) return false;
    if (n) memcpy(p, buf + *pos, n);  // (an empty level: p may be null)
    *pos += n;
    return true;
  };
  for (auto &h : st->h_kll) {
    uint32_t k = 0, nl = 0;
    if (!get(&k, 4) || !get(&nl, 4) || !get(&h.n, 8) || !get(&h.min_v, 8) || !get(&h.max_v, 8))
      return kfail(err, TGX_INVALID_ARGUMENT, "truncated state blob (kll)");
    if (k != h.k) return kfail(err, TGX_INVALID_ARGUMENT, "state blob was produced with k=%u, plan has k=%u", k, h.k);
    if (nl > 64) return kfail(err, TGX_INVALID_ARGUMENT, "corrupt state blob (kll levels)");
    h.levels.assign(nl, {});
    for (uint32_t l = 0; l < nl; l++) {
      uint32_t c = 0;
      if (!get(&c, 4) || c > (1u << 20)) return kfail(err, TGX_INVALID_ARGUMENT, "truncated state blob (kll)");
      h.levels[l].resize(c);
      if (!get(h.levels[l].data(), (size_t)c * 8)) return kfail(err, TGX_INVALID_ARGUMENT, "truncated state blob (kll)");
    }
  }
  return TGX_OK;
}

static tgx_status kll_slot(const tgx_plan *plan, tgx_state *st, size_t spec_index, int *slot, tgx_error *err) {
  if (!plan || !st || st->plan != plan) return kfail(err, TGX_INVALID_ARGUMENT, "state does not belong to plan");
  if (spec_index >= plan->specs.size() || plan->specs[spec_index].kind != TGX_CHECK_KLL)
    return kfail(err, TGX_INVALID_ARGUMENT, "spec %zu is not a KLL check", spec_index);
  *slot = plan->bind[spec_index].slot;
  tgx_status cs = coalesce_flush(st, err);  // batches tgx_update has only noted so far
  if (cs != TGX_OK) return cs;
  return kll_flush(st, err);
}

}  // namespace tgx

using namespace tgx;

extern "C" tgx_status tgx_kll_quantile(const tgx_plan *plan, tgx_state *st, size_t spec_index, double phi,
                                       double *out, tgx_error *err) try {
  bind_thread();
  int slot = 0;
  tgx_status s = kll_slot(plan, st, spec_index, &slot, err);
  if (s != TGX_OK) return s;
  if (!out) return kfail(err, TGX_INVALID_ARGUMENT, "out is NULL");
  int rc = st->h_kll[slot].quantile(phi, out);
  // messages of KllSketch::get_quantile (kll_sketch.rs:247-257)
  if (rc == 1) return kfail(err, TGX_INVALID_ARGUMENT, "Cannot compute quantile on empty sketch");
  if (rc == 2) return kfail(err, TGX_INVALID_ARGUMENT, "Quantile phi must be in [0, 1], got %g", phi);
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

extern "C" tgx_status tgx_kll_summary(const tgx_plan *plan, tgx_state *st, size_t spec_index, uint64_t *n,
                                      double *min_value, double *max_value, uint64_t *num_levels,
                                      uint64_t *num_retained, tgx_error *err) try {
  bind_thread();
  int slot = 0;
  tgx_status s = kll_slot(plan, st, spec_index, &slot, err);
  if (s != TGX_OK) return s;
  const KllHost &h = st->h_kll[slot];
  if (n) *n = h.n;
  if (min_value) *min_value = h.min_v;
  if (max_value) *max_value = h.max_v;
  if (num_levels) *num_levels = h.levels.size();
  if (num_retained) *num_retained = h.retained();
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

extern "C" tgx_status tgx_kll_level_items(const tgx_plan *plan, tgx_state *st, size_t spec_index,
                                          uint64_t level, double *out, uint64_t cap, uint64_t *count,
                                          tgx_error *err) try {
  bind_thread();
  int slot = 0;
  tgx_status s = kll_slot(plan, st, spec_index, &slot, err);
  if (s != TGX_OK) return s;
  const KllHost &h = st->h_kll[slot];
  uint64_t c = level < h.levels.size() ? h.levels[level].size() : 0;
  if (count) *count = c;
  if (out)
    for (uint64_t i = 0; i < c && i < cap; i++) out[i] = h.levels[level][i];
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

extern "C" double tgx_kll_relative_error_bound(uint32_t k) { return 1.65 / sqrt((double)k); }
