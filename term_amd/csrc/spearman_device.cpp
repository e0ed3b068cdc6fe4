// spearman_device.cpp -- keeps the (x, y) pairs of every batch on the device and ranks them at finalize.
#include "spearman_device.h"

#include "kernels/sortrank.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

namespace tgx {

void launch_spearman_compact(const ComomentColDesc &d, uint64_t *kx, uint64_t *ky, unsigned long long *count,
                             hipStream_t stream);
size_t spearman_rank_sums_bytes();
void launch_sample_sorted(const uint64_t *sorted, uint64_t n, uint32_t count, uint64_t *out, hipStream_t stream);
void launch_lower_bounds(const uint64_t *sorted, uint64_t n, const uint64_t *splitters, uint32_t k, uint64_t *out,
                         hipStream_t stream);
void launch_unsort(const uint64_t *vals, const uint32_t *perm, uint64_t n, uint64_t *out, hipStream_t stream);
int launch_rank_sums(const uint64_t *rx, const uint64_t *ry, uint64_t n, uint64_t plus, void *partials,
                     hipStream_t stream);

namespace {
struct SpearmanPlan {
  std::vector<SpearmanTask> tasks;
};
// A first batch without NULLs whose columns lie in the caller's DEVICE memory is not copied: the state keeps a view, and
// the first ranking's first pass reads the columns themselves (kernels/sortrank.h, SrSource) -- the pass that used to
// write the converted pairs (32 GB moved per 1 G pairs) is gone, and the pairs the state owns afterwards are the first
// ranking's output.  Another batch, a synchronisation or a reduction across ranks first turns the view into pairs.
struct LentBatch {
  const void *x = nullptr, *y = nullptr;  // element 0 of the batch's rows
  int64_t length = 0;
  bool x_is_float = false, y_is_float = false;
};
struct SpearmanTaskState {
  LentBatch lent;
  unsigned long long count_seed[2] = {0, 0};  // (source of a copy to `count`: lives as long as the state)
  // after the ranking of a lent batch: the pairs lie in two of the state's work arrays (and stay there while this is
  // the only task and nothing else wants the arrays: a state that is reset next never pays for a change of owners)
  DevBuf *work_x = nullptr, *work_y = nullptr;
  uint64_t work_pairs = 0;
  // the last result, until a batch arrives or the state is reset
  bool cached = false;
  uint64_t cached_pairs = 0;
  double cached_out[5];
  DevBuf kx, ky, count;
  uint64_t capacity = 0;     // pairs the buffers can hold
  uint64_t rows_upper = 0;   // host-side bound on pairs appended so far
  int64_t total_rows = 0;
  bool resolved = false;     // reduced across ranks: `res` is the answer, the pairs are history
  SpearmanResolved res;
};
struct SpearmanState {
  std::vector<SpearmanTaskState> tasks;
  // work buffers of the ranking (the two partition passes' ping-pong space, ranks, the sorter's tables): 28 bytes per
  // pair, shared by the tasks (they are ranked one after the other) and kept between calls -- allocating and freeing
  // them inside every fill_result was 160 ms of a 188 ms step at 100 M rows (hipMalloc / hipFree of 4.4 GB)
  DevBuf keys_sorted, idx, idx_sorted, heads, rx, ry, temp, partials, rank32;
  // the cross-rank ranking: this rank's sorted keys and their permutation, the keys it owns, their ranks, the ranks
  // that came back, samples / splitters / boundaries
  DevBuf loc_sorted, loc_perm, recv, recv_ranks, back, small;
  bool reducing = false;
};

tgx_status sfail(tgx_error *err, tgx_status code, const char *fmt, ...) {
  if (err) {
    err->code = code;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err->msg, sizeof(err->msg), fmt, ap);
    va_end(ap);
  }
  return code;
}
#define SHIP(expr)                                                                                         \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess)                                                                                  \
      return sfail(err, e_ == hipErrorOutOfMemory ? TGX_OUT_OF_MEMORY : TGX_DEVICE_ERROR, "%s failed: %s", \
                   #expr, hipGetErrorString(e_));                                                          \
  } while (0)

#define STRY(expr)               \
  do {                           \
    tgx_status s_ = (expr);      \
    if (s_ != TGX_OK) return s_; \
  } while (0)

const SpearmanPlan *splan(const tgx_plan *p) { return (const SpearmanPlan *)p->spearman; }
SpearmanState *sstate(tgx_state *s) { return (SpearmanState *)s->spearman; }

struct RankSumsHost {
  unsigned long long wrapped[5], exact_lo[5], exact_hi[5];
};
}  // namespace

tgx_status spearman_plan_add(tgx_plan *plan, int spec_index, int *slot, tgx_error *err) {
  if (!plan->spearman) plan->spearman = new SpearmanPlan();
  SpearmanPlan *sp = (SpearmanPlan *)plan->spearman;
  const tgx_check_spec &s = plan->specs[spec_index];
  if (s.column2 < 0) return sfail(err, TGX_INVALID_ARGUMENT, "spec %d: SPEARMAN needs column2", spec_index);
  const bool exact = (s.flags & TGX_FLAG_EXACT_RANK_SUMS) != 0;
  for (size_t i = 0; i < sp->tasks.size(); i++)
    if (sp->tasks[i].col_x == s.column && sp->tasks[i].col_y == s.column2 && sp->tasks[i].exact_sums == exact) {
      *slot = (int)i;
      return TGX_OK;
    }
  sp->tasks.push_back({s.column, s.column2, exact});
  *slot = (int)sp->tasks.size() - 1;
  return TGX_OK;
}

void spearman_plan_free(tgx_plan *plan) {
  delete (SpearmanPlan *)plan->spearman;
  plan->spearman = nullptr;
}

size_t spearman_num_tasks(const tgx_plan *plan) { return plan->spearman ? splan(plan)->tasks.size() : 0; }

void spearman_mark_used(const tgx_plan *plan, std::vector<char> &used, std::vector<char> &reads_values) {
  if (!plan->spearman) return;
  for (auto &t : splan(plan)->tasks) used[t.col_x] = used[t.col_y] = reads_values[t.col_x] = reads_values[t.col_y] = 1;
}

void spearman_state_init(tgx_state *st) {
  if (st->spearman) return;
  SpearmanState *s = new SpearmanState();
  s->tasks.resize(spearman_num_tasks(st->plan));
  st->spearman = s;
}

void spearman_state_free(tgx_state *st) {
  delete sstate(st);
  st->spearman = nullptr;
}

void spearman_state_reset(tgx_state *st) {
  SpearmanState *s = sstate(st);
  if (!s) return;
  for (auto &t : s->tasks) {
    t.rows_upper = 0;
    t.total_rows = 0;
    t.resolved = false;
    t.lent = LentBatch();
    t.work_x = t.work_y = nullptr;
    t.cached = false;
    if (t.count.p) (void)hipMemsetAsync(t.count.p, 0, 8, st->stream);
  }
}

namespace {
// the pairs of `length` rows without NULLs behind the pairs the task holds
tgx_status append_rows(tgx_state *st, SpearmanTaskState &ts, const void *xv, const void *yv, const uint8_t *xval,
                       const uint8_t *yval, int64_t xoff, int64_t yoff, int64_t length, bool x_is_float, bool y_is_float,
                       tgx_error *err) {
  if (!ts.count.p) {
    SHIP(ts.count.reserve(16));
    SHIP(hipMemsetAsync(ts.count.p, 0, 16, st->stream));
  }
  const uint64_t need = ts.rows_upper + (uint64_t)length;
  if (need > ts.capacity) {
    const uint64_t cap = std::max<uint64_t>(need, ts.capacity * 2);
    DevBuf nx, ny;
    SHIP(nx.reserve(cap * 8));
    SHIP(ny.reserve(cap * 8));
    if (ts.capacity && ts.rows_upper) {
      SHIP(hipMemcpyAsync(nx.p, ts.kx.p, ts.rows_upper * 8, hipMemcpyDeviceToDevice, st->stream));
      SHIP(hipMemcpyAsync(ny.p, ts.ky.p, ts.rows_upper * 8, hipMemcpyDeviceToDevice, st->stream));
      SHIP(hipStreamSynchronize(st->stream));
    }
    ts.kx = std::move(nx);
    ts.ky = std::move(ny);
    ts.capacity = cap;
  }
  ComomentColDesc d;
  d.x = xv;
  d.y = yv;
  d.xv = xval;
  d.yv = yval;
  d.xoff = xoff;
  d.yoff = yoff;
  d.length = length;
  d.x_is_float = x_is_float;
  d.y_is_float = y_is_float;
  launch_spearman_compact(d, ts.kx.as<uint64_t>(), ts.ky.as<uint64_t>(), ts.count.as<unsigned long long>(), st->stream);
  ts.rows_upper = need;
  return TGX_OK;
}
tgx_status resolve_lent(tgx_state *st, SpearmanTaskState &ts, tgx_error *err) {
  if (ts.work_x) {  // the pairs a ranking left in the work arrays become the state's (the arrays change owners)
    std::swap(*ts.work_x, ts.kx);
    std::swap(*ts.work_y, ts.ky);
    ts.capacity = std::min(ts.kx.cap, ts.ky.cap) / 8;
    ts.rows_upper = ts.work_pairs;
    ts.work_x = ts.work_y = nullptr;
    if (!ts.count.p) SHIP(ts.count.reserve(16));
    ts.count_seed[0] = ts.work_pairs;
    ts.count_seed[1] = 0;
    SHIP(hipMemcpyAsync(ts.count.p, ts.count_seed, 16, hipMemcpyHostToDevice, st->stream));
  }
  if (!ts.lent.length) return TGX_OK;
  const LentBatch b = ts.lent;
  ts.lent = LentBatch();
  return append_rows(st, ts, b.x, b.y, nullptr, nullptr, 0, 0, b.length, b.x_is_float, b.y_is_float, err);
}
}  // namespace

tgx_status spearman_resolve_all(tgx_state *st, tgx_error *err) {
  if (!st->spearman) return TGX_OK;
  for (auto &ts : sstate(st)->tasks) STRY(resolve_lent(st, ts, err));
  return TGX_OK;
}

tgx_status spearman_update(tgx_state *st, const tgx_column *dev, const tgx_column *columns, bool lendable,
                           tgx_error *err) {
  if (!st->plan->spearman) return TGX_OK;
  const SpearmanPlan *sp = splan(st->plan);
  SpearmanState *ss = sstate(st);
  for (size_t i = 0; i < sp->tasks.size(); i++) {
    const tgx_column &x = dev[sp->tasks[i].col_x], &y = dev[sp->tasks[i].col_y];
    const tgx_column &ox = columns[sp->tasks[i].col_x], &oy = columns[sp->tasks[i].col_y];
    SpearmanTaskState &ts = ss->tasks[i];
    auto numeric = [](int t) { return t == TGX_INT64 || t == TGX_FLOAT64; };
    if (!numeric(x.type) || !numeric(y.type))
      return sfail(err, TGX_INVALID_ARGUMENT, "SPEARMAN needs numeric columns (%d, %d)", x.type, y.type);
    if (ts.resolved)
      return sfail(err, TGX_UNSUPPORTED, "SPEARMAN: the state holds the result of a cross-rank reduction; reset it first");
    ts.total_rows += x.length;
    if (x.length == 0) continue;
    ts.cached = false;
    STRY(resolve_lent(st, ts, err));  // (a second batch: the first becomes pairs of the state's own)
    // the first batch, in the caller's own DEVICE buffers, without NULLs, large enough for the ranking to take it
    const bool lend = lendable && ts.rows_upper == 0 && !x.validity && !y.validity && ox.mem == TGX_MEM_DEVICE &&
                      oy.mem == TGX_MEM_DEVICE && x.values == ox.values && y.values == oy.values &&
                      (uint64_t)x.length >= sr_tuning().optimistic_min && sr_optimistic_applies((uint64_t)x.length);
    if (lend) {
      ts.lent.x = (const int64_t *)x.values + x.offset;
      ts.lent.y = (const int64_t *)y.values + y.offset;
      ts.lent.length = x.length;
      ts.lent.x_is_float = x.type == TGX_FLOAT64;
      ts.lent.y_is_float = y.type == TGX_FLOAT64;
      continue;
    }
    STRY(append_rows(st, ts, x.values, y.values, x.validity, y.validity, x.offset, y.offset, x.length,
                     x.type == TGX_FLOAT64, y.type == TGX_FLOAT64, err));
  }
  return TGX_OK;
}

tgx_status spearman_fill_result(tgx_state *st, int slot, tgx_result *r, tgx_error *err) {
  SpearmanTaskState &ts = sstate(st)->tasks[slot];
  const bool exact = splan(st->plan)->tasks[slot].exact_sums;
  if (ts.resolved) {
    r->total = ts.res.total_rows;
    r->non_null = (int64_t)ts.res.pairs;
    double out[5];
    for (int k = 0; k < 5; k++)
      out[k] = exact ? (double)(((unsigned __int128)ts.res.exact_hi[k] << 64) | ts.res.exact_lo[k])
                     : (double)ts.res.wrapped[k];
    r->sum_x = out[0];
    r->sum_y = out[1];
    r->sum_x2 = out[2];
    r->sum_y2 = out[3];
    r->sum_xy = out[4];
    return TGX_OK;
  }
  r->total = ts.total_rows;
  if (ts.cached) {  // (nothing arrived since the last ranking)
    r->non_null = (int64_t)ts.cached_pairs;
    const double *c = ts.cached_out;
    r->sum_x = c[0];
    r->sum_y = c[1];
    r->sum_x2 = c[2];
    r->sum_y2 = c[3];
    r->sum_xy = c[4];
    return TGX_OK;
  }
  unsigned long long m = 0;
  bool lent = ts.lent.length > 0;
  if (lent) {
    m = (unsigned long long)ts.lent.length;  // (no NULLs: every row is a pair)
  } else if (ts.count.p) {
    SHIP(hipMemcpyAsync(&m, ts.count.p, 8, hipMemcpyDeviceToHost, st->stream));
    SHIP(hipStreamSynchronize(st->stream));
  }
  r->non_null = (int64_t)m;
  if (m == 0) return TGX_OK;
  if (m > 0xFFFFFFF0ull) return sfail(err, TGX_UNSUPPORTED, "SPEARMAN over more than 2^32 rows is not supported");
  SpearmanState *ws = sstate(st);
  // RANK(x) with y as the payload, then RANK(y) with RANK(x) as the payload, the five sums taken where the second
  // ranking ends: no rank is ever scattered back to its row and nothing is laid out in order (kernels/sortrank.hip).
  DevBuf &ka = ws->keys_sorted, &pa = ws->heads, &ra = ws->idx, &rb = ws->idx_sorted, &rank32 = ws->rank32,
         &temp = ws->temp, &partials = ws->partials, &kc = ws->rx, &pc = ws->ry;
  const int blocks = sr_partials_count();
  const size_t sums_bytes = (size_t)blocks * spearman_rank_sums_bytes();
  SHIP(partials.reserve(sums_bytes + 64));
  uint32_t *d_status = (uint32_t *)((char *)partials.p + sums_bytes);  // [0] the first ranking's, [1] the second's
  const size_t temp_bytes = sr_workspace_bytes(m);
  SHIP(temp.reserve(temp_bytes));
  SHIP(rank32.reserve(m * 4));
  // Bucket sizes from the sample instead of a counting read per pass (kernels/sortrank.h): the passes' buckets then lie
  // apart, in arrays a third larger, none of them the state's own (a pass that finds a bucket full leaves nothing
  // valid behind: the pairs have to be intact for the second try, with counted buckets).  Taken when the device has
  // the room: ~57 instead of 28 bytes of work buffers per pair.
  const uint64_t roomy = sr_roomy_elems(m);
  bool optimistic = m >= sr_tuning().optimistic_min;
  if (optimistic) {
    size_t free_b = 0, total_b = 0;
    const size_t have = ka.cap + pa.cap + kc.cap + pc.cap + ra.cap + rb.cap;
    const size_t want = (size_t)roomy * 40;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (want > have && free_b < want - have + (2ull << 30)))
      optimistic = false;
  }
  if (lent && !optimistic) {  // (no room for a ranking that leaves its input alone: pairs of the state's own first)
    STRY(resolve_lent(st, ts, err));
    lent = false;
  }
  std::vector<RankSumsHost> h(blocks);
  SrPlaced at;
  for (int attempt = 0; attempt < 2; attempt++) {
    const uint64_t elems = optimistic ? roomy : m;
    SHIP(ka.reserve(elems * 8));
    SHIP(pa.reserve(elems * 8));
    SHIP(ra.reserve(elems * 4));
    SHIP(rb.reserve(elems * 4));
    if (optimistic) {
      SHIP(kc.reserve(elems * 8));
      SHIP(pc.reserve(elems * 8));
    }
    SrJob jx;
    jx.keys = lent ? (const uint64_t *)ts.lent.x : ts.kx.as<uint64_t>();
    jx.pay = lent ? ts.lent.y : ts.ky.p;
    if (lent) {  // the caller's columns, made sort keys as the first pass reads them
      jx.key_source = ts.lent.x_is_float ? kSrFloat64 : kSrInt64;
      jx.pay_source = ts.lent.y_is_float ? kSrFloat64 : kSrInt64;
    }
    jx.n = m;
    jx.pay_bytes = 8;
    jx.k[0] = ka.as<uint64_t>();
    jx.p[0] = pa.p;
    // (counted buckets: the second pass writes back into the state's own arrays, the pairs come back permuted, every
    // x still beside its y)
    jx.k[1] = optimistic ? kc.as<uint64_t>() : ts.kx.as<uint64_t>();
    jx.p[1] = optimistic ? pc.p : ts.ky.p;
    jx.optimistic = optimistic;
    jx.cap[0] = jx.cap[1] = elems;
    jx.status = d_status;
    jx.sink = kSrRank32;
    jx.rank32 = rank32.as<uint32_t>();
    SHIP(sr_run(jx, temp.p, temp_bytes, st->stream, &at));
    SrJob jy;
    jy.keys = (const uint64_t *)at.pay;  // the y keys, slot for slot beside rank32
    jy.pay = rank32.p;
    jy.n = m;
    jy.pay_bytes = 4;
    // the first ranking is through: its arrays are free again, but (with room from the sample) the one the y keys lie in
    uint64_t *spare[4] = {ka.as<uint64_t>(), pa.as<uint64_t>(), optimistic ? kc.as<uint64_t>() : nullptr,
                          optimistic ? pc.as<uint64_t>() : nullptr};
    int took = 0;
    for (int c = 0; c < 4 && took < 2; c++)
      if (spare[c] && (!optimistic || (const void *)spare[c] != at.pay) && (!lent || spare[c] != at.keys))
        jy.k[took++] = spare[c];  // (a lent batch: the x keys are kept too -- the state's pairs from now on)
    // (counted buckets: k[1] may be where the y keys lie now -- they are read by the first pass only)
    jy.p[0] = ra.p;
    jy.p[1] = rb.p;
    jy.optimistic = optimistic;
    jy.cap[0] = jy.cap[1] = elems;
    jy.status = d_status + 1;
    jy.sink = kSrSums;
    jy.partials = (RankSums *)partials.p;
    SHIP(sr_run(jy, temp.p, temp_bytes, st->stream, nullptr));
    uint32_t status[2] = {0, 0};
    SHIP(hipMemcpyAsync(h.data(), partials.p, blocks * sizeof(RankSumsHost), hipMemcpyDeviceToHost, st->stream));
    SHIP(hipMemcpyAsync(status, d_status, sizeof(status), hipMemcpyDeviceToHost, st->stream));
    SHIP(hipStreamSynchronize(st->stream));
    if ((status[0] | status[1]) == 0) break;
    if (!optimistic) return sfail(err, TGX_INTERNAL, "SPEARMAN: the ranking failed with counted buckets (status %u, %u)",
                                  status[0], status[1]);
    optimistic = false;  // a bucket outgrew the room its share of the sample gave it: once more, counting
    if (lent) {
      STRY(resolve_lent(st, ts, err));
      lent = false;
    }
    if (getenv("TGX_SORT_DEBUG")) fprintf(stderr, "tgx sort: a bucket was full (status %u, %u): again with counted buckets\n", status[0], status[1]);
  }
  if (lent) {
    // the first ranking left every x beside its y, as sort keys, in two of the work arrays: the state's pairs from now
    // on, the view of the caller's batch is dropped
    DevBuf *work[4] = {&ka, &pa, &kc, &pc}, *bx = nullptr, *by = nullptr;
    for (DevBuf *b : work) {
      if (b->p == (const void *)at.keys) bx = b;
      if (b->p == at.pay) by = b;
    }
    if (!bx || !by || bx == by) return sfail(err, TGX_INTERNAL, "SPEARMAN: the ranking's output is not where it was expected");
    ts.work_x = bx;
    ts.work_y = by;
    ts.work_pairs = m;
    ts.rows_upper = m;
    ts.lent = LentBatch();
    // (another task ranks in the same arrays next)
    if (sstate(st)->tasks.size() > 1) STRY(resolve_lent(st, ts, err));
  }
  double out[5];
  for (int k = 0; k < 5; k++) {
    unsigned long long w = 0;
    unsigned __int128 e = 0;
    for (auto &p : h) {
      w += p.wrapped[k];
      e += ((unsigned __int128)p.exact_hi[k] << 64) | p.exact_lo[k];
    }
    out[k] = exact ? (double)e : (double)w;
    ts.cached_out[k] = out[k];
  }
  ts.cached = true;
  ts.cached_pairs = m;
  r->sum_x = out[0];
  r->sum_y = out[1];
  r->sum_x2 = out[2];
  r->sum_y2 = out[3];
  r->sum_xy = out[4];
  return TGX_OK;
}

tgx_status spearman_check_mergeable(tgx_state *st, tgx_error *err) {
  if (!st->spearman || sstate(st)->reducing) return TGX_OK;
  for (auto &t : sstate(st)->tasks)
    if (t.rows_upper > 0 || t.total_rows > 0 || t.resolved)
      return sfail(err, TGX_UNSUPPORTED,
                   "Spearman states hold ranks of one data set and cannot be merged or serialized "
                   "(as in the reference, analyzers/advanced/correlation.rs:103-109)");
  return TGX_OK;
}

// ---- across ranks ---------------------------------------------------------------------------------------------
namespace {
constexpr uint32_t kSamplesPerRank = 1024;

// ranks of `keys` (m of them, this rank's) among ALL ranks' keys -> out[i] for keys[i]
tgx_status rank_across(tgx_state *st, SpearmanState *ws, const SpearmanExchange &X, uint64_t *keys, uint64_t m,
                       uint64_t *out, tgx_error *err) {
  const int32_t W = X.world, R = X.rank;
  hipStream_t s = st->device_ready ? st->stream : nullptr;
  size_t tb = 0;
  // (a) this rank's keys in order, with the permutation that leads back to the rows (the keys' positions travel as the
  //     payload; the state's own arrays are only read: x and y must stay side by side)
  if (m) {
    SHIP(ws->loc_sorted.reserve(m * 8));
    SHIP(ws->loc_perm.reserve(m * 4));
    SHIP(ws->keys_sorted.reserve(m * 8));
    SHIP(ws->heads.reserve(m * 8));
    SHIP(ws->idx.reserve(m * 4));
    SHIP(ws->idx_sorted.reserve(m * 4));
    tb = sr_workspace_bytes(m);
    SHIP(ws->temp.reserve(tb));
    SrJob j;
    j.keys = keys;
    j.n = m;
    j.pay_bytes = 4;  // (no payload array: the index)
    j.k[0] = ws->keys_sorted.as<uint64_t>();
    j.p[0] = ws->idx.p;
    j.k[1] = ws->heads.as<uint64_t>();
    j.p[1] = ws->idx_sorted.p;
    j.sink = kSrSorted;
    j.out_keys = ws->loc_sorted.as<uint64_t>();
    j.out_pay = ws->loc_perm.p;
    SHIP(sr_run(j, ws->temp.p, tb, s, nullptr));
  }
  // (b) regular samples of every rank -> the same world-1 splitters everywhere
  std::vector<uint64_t> mine(1 + kSamplesPerRank, 0), all((size_t)W * (1 + kSamplesPerRank), 0);
  const uint32_t ns = (uint32_t)std::min<uint64_t>(m, kSamplesPerRank);
  mine[0] = ns;
  if (ns) {
    SHIP(ws->small.reserve((2 * kSamplesPerRank + 512) * 8));
    launch_sample_sorted(ws->loc_sorted.as<uint64_t>(), m, ns, ws->small.as<uint64_t>(), s);
    SHIP(hipMemcpyAsync(&mine[1], ws->small.p, ns * 8, hipMemcpyDeviceToHost, s));
    SHIP(hipStreamSynchronize(s));
  }
  STRY(X.allgather_host(mine.data(), all.data(), mine.size() * 8));
  std::vector<uint64_t> pool;
  for (int32_t r = 0; r < W; r++) {
    const uint64_t *p = &all[(size_t)r * (1 + kSamplesPerRank)];
    if (p[0] > kSamplesPerRank) return sfail(err, TGX_INTERNAL, "SPEARMAN exchange: bad sample header from rank %d", r);
    pool.insert(pool.end(), p + 1, p + 1 + p[0]);
  }
  std::sort(pool.begin(), pool.end());
  std::vector<uint64_t> split((size_t)std::max(W - 1, 0), 0);
  for (int32_t j = 0; j + 1 < W; j++) split[j] = pool.empty() ? 0 : pool[(size_t)(j + 1) * pool.size() / W];
  // (c) rank q owns the keys with exactly q splitters <= key: contiguous stretches of the sorted keys
  std::vector<uint64_t> sc((size_t)W, 0);
  if (m && W > 1) {
    uint64_t *d_split = ws->small.as<uint64_t>() + kSamplesPerRank, *d_bound = d_split + 256;
    if (W - 1 > 256) return sfail(err, TGX_UNSUPPORTED, "SPEARMAN across more than 257 ranks");
    std::vector<uint64_t> bound((size_t)W - 1);
    SHIP(hipMemcpyAsync(d_split, split.data(), split.size() * 8, hipMemcpyHostToDevice, s));
    launch_lower_bounds(ws->loc_sorted.as<uint64_t>(), m, d_split, (uint32_t)(W - 1), d_bound, s);
    SHIP(hipMemcpyAsync(bound.data(), d_bound, bound.size() * 8, hipMemcpyDeviceToHost, s));
    SHIP(hipStreamSynchronize(s));
    uint64_t prev = 0;
    for (int32_t j = 0; j + 1 < W; j++) {
      sc[j] = bound[j] - prev;
      prev = bound[j];
    }
    sc[W - 1] = m - prev;
  } else if (m) {
    sc[0] = m;
  }
  std::vector<uint64_t> mat((size_t)W * W, 0), rc((size_t)W, 0);
  STRY(X.allgather_host(sc.data(), mat.data(), (size_t)W * 8));  // mat[p * W + q]: p sends q
  uint64_t M = 0, base = 0, everywhere = 0;
  for (int32_t p = 0; p < W; p++) {
    rc[p] = mat[(size_t)p * W + R];
    M += rc[p];
    for (int32_t q = 0; q < R; q++) base += mat[(size_t)p * W + q];
    for (int32_t q = 0; q < W; q++) everywhere += mat[(size_t)p * W + q];
  }
  if (everywhere == 0) return TGX_OK;  // no rank holds a pair (every rank sees that): nothing to exchange
  if (M > 0xFFFFFFF0ull)
    return sfail(err, TGX_UNSUPPORTED, "SPEARMAN: more than 2^32 keys fall to one rank (heavily repeated values)");
  if (M || m) {  // (a rank that saw no batch may still own a value range)
    STRY(need_device(err));
    STRY(state_init_device(st, err));
    s = st->stream;
  }
  // (d) the keys to their owners, ranked there, the ranks back
  SHIP(ws->recv.reserve(std::max<uint64_t>(M, 1) * 8));
  SHIP(ws->loc_sorted.reserve(8));  // (a rank without keys still names a buffer)
  STRY(X.alltoallv(ws->loc_sorted.p, sc.data(), ws->recv.p, rc.data(), 8));
  SHIP(ws->recv_ranks.reserve(std::max<uint64_t>(M, 1) * 8));
  if (M) {
    // RANK() of a received key = the keys lower ranks own + its rank among the keys owned here, written to the slot
    // the key arrived in
    SHIP(ws->keys_sorted.reserve(M * 8));
    SHIP(ws->heads.reserve(M * 8));
    SHIP(ws->idx.reserve(M * 4));
    SHIP(ws->idx_sorted.reserve(M * 4));
    tb = sr_workspace_bytes(M);
    SHIP(ws->temp.reserve(tb));
    SrJob j;
    j.keys = ws->recv.as<uint64_t>();
    j.n = M;
    j.pay_bytes = 4;
    j.k[0] = ws->keys_sorted.as<uint64_t>();
    j.p[0] = ws->idx.p;
    j.k[1] = ws->heads.as<uint64_t>();
    j.p[1] = ws->idx_sorted.p;
    j.sink = kSrRankScatter;
    j.rank_out = ws->recv_ranks.as<uint64_t>();
    j.ext_base = base;
    SHIP(sr_run(j, ws->temp.p, tb, s, nullptr));
  }
  SHIP(ws->back.reserve(std::max<uint64_t>(m, 1) * 8));
  STRY(X.alltoallv(ws->recv_ranks.p, rc.data(), ws->back.p, sc.data(), 8));
  if (m) launch_unsort(ws->back.as<uint64_t>(), ws->loc_perm.as<uint32_t>(), m, out, s);
  return TGX_OK;
}
}  // namespace

tgx_status spearman_allreduce(tgx_state *st, const SpearmanExchange &X, std::vector<SpearmanResolved> *out,
                              tgx_error *err) {
  out->clear();
  if (!st->plan->spearman) return TGX_OK;
  STRY(spearman_resolve_all(st, err));  // (the exchange reads the state's own pairs)
  const SpearmanPlan *sp = splan(st->plan);
  SpearmanState *ws = sstate(st);
  hipStream_t s = st->device_ready ? st->stream : nullptr;
  struct Wire {
    int64_t total_rows;
    uint64_t pairs;
    RankSumsHost sums;
  };
  for (size_t t = 0; t < sp->tasks.size(); t++) {
    SpearmanTaskState &ts = ws->tasks[t];
    if (ts.resolved)
      return sfail(err, TGX_UNSUPPORTED, "SPEARMAN: the state already holds the result of a cross-rank reduction");
    unsigned long long m = 0;
    if (ts.count.p) {
      SHIP(hipMemcpyAsync(&m, ts.count.p, 8, hipMemcpyDeviceToHost, s));
      SHIP(hipStreamSynchronize(s));
    }
    if (m > 0xFFFFFFF0ull) return sfail(err, TGX_UNSUPPORTED, "SPEARMAN over more than 2^32 rows per rank is not supported");
    if (m) {
      SHIP(ws->rx.reserve(m * 8));
      SHIP(ws->ry.reserve(m * 8));
    }
    STRY(rank_across(st, ws, X, ts.kx.as<uint64_t>(), m, ws->rx.as<uint64_t>(), err));
    STRY(rank_across(st, ws, X, ts.ky.as<uint64_t>(), m, ws->ry.as<uint64_t>(), err));
    Wire mine;
    memset(&mine, 0, sizeof(mine));
    mine.total_rows = ts.total_rows;
    mine.pairs = m;
    if (m) {
      SHIP(ws->partials.reserve(2048 * spearman_rank_sums_bytes()));
      const int blocks = launch_rank_sums(ws->rx.as<uint64_t>(), ws->ry.as<uint64_t>(), m, 0, ws->partials.p, s);
      std::vector<RankSumsHost> h(blocks);
      SHIP(hipMemcpyAsync(h.data(), ws->partials.p, blocks * sizeof(RankSumsHost), hipMemcpyDeviceToHost, s));
      SHIP(hipStreamSynchronize(s));
      for (int k = 0; k < 5; k++) {
        unsigned __int128 e = 0;
        for (auto &p : h) {
          mine.sums.wrapped[k] += p.wrapped[k];
          e += ((unsigned __int128)p.exact_hi[k] << 64) | p.exact_lo[k];
        }
        mine.sums.exact_lo[k] = (unsigned long long)e;
        mine.sums.exact_hi[k] = (unsigned long long)(e >> 64);
      }
    }
    std::vector<Wire> all((size_t)X.world);
    STRY(X.allgather_host(&mine, all.data(), sizeof(Wire)));
    SpearmanResolved res;
    memset(&res, 0, sizeof(res));
    for (auto &w : all) {
      res.total_rows += w.total_rows;
      res.pairs += w.pairs;
      for (int k = 0; k < 5; k++) {
        res.wrapped[k] += w.sums.wrapped[k];
        const unsigned __int128 e = (((unsigned __int128)res.exact_hi[k] << 64) | res.exact_lo[k]) +
                                    (((unsigned __int128)w.sums.exact_hi[k] << 64) | w.sums.exact_lo[k]);
        res.exact_lo[k] = (unsigned long long)e;
        res.exact_hi[k] = (unsigned long long)(e >> 64);
      }
    }
    out->push_back(res);
  }
  return TGX_OK;
}

void spearman_install(tgx_state *st, const std::vector<SpearmanResolved> &res) {
  SpearmanState *ws = sstate(st);
  if (!ws) return;
  for (size_t t = 0; t < res.size() && t < ws->tasks.size(); t++) {
    ws->tasks[t].resolved = true;
    ws->tasks[t].res = res[t];
  }
}

void spearman_set_reducing(tgx_state *st, bool on) {
  if (sstate(st)) sstate(st)->reducing = on;
}

}  // namespace tgx
