// spearman_device.cpp -- keeps the (x, y) pairs of every batch on the device and ranks them at finalize.
#include "spearman_device.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

namespace tgx {

void launch_spearman_compact(const ComomentColDesc &d, uint64_t *kx, uint64_t *ky, unsigned long long *count,
                             hipStream_t stream);
size_t spearman_rank_sums_bytes();
hipError_t spearman_rank(uint64_t *keys, uint64_t n, uint64_t *keys_sorted, uint32_t *idx, uint32_t *idx_sorted,
                         uint64_t *heads, uint64_t *rank, void *temp, size_t *temp_bytes, hipStream_t stream);
int launch_rank_sums(const uint64_t *rx, const uint64_t *ry, uint64_t n, void *partials, hipStream_t stream);

namespace {
struct SpearmanPlan {
  std::vector<SpearmanTask> tasks;
};
struct SpearmanTaskState {
  DevBuf kx, ky, count;
  uint64_t capacity = 0;     // pairs the buffers can hold
  uint64_t rows_upper = 0;   // host-side bound on pairs appended so far
  int64_t total_rows = 0;
};
struct SpearmanState {
  std::vector<SpearmanTaskState> tasks;
  // work buffers of the ranking (sorted keys, permutation, run heads, ranks, sort scratch): 40 bytes per pair, shared by
  // the tasks (they are ranked one after the other) and kept between calls -- allocating and freeing them inside
  // every fill_result was 160 ms of a 188 ms step at 100 M rows (hipMalloc / hipFree of 4.4 GB)
  DevBuf keys_sorted, idx, idx_sorted, heads, rx, ry, temp, partials;
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
    if (t.count.p) (void)hipMemsetAsync(t.count.p, 0, 8, st->stream);
  }
}

tgx_status spearman_update(tgx_state *st, const tgx_column *dev, tgx_error *err) {
  if (!st->plan->spearman) return TGX_OK;
  const SpearmanPlan *sp = splan(st->plan);
  SpearmanState *ss = sstate(st);
  for (size_t i = 0; i < sp->tasks.size(); i++) {
    const tgx_column &x = dev[sp->tasks[i].col_x], &y = dev[sp->tasks[i].col_y];
    SpearmanTaskState &ts = ss->tasks[i];
    auto numeric = [](int t) { return t == TGX_INT64 || t == TGX_FLOAT64; };
    if (!numeric(x.type) || !numeric(y.type))
      return sfail(err, TGX_INVALID_ARGUMENT, "SPEARMAN needs numeric columns (%d, %d)", x.type, y.type);
    ts.total_rows += x.length;
    if (x.length == 0) continue;
    if (!ts.count.p) {
      SHIP(ts.count.reserve(16));
      SHIP(hipMemsetAsync(ts.count.p, 0, 16, st->stream));
    }
    const uint64_t need = ts.rows_upper + (uint64_t)x.length;
    if (need > ts.capacity) {
      const uint64_t cap = std::max<uint64_t>(need, ts.capacity * 2);
      DevBuf nx, ny;
      SHIP(nx.reserve(cap * 8));
      SHIP(ny.reserve(cap * 8));
      if (ts.capacity) {
        SHIP(hipMemcpyAsync(nx.p, ts.kx.p, ts.rows_upper * 8, hipMemcpyDeviceToDevice, st->stream));
        SHIP(hipMemcpyAsync(ny.p, ts.ky.p, ts.rows_upper * 8, hipMemcpyDeviceToDevice, st->stream));
        SHIP(hipStreamSynchronize(st->stream));
      }
      ts.kx = std::move(nx);
      ts.ky = std::move(ny);
      ts.capacity = cap;
    }
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
    launch_spearman_compact(d, ts.kx.as<uint64_t>(), ts.ky.as<uint64_t>(), ts.count.as<unsigned long long>(), st->stream);
    ts.rows_upper = need;
  }
  return TGX_OK;
}

tgx_status spearman_fill_result(tgx_state *st, int slot, tgx_result *r, tgx_error *err) {
  SpearmanTaskState &ts = sstate(st)->tasks[slot];
  const bool exact = splan(st->plan)->tasks[slot].exact_sums;
  r->total = ts.total_rows;
  unsigned long long m = 0;
  if (ts.count.p) {
    SHIP(hipMemcpyAsync(&m, ts.count.p, 8, hipMemcpyDeviceToHost, st->stream));
    SHIP(hipStreamSynchronize(st->stream));
  }
  r->non_null = (int64_t)m;
  if (m == 0) return TGX_OK;
  if (m > 0xFFFFFFF0ull) return sfail(err, TGX_UNSUPPORTED, "SPEARMAN over more than 2^32 rows is not supported");
  SpearmanState *ws = sstate(st);
  DevBuf &keys_sorted = ws->keys_sorted, &idx = ws->idx, &idx_sorted = ws->idx_sorted, &heads = ws->heads,
         &rx = ws->rx, &ry = ws->ry, &temp = ws->temp, &partials = ws->partials;
  SHIP(keys_sorted.reserve(m * 8));
  SHIP(idx.reserve(m * 4));
  SHIP(idx_sorted.reserve(m * 4));
  SHIP(heads.reserve(m * 8));
  SHIP(rx.reserve(m * 8));
  SHIP(ry.reserve(m * 8));
  size_t temp_bytes = 0;
  SHIP(spearman_rank(ts.kx.as<uint64_t>(), m, keys_sorted.as<uint64_t>(), idx.as<uint32_t>(), idx_sorted.as<uint32_t>(),
                     heads.as<uint64_t>(), rx.as<uint64_t>(), nullptr, &temp_bytes, st->stream));
  SHIP(temp.reserve(temp_bytes + 256));
  SHIP(spearman_rank(ts.kx.as<uint64_t>(), m, keys_sorted.as<uint64_t>(), idx.as<uint32_t>(), idx_sorted.as<uint32_t>(),
                     heads.as<uint64_t>(), rx.as<uint64_t>(), temp.p, &temp_bytes, st->stream));
  SHIP(spearman_rank(ts.ky.as<uint64_t>(), m, keys_sorted.as<uint64_t>(), idx.as<uint32_t>(), idx_sorted.as<uint32_t>(),
                     heads.as<uint64_t>(), ry.as<uint64_t>(), temp.p, &temp_bytes, st->stream));
  SHIP(partials.reserve(2048 * spearman_rank_sums_bytes()));
  const int blocks = launch_rank_sums(rx.as<uint64_t>(), ry.as<uint64_t>(), m, partials.p, st->stream);
  std::vector<RankSumsHost> h(blocks);
  SHIP(hipMemcpyAsync(h.data(), partials.p, blocks * sizeof(RankSumsHost), hipMemcpyDeviceToHost, st->stream));
  SHIP(hipStreamSynchronize(st->stream));
  double out[5];
  for (int k = 0; k < 5; k++) {
    unsigned long long w = 0;
    unsigned __int128 e = 0;
    for (auto &p : h) {
      w += p.wrapped[k];
      e += ((unsigned __int128)p.exact_hi[k] << 64) | p.exact_lo[k];
    }
    out[k] = exact ? (double)e : (double)w;
  }
  r->sum_x = out[0];
  r->sum_y = out[1];
  r->sum_x2 = out[2];
  r->sum_y2 = out[3];
  r->sum_xy = out[4];
  return TGX_OK;
}

tgx_status spearman_check_mergeable(tgx_state *st, tgx_error *err) {
  if (!st->spearman) return TGX_OK;
  for (auto &t : sstate(st)->tasks)
    if (t.rows_upper > 0 || t.total_rows > 0)
      return sfail(err, TGX_UNSUPPORTED,
                   "Spearman states hold ranks of one data set and cannot be merged or serialized "
                   "(as in the reference, analyzers/advanced/correlation.rs:103-109)");
  return TGX_OK;
}

}  // namespace tgx
