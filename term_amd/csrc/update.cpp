// update.cpp -- one batch through the fused pass: HOST columns staged, what rides on what decided, every kernel of
// the plan queued (tgx_update, update_impl).  Split off tgx_api.cpp in round 4; see api_internal.h.
#include "api_internal.h"

#include <chrono>
namespace {
// TGX_UPDATE_TIMING=1: where the HOST's time inside tgx_update goes (microseconds since the call began, at a few marks)
struct UpdateMarks {
  bool on;
  std::chrono::steady_clock::time_point t0;
  char line[512];
  int len = 0;
  UpdateMarks() {
    static const bool enabled = getenv("TGX_UPDATE_TIMING") != nullptr;
    on = enabled;
    if (on) t0 = std::chrono::steady_clock::now();
  }
  void mark(const char *what) {
    if (!on) return;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    len += snprintf(line + len, sizeof(line) - (size_t)len > 0 ? sizeof(line) - (size_t)len : 0, " %s %.1f", what, us);
    if (len > (int)sizeof(line) - 1) len = (int)sizeof(line) - 1;
  }
  ~UpdateMarks() {
    if (on && len) fprintf(stderr, "tgx_update (us):%s\n", line);
  }
};
thread_local UpdateMarks *tl_marks = nullptr;
inline void update_mark(const char *what) {
  if (tl_marks) tl_marks->mark(what);
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// update

// copies a HOST column's buffers to the device; `out` is the device view


// `widen32`: a TGX_INT32 / TGX_FLOAT32 column is needed as 8-byte values (DISTINCT, KLL, co-moments, Spearman); the
// scan alone reads 4-byte values as they are
tgx_status stage_column(tgx_state *st, const tgx_column &c, tgx_column *out, tgx_error *err,
                               bool widen32) {
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
        HIP_TRY(pinned_alloc(&st->arena_host[k], kArenaBytes));
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
  if (c.type == TGX_UINT64) {
    // 8-byte values read in place: a key is its bit pattern (COUNT / DISTINCT only: update_validate)
    tgx_column as_i64 = c;
    as_i64.type = TGX_INT64;
    TGX_TRY(stage_column(st, as_i64, out, err, false));
    return TGX_OK;
  }
  if (c.mem == TGX_MEM_DEVICE && !(is_widened(c.type) && widen32)) return TGX_OK;
  // Only the window the batch views is copied: a sliced array (offset > 0 into big buffers) costs its own rows,
  // not everything before them.  The window starts at slot e0 = offset rounded down to 64 (keeps the validity
  // byte / word alignment the kernels like); the device view gets offset - e0 as its Arrow offset.
  const int64_t e0 = c.offset & ~(int64_t)63;
  const int64_t slots = c.offset - e0 + c.length;  // slots of the window
  const void *p = nullptr;
  if (is_widened(c.type)) {
    // 4-byte numerics, narrow integers, Booleans (include/tgx.h): the window is widened to 8-byte values in a staging buffer on the device; the
    // kernels then see an Int64 / Float64 column.  DEVICE columns keep their validity bitmap and Arrow offset as they
    // are (only the values move: slot e0 of the source becomes slot 0 of the widened buffer, so the bitmap of a device
    // column is re-based by staging nothing and pointing at byte e0 / 8).
    const bool host = c.mem == TGX_MEM_HOST;
    // (e0 is a multiple of 64: a whole number of bytes of a Boolean column's bits too)
    const void *src = c.values ? (const uint8_t *)c.values + narrow_bytes(c.type, e0) : nullptr;
    if (host) {
      TGX_TRY(stage(c.validity ? c.validity + (e0 >> 3) : nullptr, c.validity ? (size_t)((slots + 7) / 8) : 0, &p));
      out->validity = (const uint8_t *)p;
      TGX_TRY(stage(src, narrow_bytes(c.type, slots), &p));
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
    out->type = widened_type(c.type);
    out->values = nullptr;
    if (src) {
      if (st->staging_used == st->staging.size()) st->staging.emplace_back(new DevBuf());
      DevBuf *w = st->staging[st->staging_used++].get();
      HIP_TRY(w->reserve((size_t)slots * 8 + 16));
      // launched by tgx_update once the pinned arena (small HOST buffers travel in it) has been uploaded
      st->pending_widen.push_back({src, w->p, slots, widen_mode(c.type)});
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

void fill_scan_desc(const tgx_column &c, bool variance, const double *pivot, ScanColDesc *d) {
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

int scan_blocks_for(const ScanColDesc &d, int n_cols_in_launch, int per_cu) {
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

// The pivots of the pairs of one launch (kernels/comoments.hip, como_pivot_kernel): picked from the first batches that
// bring rows -- the kernel leaves a pair alone once rows have been folded into it; after a few batches nothing is
// launched any more (a stream of 8192-row batches must not pay a launch per batch for a decision long taken).
tgx_status como_pivots(tgx_state *st, const ComomentLaunch &L, int n_pairs, tgx_error *err) {
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

tgx_status update_validate(const tgx_plan *plan, tgx_state *st, const tgx_column *columns, size_t n_columns,
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
    if (c.type < TGX_INT64 || c.type > TGX_BOOL) return fail(err, TGX_INVALID_ARGUMENT, "column %d: unknown type %d", i, c.type);
    if (is_keys_only(c.type) && plan->stats_on[i])
      return fail(err, TGX_UNSUPPORTED, "column %d: a %s column takes COUNT and DISTINCT checks only", i,
                  c.type == TGX_BOOL ? "Boolean" : "UInt64");
    if (c.mem != TGX_MEM_HOST && c.mem != TGX_MEM_DEVICE)
      return fail(err, TGX_INVALID_ARGUMENT, "column %d: unknown memory space %d", i, c.mem);
    if (st->col_types[i] == 0) st->col_types[i] = c.type;
    if (st->col_types[i] != c.type)
      return fail(err, TGX_INVALID_ARGUMENT, "column %d changed type between batches (%d -> %d)", i,
                  st->col_types[i], c.type);
    const bool host = c.mem == TGX_MEM_HOST;
    traits->any_host |= host;
    traits->any_utf8 |= c.type == TGX_UTF8;
    traits->any_strings |= is_any_string(c.type) || c.type == TGX_DICT32_UTF8;
    // string windows need their first / last offsets (Utf8View: the stretches its views point into; dictionaries:
    // theirs) on the host: HOST batches only (what DataFusion streams); DEVICE strings keep the immediate path
    traits->coalescible &= is_numeric(c.type) || is_numeric32(c.type) || c.type == TGX_UINT64 ||
                           (is_narrow_int(c.type) && host) ||  // (widened on the way into the arena; DEVICE: immediate)
                           (is_string(c.type) && host) ||
                           (c.type == TGX_UTF8_VIEW && host) ||
                           (c.type == TGX_DICT32_UTF8 && host && c.dictionary && c.dictionary->mem == TGX_MEM_HOST);
    if (c.length > 0) {
      if ((is_numeric(c.type) || is_widened(c.type) || c.type == TGX_UINT64) && reads_values[i] && !c.values)
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


extern "C" tgx_status tgx_update(const tgx_plan *plan, tgx_state *st, const tgx_column *columns,
                                 size_t n_columns, tgx_error *err) try {
  if (!plan || !st || st->plan != plan) return fail(err, TGX_INVALID_ARGUMENT, "state does not belong to plan");
  if ((int)n_columns < plan->n_columns_needed)
    return fail(err, TGX_INVALID_ARGUMENT, "plan reads column %d but only %zu columns were passed",
                plan->n_columns_needed - 1, n_columns);
  if (n_columns > 0 && !columns) return fail(err, TGX_INVALID_ARGUMENT, "columns is NULL");
  UpdateMarks marks;
  struct MarksScope {
    MarksScope(UpdateMarks *m) { tl_marks = m->on ? m : nullptr; }
    ~MarksScope() { tl_marks = nullptr; }
  } marks_scope(&marks);
  TGX_TRY(need_device(err));
  // TGX_MEM_HOST_RETAINED (include/tgx.h): HOST columns whose copy may wait for the flush.  Below this point such a
  // column is a HOST column; `retained` (all of the batch's HOST columns were given so) travels in the traits.
  bool any_retained = false, any_plain_host = false;
  for (int i = 0; i < plan->n_columns_needed; i++) {
    if (!plan->used[i]) continue;
    const tgx_column &c = columns[i];
    any_retained |= c.mem == TGX_MEM_HOST_RETAINED || (c.dictionary && c.dictionary->mem == TGX_MEM_HOST_RETAINED);
    any_plain_host |= c.mem == TGX_MEM_HOST || (c.dictionary && c.dictionary->mem == TGX_MEM_HOST);
  }
  static thread_local std::vector<tgx_column> as_host, as_host_dicts;
  if (any_retained) {
    as_host.assign(columns, columns + n_columns);
    as_host_dicts.clear();
    as_host_dicts.reserve(n_columns);  // (no reallocation: the columns point into it)
    for (size_t i = 0; i < n_columns; i++) {
      if ((int)i >= plan->n_columns_needed || !plan->used[i]) continue;
      tgx_column &c = as_host[i];
      if (c.mem == TGX_MEM_HOST_RETAINED) c.mem = TGX_MEM_HOST;
      if (c.dictionary && c.dictionary->mem == TGX_MEM_HOST_RETAINED) {
        as_host_dicts.push_back(*c.dictionary);
        as_host_dicts.back().mem = TGX_MEM_HOST;
        c.dictionary = &as_host_dicts.back();
      }
    }
    columns = as_host.data();
  }
  int64_t nrows = 0;
  BatchTraits traits;
  traits.retained = any_retained && !any_plain_host;
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
  update_mark("validated");
  bind_thread();  // (the noted-only path above makes no HIP call: it binds where it does, in the arena set-up and the flush)
  TGX_TRY(coalesce_flush(st, err));  // batches stay in order
  update_mark("flushed");
  const tgx_status rc = update_impl(plan, st, columns, nrows, err);
  update_mark("done");
  return rc;
} catch (...) {
  return tgx::abi_exception(err);
}

// one batch through the fused pass: device views of its columns, then every kernel of the plan
tgx_status update_impl(const tgx_plan *plan, tgx_state *st, const tgx_column *columns, int64_t nrows,
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
      if ((is_numeric32(dev[i].type) && needs_wide[i]) || is_narrow_int(dev[i].type)) dev[i].type = widened_type(dev[i].type);
      if (dev[i].type == TGX_UINT64) dev[i].type = TGX_INT64;
      continue;
    }
    // (narrow integers and Booleans are always widened: no kernel reads them in place -- unless no check reads the
    //  column's values at all: a completeness check needs the validity bitmap only)
    const bool widen = needs_wide[i] != 0 || (is_narrow_int(columns[i].type) && plan->reads_values[i]);
    if (is_narrow_int(columns[i].type) && !widen) {
      tgx_column bare = columns[i];
      bare.type = TGX_INT64;
      bare.values = nullptr;
      TGX_TRY(stage_column(st, bare, &dev[i], err, false));
      continue;
    }
    TGX_TRY(stage_column(st, columns[i], &dev[i], err, widen));
  }
  update_mark("staged");
  const bool arena_in_use = st->arena_used != 0;
  if (arena_in_use)
    HIP_TRY(hipMemcpyAsync(st->arena_dev[st->arena_cur].p, st->arena_host[st->arena_cur], st->arena_used,
                           hipMemcpyHostToDevice, st->stream));
  for (const auto &w : st->pending_widen) launch_widen32(w.src, w.dst, w.n, w.mode, g_ctx.n_cu, st->stream);
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
    update_mark("fused");
    TGX_TRY(distinct_sample_all(st, dev.data(), err));
    update_mark("sampled");
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
    update_mark("prepared");
    // BESIDE the scan (round 6, an experiment that stays OFF: TGX_KEYS_BESIDE_SCAN=1 turns it on): the key passes
    // queued on a stream of their own in front of the scan, which then runs next to them instead of behind them.  The
    // key stream starts behind everything the state's stream held when the update began (`batch_in`), and the state's
    // stream waits for `keys_ready` once the scan is queued, so whatever follows the update sees both.  Measured on
    // MI355X: 1 G rows x 16 columns 24.9 ms against 23.7 one after the other (both kernels stretch: the partition pass
    // -- one 1024-thread workgroup a CU with 152 KiB of LDS -- and the scan's waves contend for the same CUs and for
    // HBM the scan alone already saturates); one rank's shard of 125 M rows 3.44 against 3.43 ms.
    bool any_key_pass = false;
    for (size_t q = 0; q < plan->distinct.size(); q++) {
      const DistinctTask &t = plan->distinct[q];
      any_key_pass |= t.tuple.empty() && !distinct_idle(q) && is_numeric(dev[t.column].type) && dev[t.column].length > 0 &&
                      dprep[q].partitioned;
    }
    static const bool beside_on = [] {
      const char *e = getenv("TGX_KEYS_BESIDE_SCAN");
      return e && e[0] == '1';
    }();
    const bool beside = beside_on && any_key_pass && nrows >= (1 << 22) && !plan->scan.empty();
    struct KeyStreamLoan {
      tgx_state *st;
      hipStream_t own;
      bool on = false;
      ~KeyStreamLoan() {
        if (on) st->stream = own;
      }
    } key_loan{st, st->stream};
    if (beside) {
      if (!st->key_stream) {
        HIP_TRY(stream_acquire(&st->key_stream, false));
        HIP_TRY(hipEventCreateWithFlags(&st->batch_in, hipEventDisableTiming));
      }
      HIP_TRY(hipEventRecord(st->batch_in, st->stream));
      HIP_TRY(hipStreamWaitEvent(st->key_stream, st->batch_in, 0));
      st->stream = st->key_stream;
      key_loan.on = true;
    }
    for (size_t q = 0; q < plan->distinct.size(); q++) {
      const DistinctTask &t = plan->distinct[q];
      if (!t.tuple.empty() || distinct_idle(q) || !is_numeric(dev[t.column].type) || dev[t.column].length == 0) continue;
      const int stats_slot = (t.scan_slot >= 0 && stats_by_partition[t.scan_slot] == (int)q) ? t.scan_slot : -1;
      TGX_TRY(distinct_update(st, q, dev[t.column], err, nullptr, &dprep[q], stats_slot, &columns[t.column]));
      distinct_done[q] = 1;
    }
    update_mark("keys_queued");
    if (!st->keys_ready) HIP_TRY(hipEventCreateWithFlags(&st->keys_ready, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(st->keys_ready, st->stream));
    update_mark("event");
    if (beside) {
      st->stream = key_loan.own;
      key_loan.on = false;
    }
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
    const int scan_per_cu = (st->exchange_expected || beside) ? 6 : 8;
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
    // (the key passes that ran beside the scan: whatever follows on the state's stream sees them done)
    if (beside) HIP_TRY(hipStreamWaitEvent(st->stream, st->keys_ready, 0));
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
    TGX_TRY(spearman_update(st, dev.data(), columns, !st->coalesce.flushing && !g_ctx.no_coalesce, err));
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

