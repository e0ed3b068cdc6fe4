// distinct_state.cpp -- the host side of the exact key sets: bitmaps / hash sets / partitioned lists per DISTINCT task,
// the sampled range and its repair, export / import / adopt, tgx_merge.  Split off tgx_api.cpp in round 4.
#include "api_internal.h"

uint64_t next_pow2(uint64_t x) {
  uint64_t p = 1;
  while (p < x) p <<= 1;
  return p;
}

tgx_status distinct_read_counters(tgx_state *st, DistinctState &ds, unsigned long long *out,
                                         tgx_error *err) {
  memset(out, 0, kNumDistinctCounters * sizeof(unsigned long long));
  if (!ds.counters.p) return TGX_OK;
  HIP_TRY(hipMemcpyAsync(out, ds.counters.p, kNumDistinctCounters * sizeof(unsigned long long),
                         hipMemcpyDeviceToHost, st->stream));
  HIP_TRY(hipStreamSynchronize(st->stream));
  return TGX_OK;
}

HashSetView hash_view(const DistinctState &ds) {
  HashSetView v;
  v.keys = ds.keys.as<uint64_t>();
  v.dup = ds.dup.as<uint32_t>();
  v.mask = ds.capacity - 1;
  const bool exact = ds.exact && ds.wide;
  v.store = exact ? ds.key_store.as<uint64_t>() : nullptr;
  v.store_cursor = exact ? ds.key_cursor.as<unsigned long long>() : nullptr;
  v.store_words = exact ? ds.key_store_words : 0;
  v.pending = exact ? ds.key_pending.as<uint64_t>() : nullptr;
  v.pending_counts = exact ? ds.key_pending_counts.as<uint32_t>() : nullptr;
  v.pending_region = exact ? ds.key_pending_region : 0;
  v.pending_waves = exact ? ds.key_pending_waves : 0;
  v.pad_ = 0;
  return v;
}

// ---- the key store of an exact set ----
// A fresh table starts a fresh store: words 0..1 are a stand-in entry (what a key that found no room points at:
// kCntStoreFull), the cursor starts behind it.
constexpr uint64_t kKeyStoreMinWords = 1ull << 16;
constexpr uint64_t kKeyStoreSlackWords = 1ull << 26;  // (512 MiB: what the room for later batches may add at most)
constexpr uint64_t kHashSlackRows = 1ull << 23;       // (the same for a table's slots: 8 Mi keys' worth)
tgx_status key_store_init(tgx_state *st, DistinctState &ds, tgx_error *err) {
  if (ds.key_store_words < kKeyStoreMinWords) {
    HIP_TRY(ds.key_store.reserve(kKeyStoreMinWords * 8));
    ds.key_store_words = ds.key_store.cap / 8;
  }
  HIP_TRY(ds.key_cursor.reserve(4 * sizeof(unsigned long long)));
  const unsigned long long head[2] = {0ull, 1ull << 32 /* kKindFpOnly */}, cur[4] = {2, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(ds.key_store.p, head, sizeof(head), hipMemcpyHostToDevice, st->stream));
  HIP_TRY(hipMemcpyAsync(ds.key_cursor.p, cur, sizeof(cur), hipMemcpyHostToDevice, st->stream));
  HIP_TRY(hipStreamSynchronize(st->stream));  // (`head` / `cur` are on this stack frame)
  ds.key_words_ub = 2;
  return TGX_OK;
}
// Room for `incoming` more words behind the cursor (`cursor`: its value, just read back).  A bigger block takes the
// old one's words over; references are word offsets and stay as they are.
tgx_status key_store_ensure(tgx_state *st, DistinctState &ds, uint64_t cursor, uint64_t incoming, tgx_error *err) {
  if (cursor + incoming <= ds.key_store_words) return TGX_OK;
  const uint64_t want = std::max<uint64_t>(2 * ds.key_store_words, cursor + incoming);
  DevBuf bigger;
  HIP_TRY(bigger.reserve(want * 8));
  HIP_TRY(hipMemcpyAsync(bigger.p, ds.key_store.p, cursor * 8, hipMemcpyDeviceToDevice, st->stream));
  HIP_TRY(hipStreamSynchronize(st->stream));
  std::swap(ds.key_store.p, bigger.p);
  std::swap(ds.key_store.cap, bigger.cap);
  ds.key_store_words = ds.key_store.cap / 8;
  return TGX_OK;
}
// after a measuring pass has added the batch's worst case to key_cursor[1]: read it (and the cursor) and make room
tgx_status key_store_reserve_measured(tgx_state *st, DistinctState &ds, tgx_error *err) {
  unsigned long long h[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(h, ds.key_cursor.p, sizeof(h), hipMemcpyDeviceToHost, st->stream));
  HIP_TRY(hipStreamSynchronize(st->stream));
  ds.key_words_ub = h[0] + h[1];
  return key_store_ensure(st, ds, h[0], h[1], err);
}
// ... or against the host's bound on the fill, when the host can bound the batch itself (`worst` words): no wait unless
// the bound has run out of room -- then the real fill is read, and the store grows only if THAT needs it
tgx_status key_store_reserve_bound(tgx_state *st, DistinctState &ds, uint64_t worst, tgx_error *err) {
  if (ds.key_words_ub + worst <= ds.key_store_words) {
    ds.key_words_ub += worst;
    return TGX_OK;
  }
  unsigned long long cur = 0;
  HIP_TRY(hipMemcpyAsync(&cur, ds.key_cursor.p, sizeof(cur), hipMemcpyDeviceToHost, st->stream));
  HIP_TRY(hipStreamSynchronize(st->stream));
  ds.key_words_ub = cur + worst;
  // Room for a few batches of this size, not for this one alone: the bound grows by `worst` per batch whatever the
  // batch adds (a column of repeated values adds next to nothing), and every time it runs out the host WAITS for the
  // device here -- a stream of flushes of a HOST column then copies and computes in turns instead of side by side.
  return key_store_ensure(st, ds, cur, worst + std::min<uint64_t>(3 * worst, kKeyStoreSlackWords), err);
}
// before a batch's insert: the measuring pass's scratch and the pending list's fill are zero, and the list has room for
// `items` new keys (every item of the batch at worst)
tgx_status key_store_measure_begin(tgx_state *st, DistinctState &ds, uint64_t items, tgx_error *err) {
  // the pending list of a batch of `items` items: a region per wave of the insert kernel's launch (exact_blocks)
  ds.key_pending_waves = exact_waves(items);
  ds.key_pending_region = exact_region(items);
  HIP_TRY(ds.key_pending.reserve((size_t)ds.key_pending_waves * ds.key_pending_region * 16));
  HIP_TRY(ds.key_pending_counts.reserve((size_t)ds.key_pending_waves * sizeof(uint32_t)));
  HIP_TRY(hipMemsetAsync(ds.key_pending_counts.p, 0, (size_t)ds.key_pending_waves * sizeof(uint32_t), st->stream));
  HIP_TRY(hipMemsetAsync(ds.key_cursor.as<unsigned long long>() + 1, 0, 3 * sizeof(unsigned long long), st->stream));
  return TGX_OK;
}
// room for the keys of a string column batch (every valid row a new key at worst)
tgx_status key_store_reserve_utf8(tgx_state *st, DistinctState &ds, const tgx_column &c, tgx_error *err) {
  TGX_TRY(key_store_measure_begin(st, ds, (uint64_t)c.length, err));
  if (ds.batch_bytes_known && c.type != TGX_UTF8_VIEW)  // (an entry: two words + ceil(len / 8) <= len / 8 + 3 words)
    return key_store_reserve_bound(st, ds, 3 * (uint64_t)c.length + (uint64_t)ds.batch_data_bytes / 8, err);
  const bool view = c.type == TGX_UTF8_VIEW;
  launch_exact_measure_utf8(c.offsets, c.data, view ? c.values : nullptr, view ? c.variadic : nullptr, c.validity, c.offset,
                            c.length, c.type == TGX_LARGE_UTF8, nullptr, ds.key_cursor.as<unsigned long long>() + 1,
                            st->stream);
  return key_store_reserve_measured(st, ds, err);
}
BitmapView bitmap_view(const DistinctState &ds) {
  BitmapView v;
  v.seen = ds.seen.as<uint32_t>();
  v.twice = ds.twice.as<uint32_t>();
  v.base = ds.base;
  v.range = ds.range;
  return v;
}

// allocate an empty table of `capacity` slots into (keys, dup)
tgx_status hash_alloc(tgx_state *st, DevBuf &keys, DevBuf &dup, uint64_t capacity, bool mult,
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
tgx_status hash_ensure(tgx_state *st, DistinctState &ds, bool mult, uint64_t incoming,
                              tgx_error *err) {
  // (room for a few more batches of this size: see key_store_reserve_bound -- the bound below counts every row of every
  //  batch as a new key, and where it runs out the host waits for the device to read the real count)
  const uint64_t slack = std::min<uint64_t>(3 * incoming, kHashSlackRows);
  if (ds.capacity == 0) {
    uint64_t want = std::max<uint64_t>(incoming + slack, g_ctx.distinct_hint);
    ds.capacity = next_pow2(std::max<uint64_t>(2 * want, 1024));
    TGX_TRY(hash_alloc(st, ds.keys, ds.dup, ds.capacity, mult, ds.wide, err));
    ds.rows_upper_bound = 0;
    if (ds.exact && ds.wide) TGX_TRY(key_store_init(st, ds, err));
  }
  if (2 * (ds.rows_upper_bound + incoming) <= ds.capacity) {
    ds.rows_upper_bound += incoming;
    return TGX_OK;
  }
  // the bound says it might not fit: read the real key count
  unsigned long long c[kNumDistinctCounters];
  TGX_TRY(distinct_read_counters(st, ds, c, err));
  uint64_t actual = c[kCntDistinct];
  if (2 * (actual + incoming + slack) <= ds.capacity) {
    ds.rows_upper_bound = actual + incoming;
    return TGX_OK;
  }
  uint64_t new_cap = next_pow2(2 * (actual + incoming + slack));
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

tgx_status bitmap_to_hash(tgx_state *st, DistinctState &ds, bool mult, uint64_t incoming,
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
tgx_status tuple_desc_of(const std::vector<const tgx_column *> &cols, bool mult, TupleDesc *d, tgx_error *err) {
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
    } else if (c.type == TGX_DICT32_UTF8 && c.dictionary &&
               (c.dictionary->type == TGX_UTF8 || c.dictionary->type == TGX_LARGE_UTF8)) {
      // the component is the row's dictionary ENTRY (its string, or NULL): the same tuple as over the decoded column
      const tgx_column &dc = *c.dictionary;
      tc.kind = 4;
      tc.dict_large = dc.type == TGX_LARGE_UTF8 ? 1 : 0;
      tc.values = c.values;
      tc.offsets = dc.offsets;
      tc.data = dc.data;
      tc.dict_validity = dc.validity;
      tc.dict_offset = dc.offset;
    } else {
      return fail(err, TGX_UNSUPPORTED, "DISTINCT over a tuple: column type %d is not supported", c.type);
    }
  }
  return TGX_OK;
}


// COUNT(DISTINCT (a, b, ...)): every row's tuple goes into the 128-bit fingerprint set (kernels/distinct128.hip)
tgx_status distinct_tuple_update(tgx_state *st, size_t slot, const tgx_column *dev, tgx_error *err,
                                        const tgx_column *orig) {
  const DistinctTask &task = st->plan->distinct[slot];
  DistinctState &ds = st->distinct[slot];
  std::vector<const tgx_column *> cols;
  bool any_view = false;  // a component whose device view lives in per-update scratch: never retained (no lists)
  for (int c2 : task.tuple) {
    cols.push_back(&dev[c2]);
    any_view |= dev[c2].type == TGX_UTF8_VIEW || (orig && is_widened(orig[c2].type));
  }
  TupleDesc d;
  TGX_TRY(tuple_desc_of(cols, task.multiplicity, &d, err));
  d.key = st->plan->fp_key;
  ds.col_type = TGX_UTF8;  // a 128-bit fingerprint set, like a string column's
  ds.total_rows += d.length;
  if (d.length == 0) return TGX_OK;
  if (ds.fp_staged) TGX_TRY(distinct_resolve(st, slot, err));  // a second batch: the table takes over
  // the first big batch: through the partitioned lists (views read their buffers through a table staged per update)
  // (an exact set takes the lists only over the caller's own DEVICE buffers, which outlive the update: its records refer
  //  to rows, and what turns the lists into a table later needs the rows' bytes)
  bool exact_lists_ok = orig != nullptr && d.length < (int64_t)1 << 32;
  if (orig)
    for (int c2 : task.tuple) exact_lists_ok &= orig[c2].mem == TGX_MEM_DEVICE && !is_widened(orig[c2].type);
  if (ds.mode == DistinctMode::kUndecided && !any_view && (!ds.exact || exact_lists_ok) && fp_lists_fit_rows(d.length))
    return fp_lists_tuple_update(st, slot, d, cols, err);
  if (ds.mode == DistinctMode::kUndecided) {
    ds.mode = DistinctMode::kHash;
    ds.wide = true;
  }
  TGX_TRY(hash_ensure(st, ds, task.multiplicity, (uint64_t)d.length, err));
  if (ds.exact) {
    TGX_TRY(key_store_measure_begin(st, ds, (uint64_t)d.length, err));
    launch_exact_measure_tuple(d, ds.key_cursor.as<unsigned long long>() + 1, st->stream);
    TGX_TRY(key_store_reserve_measured(st, ds, err));
  }
  ProfScope ps(st, "distinct", 0);
  launch_distinct_tuple(d, hash_view(ds), ds.counters.as<unsigned long long>(), st->stream);
  return TGX_OK;
}


// ---- big Utf8 batches: partitioned fingerprint lists (kernels/distinct128.hip, fp_*) ----
// records a list is sized for when `rows` values are spread over `lists` lists: the mean, twelve standard deviations
// (values that repeat widen the spread) and a floor
uint64_t fp_list_cap(int64_t rows, uint64_t lists) {
  const double mean = (double)rows / (double)lists;
  return ((uint64_t)(mean + 12.0 * std::sqrt(mean) + 64.0) + 15) & ~15ull;
}
bool fp_lists_fit_rows(int64_t rows) {
  // TGX_FP_LISTS_MIN_ROWS: smallest batch that takes this path (tests lower it; a huge value turns the path off)
  int64_t min_rows = kFpMinRows;
  if (const char *e = getenv("TGX_FP_LISTS_MIN_ROWS")) min_rows = std::max<int64_t>(1, atoll(e));
  return rows >= min_rows && fp_list_cap(rows, (uint64_t)kFpFan * kFpFan) <= kFpListMax;
}
bool fp_lists_fit(const tgx_column &c) {
  return (is_any_string(c.type) || c.type == TGX_INT64 || c.type == TGX_FLOAT64) && fp_lists_fit_rows(c.length);
}
void fp_views(const DistinctState &ds, FpLists *l1, FpLists *l2) {
  l1->recs = ds.fp_level1.as<uint64_t>();
  l1->offered = ds.fp_offered.as<uint32_t>();
  l1->cap = ds.fp_cap1;
  l2->recs = ds.fp_level2.as<uint64_t>();
  l2->offered = ds.fp_offered.as<uint32_t>() + kFpXcds * kFpFan;
  l2->cap = ds.fp_cap2;
}
// sizes and clears the two levels of lists for a batch of `rows` records of `rec_bytes` bytes
tgx_status fp_lists_prepare(tgx_state *st, DistinctState &ds, int64_t rows, size_t rec_bytes, tgx_error *err) {
  constexpr uint64_t kLists2 = (uint64_t)kFpFan * kFpFan;
  constexpr uint64_t kLists1 = (uint64_t)kFpXcds * kFpFan;
  ds.fp_cap1 = fp_list_cap(rows, kLists1);
  ds.fp_cap2 = fp_list_cap(rows, kLists2);
  // TGX_FP_LIST_CAPS="cap1:cap2" (tests): room for longer lists than the batch's size asks for, so that a batch of a
  // few hundred thousand keys chosen to fall into a dozen lists runs the kernels that otherwise need 800 M rows
  if (const char *e = getenv("TGX_FP_LIST_CAPS")) {
    unsigned long long c1 = 0, c2 = 0;
    if (sscanf(e, "%llu:%llu", &c1, &c2) == 2 && c2 <= kFpListMax) {
      ds.fp_cap1 = std::max<uint64_t>(ds.fp_cap1, (c1 + 15) & ~15ull);
      ds.fp_cap2 = std::max<uint64_t>(ds.fp_cap2, (c2 + 15) & ~15ull);
    }
  }
  HIP_TRY(ds.fp_level1.reserve(kLists1 * ds.fp_cap1 * rec_bytes));
  HIP_TRY(ds.fp_level2.reserve(kLists2 * ds.fp_cap2 * rec_bytes));
  HIP_TRY(ds.fp_offered.reserve((kLists1 + kLists2) * sizeof(uint32_t)));
  HIP_TRY(ds.fp_per_list.reserve(kLists2 * sizeof(uint2)));
  HIP_TRY(hipMemsetAsync(ds.fp_offered.p, 0, (kLists1 + kLists2) * sizeof(uint32_t), st->stream));
  return TGX_OK;
}

tgx_status fp_lists_update(tgx_state *st, size_t slot, const tgx_column &c, tgx_error *err) {
  DistinctState &ds = st->distinct[slot];
  const bool mult = st->plan->distinct[slot].multiplicity;
  TGX_TRY(fp_lists_prepare(st, ds, c.length, 16, err));
  FpLists l1, l2;
  fp_views(ds, &l1, &l2);
  ProfScope ps(st, "distinct", 0), ps_lists(st, "distinct_lists", 0);
  unsigned long long *counters = ds.counters.as<unsigned long long>();
  // an exact set: records carry their row, the count settles equal fingerprints on the rows' bytes
  uint32_t *fb_lo = nullptr;
  if (ds.exact) {
    HIP_TRY(ds.fp_fb_lo.reserve((size_t)c.length * sizeof(uint32_t) + 16));
    fb_lo = ds.fp_fb_lo.as<uint32_t>();
  }
  tgx_column kept = c;
  const bool view = c.type == TGX_UTF8_VIEW;
  if (view) {
    // the table of data-buffer pointers the kernels read through is staged per update: the retained view gets a copy
    const size_t bytes = (size_t)std::max(c.n_variadic, 1) * sizeof(void *);
    HIP_TRY(ds.fp_buffers.reserve(bytes));
    if (c.n_variadic > 0)
      HIP_TRY(hipMemcpyAsync(ds.fp_buffers.p, c.variadic, (size_t)c.n_variadic * sizeof(void *), hipMemcpyDeviceToDevice,
                             st->stream));
    kept.variadic = (const uint8_t *const *)ds.fp_buffers.p;
    launch_fp_partition_views(c.values, kept.variadic, c.validity, c.offset, c.length, l1, st->plan->fp_key, fb_lo, counters,
                              st->stream);
  } else {
    launch_fp_partition_strings(c.offsets, c.data, c.validity, c.offset, c.length, c.type == TGX_LARGE_UTF8, l1,
                                st->plan->fp_key, fb_lo, counters, st->stream);
  }
  launch_fp_partition_lists(l1, l2, counters, st->stream);
  if (ds.exact)
    launch_fp_count_exact_utf8(l2, mult ? 1 : 0, ds.fp_per_list.as<uint2>(), l1.offered, c.offsets, c.data,
                               view ? c.values : nullptr, view ? kept.variadic : nullptr, c.offset, c.length,
                               c.type == TGX_LARGE_UTF8, counters, st->stream);
  else
    launch_fp_count(l2, mult ? 1 : 0, ds.fp_per_list.as<uint2>(), l1.offered, counters, st->stream);
  ds.mode = DistinctMode::kHash;
  ds.wide = true;
  ds.capacity = 0;  // no table yet
  ds.rows_upper_bound = 0;
  ds.fp_staged = true;
  ds.fp_exact_lists = ds.exact;
  ds.retained.push_back(kept);  // (a DEVICE view, or a staged one looked at before the update returns)
  return TGX_OK;
}

// the same for the first big batch of a tuple task: its components are retained in tuple order
tgx_status fp_lists_tuple_update(tgx_state *st, size_t slot, const TupleDesc &d,
                                        const std::vector<const tgx_column *> &cols, tgx_error *err) {
  DistinctState &ds = st->distinct[slot];
  const bool mult = st->plan->distinct[slot].multiplicity;
  TGX_TRY(fp_lists_prepare(st, ds, d.length, 16, err));
  FpLists l1, l2;
  fp_views(ds, &l1, &l2);
  ProfScope ps(st, "distinct", 0), ps_lists(st, "distinct_lists", 0);
  unsigned long long *counters = ds.counters.as<unsigned long long>();
  uint32_t *fb_lo = nullptr;
  if (ds.exact) {
    HIP_TRY(ds.fp_fb_lo.reserve((size_t)d.length * sizeof(uint32_t) + 16));
    fb_lo = ds.fp_fb_lo.as<uint32_t>();
  }
  launch_fp_partition_tuples(d, l1, fb_lo, counters, st->stream);
  launch_fp_partition_lists(l1, l2, counters, st->stream);
  if (ds.exact)
    launch_fp_count_exact_tuple(l2, mult ? 1 : 0, ds.fp_per_list.as<uint2>(), d, counters, st->stream);
  else
    launch_fp_count(l2, mult ? 1 : 0, ds.fp_per_list.as<uint2>(), nullptr, counters, st->stream);  // (valid rows: level 1)
  ds.mode = DistinctMode::kHash;
  ds.wide = true;
  ds.capacity = 0;  // no table yet
  ds.rows_upper_bound = 0;
  ds.fp_staged = true;
  ds.fp_exact_lists = ds.exact;
  for (const tgx_column *c : cols) ds.retained.push_back(*c);
  return TGX_OK;
}

// `orig`: the caller's own view of the column when `c` is a per-update staging copy of a DEVICE column (a 4-byte
// numeric column widened for this pass): what a key set retains for a later repair must outlive the update
tgx_status distinct_update(tgx_state *st, size_t slot, const tgx_column &c, tgx_error *err,
                                  const std::vector<DictGather> *gathers, const NumericPrep *ready,
                                  int stats_slot, const tgx_column *orig) {
  const DistinctTask &task = st->plan->distinct[slot];
  DistinctState &ds = st->distinct[slot];
  const bool mult = task.multiplicity;
  if (is_any_string(c.type)) {
    // values are reduced to 128-bit fingerprints on the fly (kernels/distinct128.hip)
    ds.col_type = c.type;
    ds.total_rows += c.length;
    if (c.length == 0) return TGX_OK;
    if (ds.fp_staged) TGX_TRY(distinct_resolve(st, slot, err));  // a second batch: the table takes over
    // (an exact set takes the lists only over the caller's own DEVICE buffers, which outlive the update: its records
    //  refer to rows, and what turns the lists into a table later needs the rows' bytes)
    const bool exact_lists_ok = orig && orig->mem == TGX_MEM_DEVICE && c.length < (int64_t)1 << 32;
    if (ds.mode == DistinctMode::kUndecided && (!ds.exact || exact_lists_ok) && fp_lists_fit(c))
      return fp_lists_update(st, slot, c, err);
    if (ds.mode == DistinctMode::kUndecided) {
      ds.mode = DistinctMode::kHash;
      ds.wide = true;
    }
    TGX_TRY(hash_ensure(st, ds, mult, (uint64_t)c.length, err));
    if (ds.exact) TGX_TRY(key_store_reserve_utf8(st, ds, c, err));
    ProfScope ps(st, "distinct", 0);
    const bool view = c.type == TGX_UTF8_VIEW;
    launch_distinct_utf8(c.offsets, c.data, view ? c.values : nullptr, view ? c.variadic : nullptr, c.validity,
                         c.offset, c.length, c.type == TGX_LARGE_UTF8, mult ? 1 : 0, hash_view(ds), st->plan->fp_key,
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
    if (ds.exact) {  // room for the referenced entries
      TGX_TRY(key_store_measure_begin(st, ds, (uint64_t)dict.length, err));
      if (ds.batch_bytes_known) {
        TGX_TRY(key_store_reserve_bound(st, ds, 3 * (uint64_t)dict.length + (uint64_t)ds.batch_data_bytes / 8, err));
      } else {
      launch_exact_measure_utf8(dict.offsets, dict.data, nullptr, nullptr, dict.validity, dict.offset, dict.length,
                                dict.type == TGX_LARGE_UTF8, u_seen, ds.key_cursor.as<unsigned long long>() + 1, st->stream);
      TGX_TRY(key_store_reserve_measured(st, ds, err));
      }
    }
    launch_dict_insert(dict.offsets, dict.data, dict.validity, dict.offset, dict.length,
                       dict.type == TGX_LARGE_UTF8, mult ? 1 : 0, u_seen, u_twice, hash_view(ds), st->plan->fp_key,
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
void bitmap_shape(const DistinctState &ds, int64_t length, bool mult, uint32_t *sub_bits_out, bool *key16_out,
                         uint64_t *n_buckets_out, bool *partitioned_out) {
  // slices of 2^sub_bits keys; one slice (two with multiplicity) must fit 128 KiB of LDS.  How many buckets: every
  // (tile, bucket) run is padded to a 64-byte line, so FEWER buckets mean longer runs and less padding -- and fewer
  // workgroups for the replay, one per bucket.  With multiplicity (4-byte entries, padded with a filler): <= 1024
  // buckets, as ever.  Without (round 5): <= 256 -- a list entry is then 2 bytes where a bucket holds <= 2^16 keys, else
  // 20 bits (three to an 8-byte word: distinct_run_numeric, `pack20`).  Measured at 1 G keys over 10^8 values: 1526
  // buckets of 2-byte entries 3.26 ms, 763 / 382 / 191 / 96 buckets of 20-bit entries 3.09 / 3.07 / 2.93 / 3.36 ms.
  // (TGX_BUCKET_TARGET overrides the aim, TGX_KEY16=0 forbids 2-byte entries: experiments)
  const char *bt = getenv("TGX_BUCKET_TARGET");
  const uint64_t bucket_target = bt ? std::max<uint64_t>(1, strtoull(bt, nullptr, 10)) : (mult ? 1024 : 256);
  const char *k16 = getenv("TGX_KEY16");
  const bool allow_key16 = !(k16 && atoi(k16) == 0);
  uint32_t sub_bits = 14;
  while (sub_bits < (mult ? 19u : 20u) && ((ds.range + (1ull << sub_bits) - 1) >> sub_bits) > bucket_target) sub_bits++;
  const bool key16 = allow_key16 && !mult && sub_bits <= 16;
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

tgx_status pinned_readback(tgx_state *st, size_t bytes, tgx_error *err) {
  if (bytes <= st->h_pinned_cap) return TGX_OK;
  pinned_free(st->h_pinned, st->h_pinned_cap);
  st->h_pinned = nullptr;
  st->h_pinned_cap = 0;
  const size_t want = std::max<size_t>(bytes + bytes / 2, 4096);
  HIP_TRY(pinned_alloc(&st->h_pinned, want));
  st->h_pinned_cap = want;
  return TGX_OK;
}

// does this batch of an undecided Int64 key set get its range from a sample? (see distinct_prepare_numeric)
bool distinct_wants_sample(const DistinctState &ds, const tgx_column &c) {
  return ds.mode == DistinctMode::kUndecided && c.type == TGX_INT64 && !ds.has_hint && !ds.batch_range_known &&
         !ds.remembered && c.length >= (1 << 16);
}
// ... or the exact MIN / MAX of a coalesced flush whose key windows were DEVICE memory?  While the key set is undecided,
// or a bitmap no batch can have left outliers under: the flush then lays the bitmap out / grows it like a HOST flush
// (a stream of DEVICE batches of growing ids stays on the bitmap instead of going through the repair, flush after flush)
bool distinct_wants_exact_range(const DistinctState &ds, const tgx_column &c) {
  if (!ds.flush_device_keys || c.type != TGX_INT64 || ds.has_hint || ds.batch_range_known || c.length < (1 << 16))
    return false;
  return ds.mode == DistinctMode::kUndecided ||
         (ds.mode == DistinctMode::kBitmap && ds.speculative && !ds.partitioned && !ds.outliers_possible);
}

// The samples of ALL key columns of the batch, queued together and read back with ONE wait: a read-back costs the
// stream's latency (~50 us) whatever its size -- two key columns sampled one after the other were 6 % of a
// 100 M-row step.
tgx_status distinct_sample_all(tgx_state *st, const tgx_column *dev, tgx_error *err) {
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
tgx_status bitmap_grow(tgx_state *st, DistinctState &ds, bool mult, int64_t lo, int64_t hi, int64_t incoming,
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

tgx_status distinct_prepare_numeric(tgx_state *st, size_t slot, const tgx_column &c, NumericPrep *prep,
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
    } else if (c.type == TGX_INT64 && ds.remembered && c.length >= (1 << 16)) {
      have_range = true;  // what the column's sample said before the state was reset; outliers are repaired as ever
      lo = ds.remembered_lo;
      hi = ds.remembered_hi;
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
      if (ds.speculative && !ds.batch_range_known) {
        ds.remembered = true;
        ds.remembered_lo = lo;
        ds.remembered_hi = hi;
      }
    } else {
      ds.mode = DistinctMode::kHash;
      ds.remembered = false;
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
tgx_status distinct_run_numeric(tgx_state *st, size_t slot, const tgx_column &c, const NumericPrep &prep,
                                       int stats_slot, tgx_error *err, const tgx_column *orig) {
  const DistinctTask &task = st->plan->distinct[slot];
  // what a later repair walks again: never a view into the update's staging scratch (the next update reuses it) --
  // a widened DEVICE Int32 / Float32 column is retained as the caller's 4-byte column and widened again at the
  // repair (retained_numeric_view); staged copies of HOST batches are resolved before tgx_update returns
  const tgx_column &keep = (orig && orig->mem == TGX_MEM_DEVICE && is_widened(orig->type)) ? *orig : c;
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
      // list entries: 2 bytes (buckets of <= 2^16 keys, without multiplicity), else 20 bits, three to an 8-byte word, 24
      // to a 64-byte line (round 5: 2.67 instead of 4 bytes per key written and read back); 4 bytes only with TGX_PACK20=0
      const char *p20 = getenv("TGX_PACK20");  // (per batch: tests compare the two forms in one process)
      const bool no_pack20 = p20 && atoi(p20) == 0;
      // (with multiplicity the padding is a filler, 0xFFFFF: no sub-key of <= 19 bits)
      const bool pack20 = !prep.key16 && prep.sub_bits <= (mult ? 19u : 20u) && !no_pack20;
      uint64_t cap = (uint64_t)c.length / pp.n_lists;
      cap = cap + cap / 4 + (prep.key16 ? 32 : pack20 ? 24 : 16) * tiles + 4096;
      pp.cap = prep.key16 ? (cap + 31) & ~31ull : pack20 ? (cap + 23) / 24 * 24 : (cap + 15) & ~15ull;
      if (pp.cap >= (1ull << 32) - 64) return fail(err, TGX_INTERNAL, "distinct: list capacity out of range");
      pp.want_multiplicity = mult ? 1 : 0;
      pp.key16 = prep.key16 ? 1 : pack20 ? 2 : 0;
      static const bool no_probe = getenv("TGX_NO_CLUSTERED_PROBE") && atoi(getenv("TGX_NO_CLUSTERED_PROBE")) != 0;
      pp.probe = no_probe ? 0 : 1;
      // Only the form the column's last batch took is launched (PartitionParams::force_form; either form is correct on
      // any keys, the probe still runs and the next finalize remembers what it said): the form that would leave at once
      // still costs its dispatch, 1024 threads x a workgroup per CU -- C2 over 40 steps, three runs each: 1.594 against
      // 1.606 ms.  TGX_FORM_MEMORY=0 launches both again.
      static const bool form_memory = !(getenv("TGX_FORM_MEMORY") && atoi(getenv("TGX_FORM_MEMORY")) == 0);
      pp.force_form = (form_memory && !no_probe) ? ds.remembered_form : 0;
      HIP_TRY(ds.lists.reserve(prep.key16 ? (uint64_t)pp.n_lists * pp.cap * sizeof(uint16_t)
                               : pack20   ? (uint64_t)pp.n_lists * pp.cap / 3 * 8
                                          : (uint64_t)pp.n_lists * pp.cap * sizeof(uint32_t)));
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
        // (the outliers' share is folded by the same launch: scan_reduce_kernel)
        launch_scan_reduce_only(L, 1, grid, pp.stats, st->d_scan_acc.as<ScanAcc>(), st->stream, pp.outliers);
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
tgx_status retained_numeric_view(tgx_state *st, const tgx_column &col, std::vector<std::unique_ptr<DevBuf>> &tmp,
                                        tgx_column *out, tgx_error *err) {
  *out = col;
  if (!is_widened(col.type)) return TGX_OK;
  const int64_t e0 = col.offset & ~(int64_t)63;
  const int64_t slots = col.offset - e0 + col.length;
  tmp.emplace_back(new DevBuf());
  DevBuf *w = tmp.back().get();
  HIP_TRY(w->reserve((size_t)slots * 8 + 16));
  launch_widen32((const uint8_t *)col.values + narrow_bytes(col.type, e0), w->p, slots, widen_mode(col.type), g_ctx.n_cu,
                 st->stream);
  out->type = widened_type(col.type);
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
    // an exact set's lists hold (part of the fingerprint, row): what makes a table of them is the batch itself, through
    // the table path, while the batch is still there -- and the fingerprints alone once it has been released
    const bool exact_lists = ds.fp_exact_lists;
    ds.fp_exact_lists = false;
    if (c[kCntOutOfRange] != 0 || (exact_lists && !ds.retained.empty())) {
      if (ds.retained.empty())
        return fail(err, TGX_INTERNAL, "distinct: overflowed fingerprint lists and no batch to redo");
      HIP_TRY(hipMemsetAsync(ds.counters.p, 0, kNumDistinctCounters * sizeof(unsigned long long), st->stream));
      const DistinctTask &task = st->plan->distinct[slot];
      if (!task.tuple.empty()) {  // the retained columns are the tuple's components, in order
        std::vector<const tgx_column *> cols;
        for (const tgx_column &col : ds.retained) cols.push_back(&col);
        TupleDesc d;
        TGX_TRY(tuple_desc_of(cols, mult, &d, err));
        d.key = st->plan->fp_key;
        TGX_TRY(hash_ensure(st, ds, mult, (uint64_t)d.length, err));
        if (ds.exact) {
          TGX_TRY(key_store_measure_begin(st, ds, (uint64_t)d.length, err));
          launch_exact_measure_tuple(d, ds.key_cursor.as<unsigned long long>() + 1, st->stream);
          TGX_TRY(key_store_reserve_measured(st, ds, err));
        }
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
        if (ds.exact) TGX_TRY(key_store_reserve_utf8(st, ds, col, err));
        const bool view = col.type == TGX_UTF8_VIEW;
        launch_distinct_utf8(col.offsets, col.data, view ? col.values : nullptr, view ? col.variadic : nullptr,
                             col.validity, col.offset, col.length, col.type == TGX_LARGE_UTF8, mult ? 1 : 0,
                             hash_view(ds), st->plan->fp_key, ds.counters.as<unsigned long long>(), st->stream);
      }
    } else if (exact_lists) {
      // the batch is gone (tgx_finalize handed it back): the keys go on as their 128-bit fingerprints
      FpLists l1, l2;
      fp_views(ds, &l1, &l2);
      TGX_TRY(hash_ensure(st, ds, mult, c[kCntDistinct], err));
      unsigned long long cur = 0;
      HIP_TRY(hipMemcpyAsync(&cur, ds.key_cursor.p, sizeof(cur), hipMemcpyDeviceToHost, st->stream));
      HIP_TRY(hipStreamSynchronize(st->stream));
      // (one entry of two words per distinct fingerprint; values that shared one were told apart by the lists and
      //  are one entry from here on)
      TGX_TRY(key_store_ensure(st, ds, cur, 2 * (c[kCntDistinct] + 1), err));
      ds.key_words_ub = cur + 2 * (c[kCntDistinct] + 1);
      TGX_TRY(key_store_measure_begin(st, ds, (uint64_t)kFpFan * kFpFan * l2.cap, err));  // (an item: a record's place)
      HIP_TRY(hipMemsetAsync(ds.counters.p, 0, 2 * sizeof(unsigned long long), st->stream));  // (counted again as they go in)
      launch_fp_demote(l2, ds.fp_fb_lo.as<uint32_t>(), hash_view(ds), mult ? 1 : 0, ds.counters.as<unsigned long long>(),
                       st->stream);
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
  ds.remembered = false;  // (the range was not the column's)
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
    // (an exact set on the lists: the batch is about to be released -- its keys move into the table, bytes and all)
    if ((ds.speculative || ds.fp_staged) && !ds.retained.empty() &&
        (all[k * kNumDistinctCounters + kCntOutOfRange] != 0 || (ds.fp_staged && ds.fp_exact_lists)))
      TGX_TRY(distinct_resolve(st, k, err));
    else {
      ds.retained.clear();
      ds.outliers_possible = false;  // (the counters have just said so)
    }
  }
  return TGX_OK;
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
  if (ds.exact && wide) {  // incoming keys are entries of two words (they are only their fingerprints)
    unsigned long long cur = 0;
    HIP_TRY(hipMemcpyAsync(&cur, ds.key_cursor.p, sizeof(cur), hipMemcpyDeviceToHost, st->stream));
    HIP_TRY(hipStreamSynchronize(st->stream));
    TGX_TRY(key_store_ensure(st, ds, cur, 2 * n, err));
    ds.key_words_ub = cur + 2 * n;
    TGX_TRY(key_store_measure_begin(st, ds, n, err));  // (room in the pending list, its fill zeroed)
  }
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

tgx_status distinct_slot_of(const tgx_plan *plan, tgx_state *st, size_t spec_index, size_t *slot,
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

