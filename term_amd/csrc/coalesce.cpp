// coalesce.cpp -- small batches are noted, gathered into one column per flush, then take the ordinary pass; the copy
// pool behind the pinned arenas.  Split off tgx_api.cpp in round 4; see api_internal.h.
#include "api_internal.h"

// ------------------------------------------------------------------------------------------------
// coalescing: small batches are noted, gathered into one column per flush, then take the ordinary pass
// (internal.h, Coalescer; kernels/gather.hip).  Reference shape: DataFusion's `batch_size: 8192`
// (TG/core/context.rs:28-38) -- what `execute_stream()` hands a drop-in.

// The copy of a HOST batch's windows into the pinned arena is the only per-row work tgx_update does for a coalesced
// batch, and one core moves about 27 GB/s: a helper thread takes half of every batch's bytes (the calling thread the
// other half), which is what brings a stream of 8192-row batches near the PCIe rate.  The helper spins for a short
// while after a job -- batches of a stream arrive every few microseconds, a condition-variable wake-up costs more
// than a batch -- and then sleeps.  TGX_COPY_THREADS=0 keeps every copy on the calling thread.
namespace {
typedef tgx::CoalesceCopy CopyJob;
void copy_piece(CopyJob &j);  // (below: the copy, and the MIN / MAX of a key column's piece)
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
  bool finished(int w) const { return w_[w].done.load(std::memory_order_acquire) == w_[w].ticket; }  // (claimed, posted)

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
      for (size_t q = 0; q < k.n; q++) copy_piece(const_cast<CopyJob &>(k.jobs[q]));
      k.done.store(seen, std::memory_order_release);
    }
  }
  std::atomic<bool> stop_{false};
  Worker w_[kMaxWorkers];
  std::vector<std::thread> threads_;
  const int n_workers_;
};
}  // namespace

// a copy that does not pull the destination into the cache first (the arena is written once and read by the DMA
// engine): glibc's memcpy takes its streaming path only for copies of several MiB
void stream_copy(void *dst, const void *src, size_t bytes) {
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

namespace {
// narrow integers / Boolean bits as the Int64 values they stand for (a HOST window on its way into the arena)
void widen_copy(int64_t *dst, const void *src, int64_t n, int mode, int bit0) {
  switch (mode) {
    case 2: { const int8_t *s = (const int8_t *)src; for (int64_t i = 0; i < n; i++) dst[i] = s[i]; break; }
    case 3: { const int16_t *s = (const int16_t *)src; for (int64_t i = 0; i < n; i++) dst[i] = s[i]; break; }
    case 4: { const uint8_t *s = (const uint8_t *)src; for (int64_t i = 0; i < n; i++) dst[i] = s[i]; break; }
    case 5: { const uint16_t *s = (const uint16_t *)src; for (int64_t i = 0; i < n; i++) dst[i] = s[i]; break; }
    case 6: { const uint32_t *s = (const uint32_t *)src; for (int64_t i = 0; i < n; i++) dst[i] = s[i]; break; }
    default: {  // 7: bits
      const uint8_t *s = (const uint8_t *)src;
      for (int64_t i = 0; i < n; i++) {
        const int64_t b = bit0 + i;
        dst[i] = (s[b >> 3] >> (b & 7)) & 1;
      }
    }
  }
}
void copy_piece(CopyJob &j) {
  if (j.widen)
    widen_copy((int64_t *)j.dst, j.src, (int64_t)(j.bytes / 8), j.widen, j.src_bit0);
  else
    stream_copy(j.dst, j.src, j.bytes);
  if (j.mm_col >= 0)  // (of the Int64 values: a widened window's are in `dst`)
    host_minmax_i64((const int64_t *)(j.widen ? j.dst : j.src), j.mm_validity, j.mm_bit0, (int64_t)(j.bytes / 8), &j.lo, &j.hi);
}
}  // namespace

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
void host_minmax_i64_plain(const int64_t *v, const uint8_t *validity, int64_t bit0, int64_t n, int64_t *lo,
                                  int64_t *hi) {
  TGX_MINMAX_BODY
}
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
__attribute__((target("avx2"))) static void host_minmax_i64_avx2(const int64_t *v, const uint8_t *validity, int64_t bit0,
                                                                 int64_t n, int64_t *lo, int64_t *hi) {
  TGX_MINMAX_BODY
}
#endif
void host_minmax_i64(const int64_t *v, const uint8_t *validity, int64_t bit0, int64_t n, int64_t *lo, int64_t *hi) {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) return host_minmax_i64_avx2(v, validity, bit0, n, lo, hi);
#endif
  host_minmax_i64_plain(v, validity, bit0, n, lo, hi);
}

tgx_status coalesce_arena_ready(tgx_state *st, tgx_error *err) {
  Coalescer &co = st->coalesce;
  bind_thread();
  const int k = co.arena_cur;
  if (co.arena_busy[k]) {  // the flush that used this arena two turns ago (long done)
    HIP_TRY(hipEventSynchronize(co.arena_event[k]));
    co.arena_busy[k] = false;
  }
  if (co.arena_cap[k] < co.arena_want) {
    pinned_free(co.arena_host[k], co.arena_cap[k]);
    co.arena_host[k] = nullptr;
    co.arena_cap[k] = 0;
    HIP_TRY(pinned_alloc(&co.arena_host[k], co.arena_want));
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
tgx_status coalesce_prepare_window(const tgx_column &c, int64_t nrows, const CoalesceColumn &cc, int col,
                                          WindowPrep *w, tgx_error *err) {
  if (c.type == TGX_UTF8_VIEW) {
    const int32_t *v = (const int32_t *)c.values + (size_t)c.offset * 4;
    // (a window's long views point into one buffer, two where it crosses from one into the next: the buffer of the view
    // before is looked at first, its size kept at hand)
    int k = -1;
    int32_t kb = -1;
    int64_t ksize = 0;
    for (int64_t i = 0; i < nrows; i++, v += 4) {
      const int32_t len = v[0];
      if (len <= 12) continue;
      if (c.validity && !((c.validity[(c.offset + i) >> 3] >> ((c.offset + i) & 7)) & 1)) continue;
      const int32_t b = v[2];
      const int64_t off = v[3], end = off + len;
      if (b == kb) {
        if (off < 0 || end > ksize)
          return fail(err, TGX_INVALID_ARGUMENT, "column %d: a view of row %lld points outside its data buffers", col, (long long)i);
        if (off < w->vb_min[k]) w->vb_min[k] = off;
        if (end > w->vb_end[k]) w->vb_end[k] = end;
        continue;
      }
      if (b < 0 || b >= c.n_variadic || off < 0 || end > c.variadic_sizes[b])
        return fail(err, TGX_INVALID_ARGUMENT, "column %d: a view of row %lld points outside its data buffers", col, (long long)i);
      k = 0;
      while (k < w->vb_count && w->vb_index[k] != b) k++;
      kb = b;
      ksize = c.variadic_sizes[b];
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
size_t coalesce_host_bytes(const tgx_plan *plan, const tgx_column *columns, int64_t nrows,
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
    } else if (plan->reads_values[i]) {  // (a column only completeness / size look at brings its validity alone)
      total += (size_t)nrows * (is_numeric32(c.type) ? 4 : 8) + 64;  // (narrow integers / Booleans arrive widened: 8)
    }
  }
  return total;
}

void run_copy_jobs(Coalescer &co, const std::vector<CoalesceCopy> &jobs);
void coalesce_start_copies(Coalescer &co);
void coalesce_finish_copies(Coalescer &co, bool fold);
namespace {
size_t coalesce_eager_bytes();
}

// pending HOST bytes that trigger a flush (below); TGX_COALESCE_FLUSH_HOST_BYTES overrides
static size_t coalesce_flush_host_bytes() {
  static const size_t v = [] {
    const char *e = getenv("TGX_COALESCE_FLUSH_HOST_BYTES");
    return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)32 << 20;
  }();
  return v;
}

tgx_status coalesce_append(const tgx_plan *plan, tgx_state *st, const tgx_column *columns, int64_t nrows,
                                  const BatchTraits &traits, bool *taken, tgx_error *err) {
  Coalescer &co = st->coalesce;
  *taken = false;
  if (co.cols.size() != (size_t)plan->n_columns_needed) {
    co.cols.resize(plan->n_columns_needed);
    for (auto &cc : co.cols) cc.segs.reserve(kCoalesceFlushBatches);
  }
  const bool any_host = traits.any_host;
  // (a flush in here empties the pending lists -- the dictionaries' too: what the windows bring is then looked at again)
  // (scratch that lives with the thread: tgx_update notes a DEVICE batch in half a microsecond, and three heap
  // allocations per call were a third of that)
  static thread_local std::vector<WindowPrep> prep_tls;
  static thread_local std::vector<size_t> rb_segs, rb_dict_segs;
  static thread_local std::vector<int64_t> rb_data_bytes;
  if (prep_tls.size() < (size_t)plan->n_columns_needed) prep_tls.resize(plan->n_columns_needed);
  std::vector<WindowPrep> &prep = prep_tls;
  for (int attempt = 0;; attempt++) {
    const uint64_t flushes_before = co.flushes;
    for (int i = 0; traits.any_strings && i < plan->n_columns_needed; i++) {
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
        // (straight to the size a stream settles at -- a flush's worth plus a batch -- instead of doubling its way
        //  there: every growth pins a new arena, and 8 + 16 + 32 + 64 MB cost a first stream 10 - 20 ms)
        co.arena_want = std::min(kCoalesceArenaMax,
                                 std::max({co.arena_want * 2, need, coalesce_flush_host_bytes() + need + ((size_t)1 << 20)}));
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
    for (int i = 0; traits.any_strings && i < plan->n_columns_needed; i++) {
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
  for (int i = 0; traits.any_strings && i < plan->n_columns_needed; i++) {
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
    std::vector<size_t> &segs, &dict_segs;
    std::vector<int64_t> &data_bytes;
    size_t arena_used, deferred;
    bool armed;
    // (`needed` is false for a batch of DEVICE numeric windows: its segments go into reserved room, nothing can throw)
    Rollback(Coalescer &c, std::vector<size_t> &s, std::vector<size_t> &d, std::vector<int64_t> &b, bool needed)
        : co(c), segs(s), dict_segs(d), data_bytes(b), arena_used(c.arena_used), deferred(c.deferred.size()), armed(needed) {
      if (!needed) return;
      segs.clear();
      dict_segs.clear();
      data_bytes.clear();
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
        if (!co.cols[i].stretches.empty()) {  // (a Utf8View column: one entry per segment)
          size_t keep = 0;
          for (const CoalesceSegment &sg : co.cols[i].segs) keep = std::max(keep, (size_t)(sg.stretches + 1));
          co.cols[i].stretches.resize(keep);
        }
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
      co.deferred.resize(std::min(deferred, co.deferred.size()));  // (a flush on the way may have emptied it)
    }
  } rollback(co, rb_segs, rb_dict_segs, rb_data_bytes, any_host || traits.any_strings);
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
  const size_t deferred_from = co.deferred.size();  // (after any flush on the way)
  char *ah = any_host ? (char *)co.arena_host[co.arena_cur] : nullptr;
  const char *ad = any_host ? (const char *)co.arena_dev[co.arena_cur].p : nullptr;
  std::vector<CopyJob> &jobs = co.copy_jobs;
  jobs.clear();
  // TGX_MEM_HOST_RETAINED: the caller keeps the windows as they are until the next flushing call -- their copies (and
  // the key columns' MIN / MAX) wait for the flush, which runs all of them together on every copy thread
  const bool defer = traits.retained;
  auto to_arena = [&](const void *src, size_t bytes) -> const void * {  // returns the DEVICE twin's address
    const size_t at = (co.arena_used + 63) & ~(size_t)63;
    (defer ? co.deferred : jobs).push_back({ah + at, src, bytes});  // (jobs: copied below, shared with the helper threads)
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
      CoalesceStretches vs;
      vs.vb_count = prep[i].vb_count;
      for (int k = 0; k < prep[i].vb_count; k++) {
        vs.vb_index[k] = prep[i].vb_index[k];
        vs.vb_min[k] = prep[i].vb_min[k];
        vs.vb_len[k] = prep[i].vb_end[k] - prep[i].vb_min[k];
        vs.vb_src[k] = (const uint8_t *)to_arena(c.variadic[prep[i].vb_index[k]] + prep[i].vb_min[k], (size_t)vs.vb_len[k]);
        sg.data_len += (vs.vb_len[k] + 15) & ~(int64_t)15;  // (every stretch lands 16-byte aligned)
      }
      sg.stretches = (int32_t)cc.stretches.size();
      cc.stretches.push_back(vs);
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
    } else if (c.values && !plan->reads_values[i]) {
      // the plan reads the column's validity only (completeness, size): no value crosses the bus -- an Int8 column
      // widened here cost 8x its size in host copy, PCIe and gather traffic per flush; the flush's view has
      // no values, like the `bare` column of the immediate path
    } else if (c.values && is_narrow_int(c.type)) {
      // (HOST: update_validate) widened on the way into the arena: the flush sees an Int64 column
      cc.type = TGX_INT64;
      const bool bits = c.type == TGX_BOOL;
      sg.values = to_arena(bits ? (const uint8_t *)c.values + (c.offset >> 3)
                                : (const uint8_t *)c.values + narrow_bytes(c.type, c.offset), (size_t)nrows * 8);
      CopyJob &j = (defer ? co.deferred : jobs).back();
      j.widen = widen_mode(c.type);
      j.src_bit0 = bits ? (int32_t)(c.offset & 7) : 0;
      if (plan->key_column[i]) {
        if (cc.range_known) {  // (whoever widens the piece takes its MIN / MAX: the Int64 values are in the arena)
          j.mm_col = i;
          j.mm_validity = c.validity ? c.validity + (c.offset >> 3) : nullptr;
          j.mm_bit0 = c.offset & 7;
        }
      }
    } else if (c.values) {
      const size_t ew = is_numeric32(c.type) ? 4 : 8;
      const uint8_t *v0 = (const uint8_t *)c.values + (size_t)c.offset * ew;
      sg.values = host ? to_arena(v0, (size_t)nrows * ew) : (const void *)v0;
      if (plan->key_column[i] && c.type == TGX_INT64) {
        if (host && cc.range_known && defer) {
          CopyJob &j = co.deferred.back();  // (this window's copy, noted just above: its copier takes the MIN / MAX)
          j.mm_col = i;
          j.mm_validity = c.validity ? c.validity + (c.offset >> 3) : nullptr;
          j.mm_bit0 = c.offset & 7;
        } else if (host && cc.range_known)
          host_minmax_i64((const int64_t *)v0, c.validity ? c.validity + (c.offset >> 3) : nullptr, c.offset & 7, nrows,
                          &cc.range_lo, &cc.range_hi);
        else
          cc.range_known = false;
      }
    }
    cc.segs.push_back(sg);
  }
  run_copy_jobs(co, jobs);
  rollback.armed = false;
  if (defer) {
    for (size_t q = deferred_from; q < co.deferred.size(); q++) co.deferred_bytes += co.deferred[q].bytes;
    if (coalesce_eager_bytes() && co.deferred_bytes >= coalesce_eager_bytes()) coalesce_start_copies(co);
  }
  co.rows += nrows;
  co.batches += 1;
  co.coalesced_batches += 1;
  st->batches++;
  *taken = true;
  if (co.rows >= (co.flush_rows > 0 ? co.flush_rows : kCoalesceFlushRows) || co.batches >= kCoalesceFlushBatches)
    return coalesce_flush(st, err);
  // a HOST stream is flushed in pieces of a few tens of MB: the upload of one piece then runs beside the noting and
  // copying of the next (with 4 Mi-row flushes an 8 Mi-row table was two flushes: nothing overlapped; a piece of 32 MB
  // is 0.6 ms of PCIe time, against ~0.1 ms of launches per flush)
  // (a stream's first pieces are short ones -- 8, 16 MB, then the full size: nothing is on the link until the first
  //  goes, and a stream of a few tens of MB should not end with most of itself still to upload)
  if (any_host && co.flush_rows == 0 &&
      co.arena_used >= (co.host_flushes < 2 ? std::min(coalesce_flush_host_bytes(), (size_t)8 << (20 + co.host_flushes))
                                            : coalesce_flush_host_bytes()))
    return coalesce_flush(st, err);
  return TGX_OK;
}

namespace {
// `jobs` cut into `shares` equal shares of the bytes (a job that straddles a boundary is cut at a multiple of 64 bytes),
// one behind the other in `cut`: share s is cut[first[s] .. first[s + 1])
void cut_shares(const std::vector<CoalesceCopy> &jobs, size_t total, int shares, std::vector<CopyJob> &cut, size_t *first) {
  cut.clear();
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
      cut.push_back(CopyJob((char *)j.dst + at, (const char *)j.src + at, take));
      if (j.widen) {  // (`at` counts destination bytes, a multiple of 64: 8 values, one byte of Boolean bits)
        CopyJob &piece = cut.back();
        piece.widen = j.widen;
        piece.src_bit0 = j.src_bit0;
        const size_t e = at / 8;
        piece.src = (const char *)j.src + (j.widen == 7 ? e / 8 : j.widen == 6 ? e * 4 : (j.widen == 3 || j.widen == 5) ? e * 2 : e);
      }
      if (j.mm_col >= 0) {  // (cuts are multiples of 64 bytes = 8 values: whole validity bytes further on)
        CopyJob &piece = cut.back();
        piece.mm_col = j.mm_col;
        const int64_t bit = j.mm_bit0 + (int64_t)(at / 8);
        piece.mm_validity = j.mm_validity ? j.mm_validity + (bit >> 3) : nullptr;
        piece.mm_bit0 = bit & 7;
      }
      at += take;
      room -= std::min(room, take);
    }
  }
  while (share + 1 < shares) first[++share] = cut.size();
  first[shares] = cut.size();
}
void fold_range(Coalescer &co, const CopyJob &j) {  // a key column's piece: its MIN / MAX into the column's pending range
  if (j.mm_col < 0) return;
  CoalesceColumn &cc = co.cols[j.mm_col];
  cc.range_lo = std::min(cc.range_lo, j.lo);
  cc.range_hi = std::max(cc.range_hi, j.hi);
}
// TGX_COALESCE_EAGER_BYTES: retained windows are handed to the copy threads every so many bytes (0: at the flush only)
size_t coalesce_eager_bytes() {
  static const size_t v = [] {
    const char *e = getenv("TGX_COALESCE_EAGER_BYTES");
    return e ? (size_t)atoll(e) : (size_t)4 << 20;
  }();
  return v;
}
}  // namespace

// The retained windows noted so far go to the copy threads that are idle right now, and the caller goes on noting
// batches: with the copies left to the flush the caller's thread stood still for a flush's worth of memcpy every 32 MB
// and the stream ran at three quarters of what the link takes (its upload ran beside the NEXT piece's noting AND
// copying); now a piece's copies run beside its own noting.  (The arena is not touched by anyone else until the flush:
// it is sized when its first window is noted.)
void coalesce_start_copies(Coalescer &co) {
  // workers whose share is done go back to the pool (an earlier call's, or another state's, may want them)
  for (Coalescer::Inflight &f : co.inflight) {
    if (f.helpers < 0) continue;
    CopyPool *pool = CopyPool::get();
    bool all = pool != nullptr;
    for (int w = 0; all && w < f.helpers; w++) all = pool->finished(f.ids[w]);
    if (all) {
      for (int w = 0; w < f.helpers; w++) pool->release(f.ids[w]);
      f.helpers = -1;  // (its pieces wait for the flush: their MIN / MAX are folded there)
    }
  }
  if (co.deferred.empty()) return;
  CopyPool *pool = CopyPool::get();
  if (!pool) return;
  Coalescer::Inflight f;
  f.helpers = pool->claim(f.ids);
  if (f.helpers == 0) return;  // (all busy: the windows wait for the next call, or for the flush)
  co.inflight.emplace_back();
  Coalescer::Inflight &slot = co.inflight.back();
  slot.helpers = f.helpers;
  for (int w = 0; w < f.helpers; w++) slot.ids[w] = f.ids[w];
  size_t first[CopyPool::kMaxWorkers + 2];
  cut_shares(co.deferred, co.deferred_bytes, slot.helpers, slot.cut, first);
  for (int w = 0; w < slot.helpers; w++) pool->post(slot.ids[w], slot.cut.data() + first[w], first[w + 1] - first[w]);
  co.deferred.clear();
  co.deferred_bytes = 0;
}

// the flush (or a reset): every started copy is through before anybody looks at the arena -- or lets go of it
void coalesce_finish_copies(Coalescer &co, bool fold) {
  for (Coalescer::Inflight &f : co.inflight) {
    if (f.helpers > 0) {
      CopyPool *pool = CopyPool::get();
      for (int w = 0; pool && w < f.helpers; w++) pool->wait_and_release(f.ids[w]);
    }
    if (fold)
      for (const CopyJob &piece : f.cut) fold_range(co, piece);
  }
  co.inflight.clear();
}

// the copies of `jobs` shared between the calling thread and the copy threads that are idle right now
void run_copy_jobs(Coalescer &co, const std::vector<CoalesceCopy> &jobs) {
  if (!jobs.empty()) {
    size_t total = 0;
    for (const CopyJob &j : jobs) total += j.bytes;
    CopyPool *pool = total >= (64u << 10) ? CopyPool::get() : nullptr;
    int ids[CopyPool::kMaxWorkers];
    const int helpers = pool ? pool->claim(ids) : 0;  // (whoever is idle right now: other states may hold the rest)
    auto fold = [&](const CopyJob &j) { fold_range(co, j); };
    if (helpers == 0) {
      for (const CopyJob &j0 : jobs) {
        CopyJob j = j0;
        copy_piece(j);
        fold(j);
      }
    } else {
      // equal shares of the bytes: the first for the caller, one for every claimed worker
      const int shares = helpers + 1;
      std::vector<CopyJob> &cut = co.copy_tail;  // all shares one behind the other; first[s] = where share s begins
      size_t first[CopyPool::kMaxWorkers + 2];
      cut_shares(jobs, total, shares, cut, first);
      for (int w = 0; w < helpers; w++) pool->post(ids[w], cut.data() + first[w + 1], first[w + 2] - first[w + 1]);
      for (size_t q = first[0]; q < first[1]; q++) copy_piece(cut[q]);
      for (int w = 0; w < helpers; w++) pool->wait_and_release(ids[w]);
      for (const CopyJob &piece : cut) fold(piece);
    }
  }
}

// Region set `set` is about to be overwritten: views retained into it (a sampled-range key set keeps its batches for
// a later repair) are dropped when the counters snapshot taken after the flush that filled it shows nothing to repair;
// otherwise the repair runs now.
tgx_status coalesce_release_set(tgx_state *st, int set, tgx_error *err) {
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
    pinned_free(co.desc_host[ar], co.desc_cap[ar]);
    co.desc_host[ar] = nullptr;
    co.desc_cap[ar] = 0;
    const size_t want = std::max<size_t>(2 * n_segs * sizeof(GatherSeg), 64u << 10);
    HIP_TRY(pinned_alloc(&co.desc_host[ar], want));
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
    if (has_values) HIP_TRY(cc.values[set].reserve_roomy((size_t)(rows + 1) * ew + 64));
    if (cc.any_validity) {
      const size_t vb = ((size_t)rows + 31) / 32 * 4 + 64;
      HIP_TRY(cc.validity[set].reserve_roomy(vb));
      HIP_TRY(hipMemsetAsync(cc.validity[set].p, 0, vb, st->stream));
    }
    if (str || vw) HIP_TRY(cc.data[set].reserve_roomy((size_t)cc.data_bytes + 64));
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
        const CoalesceStretches &vs = cc.stretches[(size_t)sg.stretches];
        d.vb_count = vs.vb_count;
        int64_t at = data_at;
        for (int k = 0; k < vs.vb_count; k++) {
          d.vb_index[k] = vs.vb_index[k];
          d.vb_min[k] = vs.vb_min[k];
          d.vb_len[k] = vs.vb_len[k];
          d.vb_src[k] = vs.vb_src[k];
          d.vb_base[k] = at;
          at += (vs.vb_len[k] + 15) & ~(int64_t)15;
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
      HIP_TRY(cd.values[set].reserve_roomy((size_t)(cd.entries + 1) * dw + 64));
      HIP_TRY(hipMemsetAsync(cd.values[set].p, 0, (size_t)(cd.entries + 1) * dw, st->stream));  // (an empty dictionary: offset 0)
      if (cd.any_validity) {
        const size_t vb = ((size_t)cd.entries + 31) / 32 * 4 + 64;
        HIP_TRY(cd.validity[set].reserve_roomy(vb));
        HIP_TRY(hipMemsetAsync(cd.validity[set].p, 0, vb, st->stream));
      }
      HIP_TRY(cd.data[set].reserve_roomy((size_t)cd.data_bytes + 64));
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
  // TGX_MEM_HOST_RETAINED windows: what the copy threads have been given while the batches were noted is waited for,
  // the rest goes into the arena now, all of it together; the key columns' MIN / MAX beside them
  coalesce_finish_copies(co, true);
  if (!co.deferred.empty()) {
    run_copy_jobs(co, co.deferred);
    co.deferred.clear();
  }
  co.deferred_bytes = 0;
  if (co.arena_used) {
    // (the arena's device twin is free: whoever read it last was waited for when the arena's first window was noted)
    static const bool own_stream = !(getenv("TGX_COALESCE_COPY_STREAM") && atoi(getenv("TGX_COALESCE_COPY_STREAM")) == 0);
    if (own_stream && !co.copy_stream) HIP_TRY(stream_acquire(&co.copy_stream, false));
    if (own_stream && !co.upload_done[ar]) HIP_TRY(hipEventCreateWithFlags(&co.upload_done[ar], hipEventDisableTiming));
    hipStream_t up = own_stream ? co.copy_stream : st->stream;
    HIP_TRY(hipMemcpyAsync(co.arena_dev[ar].p, co.arena_host[ar], co.arena_used, hipMemcpyHostToDevice, up));
    if (own_stream) {
      HIP_TRY(hipEventRecord(co.upload_done[ar], up));
      HIP_TRY(hipStreamWaitEvent(st->stream, co.upload_done[ar], 0));
    }
    co.host_flushes++;
  }
  HIP_TRY(hipMemcpyAsync(co.desc_dev[ar].p, gs, g * sizeof(GatherSeg), hipMemcpyHostToDevice, st->stream));
  {
    ProfScope ps(st, "gather", 0);
    // workgroups per segment: one per ~8192 rows of the longest window (64 KB of 8-byte values), at most 32
    int64_t longest = 0;
    for (size_t q = 0; q < g; q++) longest = std::max<int64_t>(longest, gs[q].length);
    const int parts = (int)std::min<int64_t>(32, std::max<int64_t>(1, (longest + 8191) / 8192));
    launch_gather_segments(co.desc_dev[ar].as<GatherSeg>(), (int)g, parts, st->stream);
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
    ds.batch_bytes_known = false;
    if (!t.tuple.empty() || t.approx_only) continue;
    const CoalesceColumn &cc = co.cols[t.column];
    ds.flush_device_keys = false;
    // a string key column: the flush's value bytes were added up as its windows were noted (an exact key set sizes its
    // key store from them without asking the device); a dictionary column: the bytes of the flush's dictionary
    if (is_string(cc.type) && !cc.segs.empty()) {
      ds.batch_bytes_known = true;
      ds.batch_data_bytes = cc.data_bytes;
    } else if (cc.type == TGX_DICT32_UTF8 && cc.dict && !cc.segs.empty()) {
      ds.batch_bytes_known = true;
      ds.batch_data_bytes = cc.dict->data_bytes;
    }
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
    cc.stretches.clear();
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
    ds.batch_bytes_known = false;
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
      pinned_free(co.snap_host[set], co.snap_cap[set]);
      co.snap_host[set] = nullptr;
      co.snap_cap[set] = 0;
      HIP_TRY(pinned_alloc(&co.snap_host[set], bytes + 256));
      co.snap_cap[set] = bytes + 256;
    }
    if (!co.snap_event[set]) HIP_TRY(hipEventCreateWithFlags(&co.snap_event[set], hipEventDisableTiming));
    HIP_TRY(hipMemcpyAsync(co.snap_host[set], st->d_distinct_counters.p, bytes, hipMemcpyDeviceToHost, st->stream));
    HIP_TRY(hipEventRecord(co.snap_event[set], st->stream));
    co.snap_pending[set] = true;
  }
  return TGX_OK;
}

void coalesce_drop(tgx_state *st) {  // reset / destroy: pending batches are forgotten
  Coalescer &co = st->coalesce;
  for (auto &cc : co.cols) {
    cc.segs.clear();
    cc.stretches.clear();
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
  coalesce_finish_copies(co, false);  // (the copy threads let go of the arena before anybody else may have it)
  co.host_flushes = 0;
  co.deferred.clear();
  co.deferred_bytes = 0;
  co.snap_pending[0] = co.snap_pending[1] = false;
}

void copy_pool_shutdown() { CopyPool::shutdown(); }
