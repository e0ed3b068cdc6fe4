// A process-wide cache of device (and pinned host) allocations behind DevBuf / PinnedBuf (round 5).
//
// A reference user calls `ValidationSuite::run` once per table (TG/core/suite.rs:399): the state of a run is created,
// fed, read and dropped.  A state of the headline suite owns ~10 GB of lists and bitmaps; `hipMalloc` of GB-sized
// buffers costs milliseconds and `hipFree` waits for the whole device, so a process that validates table after table
// paid that once per table.  Freed blocks are kept here instead, by size class, and handed to the next state:
// from the second state of a process on, `tgx_state_create` .. `tgx_finalize` performs no hipMalloc at all.
//
//  * size classes: 8 per octave (a request is rounded up by at most 12.5 %), 256 bytes at least;
//  * a block enters the cache only when nothing on the device can still touch it: the releasing thread is inside a
//    QuiescedScope (tgx_state_destroy has waited for the device once) or the release itself waits for the device --
//    what hipFree did implicitly;
//  * bounded: TGX_DEVICE_CACHE_MAX_BYTES (default: a quarter of the device's memory, 64 GiB at most); a block that
//    does not fit is freed; an allocation that fails empties the cache and tries again;
//  * tgx_trim() frees everything cached, tgx_shutdown() too;
//  * TGX_DEVICE_CACHE=0 turns the cache off (every release is a hipFree, as before round 5);
//    TGX_DEVICE_CACHE_POISON=1 fills every block handed out with 0xA5 (tests: nothing may rely on the zero pages a
//    fresh hipMalloc happens to return).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "internal.h"

namespace tgx {
namespace {

struct Pool {
  std::mutex mu;
  std::map<size_t, std::vector<void *>> free_blocks;  // size class -> blocks
  uint64_t cached_bytes = 0, hits = 0, misses = 0, frees = 0;
};
Pool g_dev, g_host;

thread_local int tl_quiesced = 0;
// false once tgx_shutdown has emptied the cache (until the next tgx_init): what is released then -- a state destroyed
// after the shutdown -- goes back to the driver, not into a cache nobody will trim
std::atomic<bool> g_live{true};

bool cache_enabled() {
  static const bool on = [] {
    const char *e = getenv("TGX_DEVICE_CACHE");
    return !(e && e[0] == '0');
  }();
  return on;
}
bool poison() {
  static const bool on = [] {
    const char *e = getenv("TGX_DEVICE_CACHE_POISON");
    return e && e[0] == '1';
  }();
  return on;
}
// (on a stream of its own that does not synchronise with the others: a null-stream hipMemset waits for every blocking
//  stream of the device -- with ranks as threads of one process, one of them inside a collective's barrier, for ever)
void poison_block(void *p, size_t bytes) {
  static hipStream_t s = [] {
    hipStream_t q = nullptr;
    (void)hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
    return q;
  }();
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  (void)hipMemsetAsync(p, 0xA5, bytes, s);
  (void)hipStreamSynchronize(s);
}
uint64_t dev_limit() {
  static const uint64_t lim = [] {
    if (const char *e = getenv("TGX_DEVICE_CACHE_MAX_BYTES")) return (uint64_t)strtoull(e, nullptr, 10);
    size_t free_b = 0, total = 0;
    if (hipMemGetInfo(&free_b, &total) != hipSuccess) total = (size_t)64 << 30;
    return std::min<uint64_t>((uint64_t)total / 4, (uint64_t)64 << 30);
  }();
  return lim;
}
constexpr uint64_t kHostLimit = (uint64_t)2 << 30;  // pinned memory kept at most

}  // namespace

size_t cache_size_class(size_t bytes) {
  if (bytes <= 256) return 256;
  const int top = 63 - __builtin_clzll((unsigned long long)(bytes - 1));  // 2^top <= bytes - 1 < 2^(top+1)
  const size_t step = (size_t)1 << (top >= 3 ? top - 3 : 0);
  return (bytes + step - 1) / step * step;
}

QuiescedScope::QuiescedScope() { tl_quiesced++; }
QuiescedScope::~QuiescedScope() { tl_quiesced--; }

hipError_t dev_alloc(void **p, size_t *cap, size_t bytes) {
  const size_t cls = cache_enabled() ? cache_size_class(bytes) : bytes;
  if (cache_enabled()) {
    bool hit = false;
    {
      std::lock_guard<std::mutex> lock(g_dev.mu);
      auto it = g_dev.free_blocks.find(cls);
      if (it != g_dev.free_blocks.end() && !it->second.empty()) {
        *p = it->second.back();
        it->second.pop_back();
        g_dev.cached_bytes -= cls;
        g_dev.hits++;
        hit = true;
      } else {
        g_dev.misses++;
      }
    }
    if (hit) {
      *cap = cls;
      if (poison()) poison_block(*p, cls);
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(p, cls);
  if (e != hipSuccess && cache_enabled()) {
    (void)hipGetLastError();
    dev_cache_trim();  // (the cache may be what holds the memory)
    e = hipMalloc(p, cls);
  }
  if (e != hipSuccess) {
    *p = nullptr;
    return e;
  }
  *cap = cls;
  if (poison()) poison_block(*p, cls);
  return hipSuccess;
}

void dev_cache_set_live(bool on) { g_live.store(on); }

void dev_free(void *p, size_t cap) {
  if (!p) return;
  if (!cache_enabled() || !g_live.load() || cap != cache_size_class(cap) || cap > dev_limit()) {
    (void)hipFree(p);  // (waits for the device itself)
    return;
  }
  // nothing queued anywhere may still read or write the block when another state takes it
  if (tl_quiesced == 0) (void)hipDeviceSynchronize();
  {
    std::lock_guard<std::mutex> lock(g_dev.mu);
    if (g_dev.cached_bytes + cap <= dev_limit()) {
      g_dev.free_blocks[cap].push_back(p);
      g_dev.cached_bytes += cap;
      return;
    }
    g_dev.frees++;
  }
  (void)hipFree(p);
}

// (`bytes` of pinned_free must be the `bytes` the block was asked for with: both sides derive the size class from it)
hipError_t pinned_alloc(void **p, size_t bytes) {
  const size_t cls = cache_enabled() ? cache_size_class(bytes) : bytes;
  if (cache_enabled()) {
    std::lock_guard<std::mutex> lock(g_host.mu);
    auto it = g_host.free_blocks.find(cls);
    if (it != g_host.free_blocks.end() && !it->second.empty()) {
      *p = it->second.back();
      it->second.pop_back();
      g_host.cached_bytes -= cls;
      g_host.hits++;
      return hipSuccess;
    }
    g_host.misses++;
  }
  hipError_t e = hipHostMalloc(p, cls, hipHostMallocDefault);
  if (e != hipSuccess) *p = nullptr;
  return e;
}

void pinned_free(void *p, size_t bytes) {
  if (!p) return;
  // Only a release inside a QuiescedScope (tgx_state_destroy: the device has been waited for) keeps the block: the
  // others are the growth paths of the update / flush (arena, descriptors, snapshots, h_pinned), whose blocks are
  // guarded by their own events -- a device-wide wait there would stall every stream of the device, other states' and
  // the threaded ranks inside a collective included; hipHostFree only waits for what uses the block
  if (cache_enabled() && g_live.load() && tl_quiesced > 0) {
    const size_t cap = cache_size_class(bytes);
    std::lock_guard<std::mutex> lock(g_host.mu);
    if (g_host.cached_bytes + cap <= kHostLimit) {
      g_host.free_blocks[cap].push_back(p);
      g_host.cached_bytes += cap;
      return;
    }
  }
  (void)hipHostFree(p);
}

// ---- streams ----------------------------------------------------------------------------------------------------
// hipStreamCreate / hipStreamDestroy cost 0.4 - 0.5 ms each on this runtime (tools/trace_cold_step.sh): more than the
// whole device side of a 100 M-row x 8-column suite's key passes.  A destroyed state's own streams (idle: the device
// has been waited for) are kept and handed to the next state.
namespace {
std::mutex g_stream_mu;
std::vector<hipStream_t> g_streams[2];  // [0] default priority, [1] the highest
}  // namespace

hipError_t stream_acquire(hipStream_t *out, bool high_priority) {
  {
    std::lock_guard<std::mutex> lock(g_stream_mu);
    auto &pool = g_streams[high_priority ? 1 : 0];
    if (cache_enabled() && !pool.empty()) {
      *out = pool.back();
      pool.pop_back();
      return hipSuccess;
    }
  }
  if (!high_priority) return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  return hipStreamCreateWithPriority(out, hipStreamNonBlocking, hi);
}

void stream_release(hipStream_t s, bool high_priority) {
  if (!s) return;
  if (cache_enabled() && g_live.load()) {
    std::lock_guard<std::mutex> lock(g_stream_mu);
    auto &pool = g_streams[high_priority ? 1 : 0];
    if (pool.size() < 64) {
      pool.push_back(s);
      return;
    }
  }
  (void)hipStreamDestroy(s);
}

void dev_cache_trim() {
  {
    std::vector<hipStream_t> streams;
    {
      std::lock_guard<std::mutex> lock(g_stream_mu);
      for (auto &pool : g_streams) {
        streams.insert(streams.end(), pool.begin(), pool.end());
        pool.clear();
      }
    }
    for (hipStream_t s : streams) (void)hipStreamDestroy(s);
  }
  std::map<size_t, std::vector<void *>> dev, host;
  {
    std::lock_guard<std::mutex> lock(g_dev.mu);
    dev.swap(g_dev.free_blocks);
    g_dev.cached_bytes = 0;
  }
  {
    std::lock_guard<std::mutex> lock(g_host.mu);
    host.swap(g_host.free_blocks);
    g_host.cached_bytes = 0;
  }
  for (auto &kv : dev)
    for (void *p : kv.second) (void)hipFree(p);
  for (auto &kv : host)
    for (void *p : kv.second) (void)hipHostFree(p);
}

void dev_cache_stats(tgx_cache_stats *out) {
  memset(out, 0, sizeof(*out));
  {
    std::lock_guard<std::mutex> lock(g_dev.mu);
    out->device_cached_bytes = g_dev.cached_bytes;
    out->device_hits = g_dev.hits;
    out->device_misses = g_dev.misses;
    for (auto &kv : g_dev.free_blocks) out->device_cached_blocks += kv.second.size();
  }
  {
    std::lock_guard<std::mutex> lock(g_host.mu);
    out->pinned_cached_bytes = g_host.cached_bytes;
    out->pinned_hits = g_host.hits;
    out->pinned_misses = g_host.misses;
  }
}

}  // namespace tgx
