// allreduce.cpp -- the cross-rank step behind the C ABI: tgx_comm (RCCL over xGMI, or any transport of the caller's)
// and tgx_allreduce (include/tgx.h, "the cross-rank step").
//
// Reference contract: `AnalyzerState::merge` (TG/analyzers/traits.rs:160-170) applied across row shards.  What moves:
//   facts      one all-gather of a few scalars per DISTINCT column (MIN / MAX, kind of key set, rows, blob size)
//   key sets   ONE all-to-all of range-bitmap slices for all dense Int64 columns together (re-based on the agreed
//              global range while they are copied into the send buffer), hash-owner key records otherwise
//   states     one all-gather of the packed partial states, folded in rank order on every rank
// RCCL is bound at run time (dlopen of librccl.so.1 -- the copy already in the process when the host is PyTorch), so
// libtgx.so itself links against HIP only.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "internal.h"
#include "spearman_device.h"

using namespace tgx;

#define HIP_TRY(expr)                                                                                     \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess)                                                                                 \
      return fail(err, e_ == hipErrorOutOfMemory ? TGX_OUT_OF_MEMORY : TGX_DEVICE_ERROR, "%s failed: %s", \
                  #expr, hipGetErrorString(e_));                                                          \
  } while (0)
#define TGX_TRY(expr)            \
  do {                           \
    tgx_status s_ = (expr);      \
    if (s_ != TGX_OK) return s_; \
  } while (0)

// ------------------------------------------------------------------------------------------------ RCCL binding
namespace {
struct RcclApi {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  std::string error;
};

RcclApi *rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // the copy that is already mapped (a PyTorch host carries its own librccl.so.1) wins over a second one
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      const char *e = dlerror();
      api.error = std::string("librccl.so.1 cannot be loaded: ") + (e ? e : "unknown dlopen error");
      return;
    }
    api.handle = h;
    bool ok = true;
    auto sym = [&](const char *name) -> void * {
      void *p = dlsym(h, name);
      if (!p) {
        ok = false;
        api.error = std::string("librccl lacks ") + name;
      }
      return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    if (!ok) api.handle = nullptr;
  });
  return &api;
}
}  // namespace

struct tgx_comm {
  tgx_comm_ops ops;
  // RCCL transport
  RcclApi *api = nullptr;
  ncclComm_t nccl = nullptr;
  bool own_nccl = false;
  int last_nccl_error = 0;
  // scratch kept between steps: nothing is allocated per call once the sizes have settled
  DevBuf d_send, d_recv, d_small_send, d_small_recv;
  void *h_a = nullptr, *h_b = nullptr;  // pinned host staging
  size_t h_a_cap = 0, h_b_cap = 0;
  // the blob capacity the ranks agreed on for a plan (one collective per step once it has settled)
  const tgx_plan *blob_plan = nullptr;
  size_t blob_cap = 0;
  // every host wait behind a collective has a deadline (TGX_COLLECTIVE_TIMEOUT_MS, default two minutes): a peer that
  // died or left the step must surface as an error on the ranks that are still there, not as a process that hangs
  hipEvent_t wait_event = nullptr;
  ~tgx_comm() {
    if (wait_event) (void)hipEventDestroy(wait_event);
  }
};

namespace {
int32_t rccl_alltoallv(void *ctx, const void *send, const uint64_t *send_counts, void *recv,
                       const uint64_t *recv_counts, size_t elem_bytes, void *hip_stream) {
  tgx_comm *c = (tgx_comm *)ctx;
  const RcclApi *a = c->api;
  ncclResult_t r = a->GroupStart();
  size_t so = 0, ro = 0;
  for (int32_t p = 0; p < c->ops.world && r == ncclSuccess; p++) {
    const size_t sb = (size_t)send_counts[p] * elem_bytes, rb = (size_t)recv_counts[p] * elem_bytes;
    if (sb) r = a->Send((const char *)send + so, sb, ncclInt8, p, c->nccl, (hipStream_t)hip_stream);
    if (rb && r == ncclSuccess) r = a->Recv((char *)recv + ro, rb, ncclInt8, p, c->nccl, (hipStream_t)hip_stream);
    so += sb;
    ro += rb;
  }
  ncclResult_t e = a->GroupEnd();
  if (r == ncclSuccess) r = e;
  c->last_nccl_error = (int)r;
  return r == ncclSuccess ? 0 : 1;
}

// equal splits: grouped point-to-point sends, one per peer -- over xGMI every peer is one hop on its own link, so
// the 7 transfers of a rank run side by side
int32_t rccl_alltoall(void *ctx, const void *send, void *recv, size_t bytes_per_peer, void *hip_stream) {
  tgx_comm *c = (tgx_comm *)ctx;
  std::vector<uint64_t> counts((size_t)c->ops.world, (uint64_t)bytes_per_peer);
  return rccl_alltoallv(ctx, send, counts.data(), recv, counts.data(), 1, hip_stream);
}

int32_t rccl_allgather(void *ctx, const void *send, void *recv, size_t bytes, void *hip_stream) {
  tgx_comm *c = (tgx_comm *)ctx;
  ncclResult_t r = c->api->AllGather(send, recv, bytes, ncclInt8, c->nccl, (hipStream_t)hip_stream);
  c->last_nccl_error = (int)r;
  return r == ncclSuccess ? 0 : 1;
}

tgx_status comm_fail(tgx_comm *c, tgx_error *err, const char *what) {
  if (c->api && c->nccl && c->last_nccl_error)
    return fail(err, TGX_DEVICE_ERROR, "%s failed: %s", what, c->api->GetErrorString((ncclResult_t)c->last_nccl_error));
  return fail(err, TGX_DEVICE_ERROR, "%s failed in the transport", what);
}

int64_t collective_timeout_ms() {
  const char *e = getenv("TGX_COLLECTIVE_TIMEOUT_MS");
  const long long v = e && *e ? atoll(e) : 120000;
  return v > 0 ? v : 120000;
}

// waits until everything queued on `s` so far is through -- at most the collective deadline
tgx_status deadline_sync(tgx_comm *c, hipStream_t s, const char *what, tgx_error *err) {
  if (!c->wait_event) HIP_TRY(hipEventCreateWithFlags(&c->wait_event, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(c->wait_event, s));
  const auto t0 = std::chrono::steady_clock::now();
  const int64_t limit = collective_timeout_ms();
  for (uint64_t spins = 0;; spins++) {
    const hipError_t q = hipEventQuery(c->wait_event);
    if (q == hipSuccess) return TGX_OK;
    if (q != hipErrorNotReady)
      return fail(err, TGX_DEVICE_ERROR, "tgx_allreduce: waiting for %s failed: %s", what, hipGetErrorString(q));
    if ((spins & 63) != 63) continue;  // (polling costs a fraction of a microsecond: look at the clock now and then)
    const int64_t us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    // the usual wait is the rest of the step's scan, a few milliseconds at most: poll that long (a sleep of 20 us is
    // 70 with the timer's slack, and the step's tail is a handful of such waits), then back off
    if (us < 20000) continue;
    const int64_t ms = us / 1000;
    if (ms > limit)
      return fail(err, TGX_DEVICE_ERROR,
                  "tgx_allreduce: %s did not complete within %lld ms on rank %d of %d (a peer has failed or left the "
                  "step, or the transport is stuck); the state and the communicator are unusable",
                  what, (long long)limit, c->ops.rank, c->ops.world);
    std::this_thread::sleep_for(std::chrono::microseconds(ms < 100 ? 50 : 500));
  }
}

tgx_status pinned_reserve(void **p, size_t *cap, size_t bytes, tgx_error *err) {
  if (bytes <= *cap) return TGX_OK;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  HIP_TRY(hipHostMalloc(p, want, hipHostMallocDefault));
  *cap = want;
  return TGX_OK;
}

// ---- the three collectives over either kind of transport; `send` / `recv` are DEVICE buffers ----
tgx_status do_alltoall(tgx_comm *c, hipStream_t s, const void *d_send, void *d_recv, size_t per_peer, tgx_error *err) {
  const size_t total = per_peer * (size_t)c->ops.world;
  if (total == 0) return TGX_OK;
  if (c->ops.device_buffers) {
    if (c->ops.alltoall(c->ops.ctx, d_send, d_recv, per_peer, s) != 0) return comm_fail(c, err, "all-to-all");
    return TGX_OK;
  }
  TGX_TRY(pinned_reserve(&c->h_a, &c->h_a_cap, total, err));
  TGX_TRY(pinned_reserve(&c->h_b, &c->h_b_cap, total, err));
  HIP_TRY(hipMemcpyAsync(c->h_a, d_send, total, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (c->ops.alltoall(c->ops.ctx, c->h_a, c->h_b, per_peer, nullptr) != 0) return comm_fail(c, err, "all-to-all");
  HIP_TRY(hipMemcpyAsync(d_recv, c->h_b, total, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));  // h_b is reused by the next collective
  return TGX_OK;
}

tgx_status do_alltoallv(tgx_comm *c, hipStream_t s, const void *d_send, const uint64_t *sc, void *d_recv,
                        const uint64_t *rc, size_t elem, tgx_error *err) {
  size_t st = 0, rt = 0;
  for (int32_t p = 0; p < c->ops.world; p++) {
    st += (size_t)sc[p] * elem;
    rt += (size_t)rc[p] * elem;
  }
  if (c->ops.device_buffers) {
    if (c->ops.alltoallv(c->ops.ctx, d_send, sc, d_recv, rc, elem, s) != 0) return comm_fail(c, err, "all-to-all-v");
    return TGX_OK;
  }
  TGX_TRY(pinned_reserve(&c->h_a, &c->h_a_cap, std::max<size_t>(st, 16), err));
  TGX_TRY(pinned_reserve(&c->h_b, &c->h_b_cap, std::max<size_t>(rt, 16), err));
  if (st) HIP_TRY(hipMemcpyAsync(c->h_a, d_send, st, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (c->ops.alltoallv(c->ops.ctx, c->h_a, sc, c->h_b, rc, elem, nullptr) != 0) return comm_fail(c, err, "all-to-all-v");
  if (rt) HIP_TRY(hipMemcpyAsync(d_recv, c->h_b, rt, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  return TGX_OK;
}

// HOST in, HOST out (facts, blobs): `h_send` = bytes, `h_recv` = world * bytes.  `s` may be null for host transports.
tgx_status do_allgather_host(tgx_comm *c, hipStream_t s, const void *h_send, void *h_recv, size_t bytes, tgx_error *err) {
  if (!c->ops.device_buffers) {
    if (c->ops.allgather(c->ops.ctx, h_send, h_recv, bytes, nullptr) != 0) return comm_fail(c, err, "all-gather");
    return TGX_OK;
  }
  // through PINNED staging: copy in, collective and copy out are queued together and waited for once (copies from / to
  // pageable memory are each a round trip of their own: three waits per collective, ~0.1 ms of a 3.7 ms step)
  const size_t total = bytes * (size_t)c->ops.world;
  HIP_TRY(c->d_small_send.reserve(bytes + 16));
  HIP_TRY(c->d_small_recv.reserve(total + 16));
  TGX_TRY(pinned_reserve(&c->h_a, &c->h_a_cap, bytes, err));
  TGX_TRY(pinned_reserve(&c->h_b, &c->h_b_cap, total, err));
  memcpy(c->h_a, h_send, bytes);
  HIP_TRY(hipMemcpyAsync(c->d_small_send.p, c->h_a, bytes, hipMemcpyHostToDevice, s));
  if (c->ops.allgather(c->ops.ctx, c->d_small_send.p, c->d_small_recv.p, bytes, s) != 0)
    return comm_fail(c, err, "all-gather");
  HIP_TRY(hipMemcpyAsync(c->h_b, c->d_small_recv.p, total, hipMemcpyDeviceToHost, s));
  TGX_TRY(deadline_sync(c, s, "an all-gather", err));
  memcpy(h_recv, c->h_b, total);
  return TGX_OK;
}
}  // namespace

// ------------------------------------------------------------------------------------------------ comm handles
extern "C" tgx_status tgx_comm_create(const tgx_comm_ops *ops, tgx_comm **out, tgx_error *err) try {
  if (!ops || !out) return fail(err, TGX_INVALID_ARGUMENT, "ops/out is NULL");
  *out = nullptr;
  if (ops->world < 1 || ops->rank < 0 || ops->rank >= ops->world || ops->world > 256)
    return fail(err, TGX_INVALID_ARGUMENT, "bad rank %d / world %d (world must be 1..256)", ops->rank, ops->world);
  if (!ops->alltoall || !ops->alltoallv || !ops->allgather)
    return fail(err, TGX_INVALID_ARGUMENT, "a transport needs alltoall, alltoallv and allgather");
  tgx_comm *c = new tgx_comm();
  c->ops = *ops;
  *out = c;
  return TGX_OK;
} catch (...) {
  return abi_exception(err);
}

extern "C" tgx_status tgx_comm_rccl_unique_id(uint8_t id[TGX_RCCL_UNIQUE_ID_BYTES], tgx_error *err) try {
  static_assert(TGX_RCCL_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  if (!id) return fail(err, TGX_INVALID_ARGUMENT, "id is NULL");
  RcclApi *a = rccl_api();
  if (!a->handle) return fail(err, TGX_UNSUPPORTED, "%s", a->error.c_str());
  ncclUniqueId u;
  ncclResult_t r = a->GetUniqueId(&u);
  if (r != ncclSuccess) return fail(err, TGX_DEVICE_ERROR, "ncclGetUniqueId failed: %s", a->GetErrorString(r));
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return TGX_OK;
} catch (...) {
  return abi_exception(err);
}

static tgx_status make_rccl_comm(ncclComm_t nccl, bool own, int32_t rank, int32_t world, tgx_comm **out, tgx_error *err) {
  tgx_comm *c = new tgx_comm();
  c->api = rccl_api();
  c->nccl = nccl;
  c->own_nccl = own;
  memset(&c->ops, 0, sizeof(c->ops));
  c->ops.ctx = c;
  c->ops.rank = rank;
  c->ops.world = world;
  c->ops.device_buffers = 1;
  c->ops.alltoall = rccl_alltoall;
  c->ops.alltoallv = rccl_alltoallv;
  c->ops.allgather = rccl_allgather;
  *out = c;
  (void)err;
  return TGX_OK;
}

extern "C" tgx_status tgx_comm_create_rccl(const uint8_t id[TGX_RCCL_UNIQUE_ID_BYTES], int32_t rank, int32_t world,
                                           tgx_comm **out, tgx_error *err) try {
  if (!id || !out) return fail(err, TGX_INVALID_ARGUMENT, "id/out is NULL");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world || world > 256)
    return fail(err, TGX_INVALID_ARGUMENT, "bad rank %d / world %d (world must be 1..256)", rank, world);
  TGX_TRY(need_device(err));
  HIP_TRY(hipSetDevice(device_id()));  // ncclCommInitRank binds the calling thread's current device
  RcclApi *a = rccl_api();
  if (!a->handle) return fail(err, TGX_UNSUPPORTED, "%s", a->error.c_str());
  ncclUniqueId u;
  memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t nccl = nullptr;
  ncclResult_t r = a->CommInitRank(&nccl, world, u, rank);
  if (r != ncclSuccess) return fail(err, TGX_DEVICE_ERROR, "ncclCommInitRank failed: %s", a->GetErrorString(r));
  return make_rccl_comm(nccl, true, rank, world, out, err);
} catch (...) {
  return abi_exception(err);
}

extern "C" tgx_status tgx_comm_adopt_rccl(void *nccl_comm, int32_t rank, int32_t world, tgx_comm **out,
                                          tgx_error *err) try {
  if (!nccl_comm || !out) return fail(err, TGX_INVALID_ARGUMENT, "nccl_comm/out is NULL");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world || world > 256)
    return fail(err, TGX_INVALID_ARGUMENT, "bad rank %d / world %d (world must be 1..256)", rank, world);
  RcclApi *a = rccl_api();
  if (!a->handle) return fail(err, TGX_UNSUPPORTED, "%s", a->error.c_str());
  return make_rccl_comm((ncclComm_t)nccl_comm, false, rank, world, out, err);
} catch (...) {
  return abi_exception(err);
}

extern "C" void tgx_comm_destroy(tgx_comm *c) {
  if (!c) return;
  if (c->own_nccl && c->nccl && c->api && c->api->CommDestroy) (void)c->api->CommDestroy(c->nccl);
  if (c->h_a) (void)hipHostFree(c->h_a);
  if (c->h_b) (void)hipHostFree(c->h_b);
  delete c;
}

// ------------------------------------------------------------------------------------------------ the step
namespace {
enum : int64_t { kKindNone = 0, kKindBitmap = 1, kKindHash = 2 };

// what a rank tells the others about one DISTINCT task (all-gathered)
struct TaskFacts {
  int64_t lo, hi;   // value range of the Int64 keys it holds (lo > hi: none)
  int64_t kind;     // kKind*
  int64_t wide;     // 128-bit fingerprint set (Utf8 / tuple keys)
  int64_t rows;     // rows it has seen
  uint64_t spare_cap;  // bytes its spare bitmaps can take without an allocation (min of the two for a multiplicity set)
};

struct Header {
  uint64_t magic;
  uint64_t blob_len;   // payload bytes that follow the header in the blob round (0: it did not fit)
  uint64_t blob_need;  // capacity this rank needs
  // tgx_status of what this rank did on its own since the last header (resolving its key sets, exporting records,
  // packing its state): a rank that fails locally must not simply return -- its peers would wait in the next
  // collective for ever -- so it keeps taking part (with empty contributions) until the next header carries the
  // failure to everybody, and ALL ranks return an error from the same point
  uint64_t status;
  // bytes the rank's exchange buffers hold already: if every rank's buffers (and spare bitmaps, TaskFacts) are large
  // enough for what the facts call for, nobody can fail between here and the blob round's header and the step needs
  // no further status round; otherwise all ranks allocate, tell each other how that went, and only then exchange
  uint64_t xchg_cap;
  // the rank's fingerprint key (tgx_plan_set_fingerprint_key): string / tuple keys travel as fingerprints, which only
  // mean the same on every rank under ONE key
  uint64_t fp_key[2];
};
constexpr uint64_t kFactsMagic = 0x5447584641435453ull;  // "TGXFACTS"

uint64_t round_up(uint64_t x, uint64_t m) { return (x + m - 1) / m * m; }
}  // namespace

namespace {
// host-clock phases of the step, for a profiling state (tgx_profile_get: "xr_facts" -- local preparation + the facts
// round, i.e. mostly the wait for the shard's key passes --, "xr_exchange", "xr_pack" -- reading the state back, which
// waits for the scan --, "xr_gather", "xr_merge"): `total_ms` is wall time on this rank, `launches` the calls
struct HostPhase {
  tgx_state *st;
  const char *name;
  std::chrono::steady_clock::time_point t0;
  HostPhase(tgx_state *s, const char *n) : st(s), name(n), t0(std::chrono::steady_clock::now()) {}
  void end() {
    if (!st || !st->profiling) return;
    ProfileEntry &e = st->profile[name];
    e.total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    e.launches += 1;
    st = nullptr;
  }
  ~HostPhase() { end(); }
};
}  // namespace

extern "C" tgx_status tgx_allreduce(const tgx_plan *plan, tgx_state *st, tgx_comm *comm, tgx_error *err) try {
  if (!plan || !st || st->plan != plan) return fail(err, TGX_INVALID_ARGUMENT, "state does not belong to plan");
  if (!comm) return fail(err, TGX_INVALID_ARGUMENT, "comm is NULL");
  bind_thread();  // (a tokio worker / any thread: HIP's current device is per thread)
  TGX_TRY(coalesce_flush(st, err));  // batches tgx_update has only noted so far
  const int32_t W = comm->ops.world, R = comm->ops.rank;
  const size_t nd = plan->distinct.size();
  if (comm->ops.device_buffers) {
    TGX_TRY(need_device(err));
    TGX_TRY(state_init_device(st, err));
  }
  hipStream_t s = st->device_ready ? st->stream : nullptr;
  // Facts and key sets travel on a SECOND stream that waits only for the key columns' uniqueness passes
  // (tgx_update records `keys_ready` right after them and queues the scan of the other columns behind): the exchange
  // -- range / 8 bytes per rank and dense key column over xGMI, three collective latencies -- then runs while the
  // state's own stream is still scanning.  Every helper below works on `st->stream`, so the state borrows the second
  // stream for the two phases (StreamLoan) and the two are joined by an event before the states are packed.
  const bool has_spearman_tasks = plan->spearman && spearman_num_tasks(plan) > 0;
  const bool overlap = comm->ops.device_buffers && st->device_ready && st->keys_ready_recorded && nd > 0 &&
                       !has_spearman_tasks && getenv("TGX_NO_EXCHANGE_OVERLAP") == nullptr;
  st->exchange_expected = comm->ops.device_buffers != 0 && nd > 0;
  struct StreamLoan {
    tgx_state *st;
    hipStream_t own;
    bool on = false;
    ~StreamLoan() {
      if (on) st->stream = own;
    }
  } loan{st, st->stream};
  if (overlap) {
    if (!st->aux_stream) {
      HIP_TRY(stream_acquire(&st->aux_stream, true));  // (the most urgent priority; from the library's pool)
      HIP_TRY(hipEventCreateWithFlags(&st->aux_done, hipEventDisableTiming));
    }
    HIP_TRY(hipStreamWaitEvent(st->aux_stream, st->keys_ready, 0));
    st->stream = st->aux_stream;
    loan.on = true;
    s = st->aux_stream;
  }

  // ---- 0. SPEARMAN: ranks over the union of the ranks' pairs (a distributed sort, spearman_device.cpp) -----------
  // Rank-based states do not merge (the reference's neither: analyzers/advanced/correlation.rs:103-109), so their
  // results are computed here and carried past the reset + merge below by hand.
  std::vector<SpearmanResolved> spearman_results;
  const bool has_spearman = has_spearman_tasks;
  if (has_spearman) {
    SpearmanExchange X;
    X.rank = R;
    X.world = W;
    X.allgather_host = [&](const void *h_send, void *h_recv, size_t bytes) {
      return do_allgather_host(comm, st->device_ready ? st->stream : nullptr, h_send, h_recv, bytes, err);
    };
    X.alltoallv = [&](const void *d_send, const uint64_t *sc, void *d_recv, const uint64_t *rc, size_t elem) {
      return do_alltoallv(comm, st->device_ready ? st->stream : nullptr, d_send, sc, d_recv, rc, elem, err);
    };
    TGX_TRY(spearman_allreduce(st, X, &spearman_results, err));
    s = st->device_ready ? st->stream : nullptr;
    spearman_set_reducing(st, true);  // (the state's own blob does not carry these tasks)
  }
  struct ReducingGuard {
    tgx_state *st;
    bool on;
    ~ReducingGuard() {
      if (on) spearman_set_reducing(st, false);
    }
  } reducing_guard{st, has_spearman};

  HostPhase ph_facts(st, "xr_facts");
  // ---- 1. facts --------------------------------------------------------------------------------------------
  // A rank that fails ON ITS OWN between two collectives must not return: its peers would wait in the next one for
  // ever.  It notes the failure in `local`, keeps taking part with what it has, and the next status word the ranks
  // exchange anyway (facts header, the "ready" rounds below, the blob header) makes ALL ranks return an error from the
  // same point.  TGX_FAULT_INJECT="rank:site" makes that rank fail at site 1..7 (tests/test_gpu_distributed_sim.py).
  int fault_rank = -1, fault_site = 0;
  if (const char *fi = getenv("TGX_FAULT_INJECT")) (void)sscanf(fi, "%d:%d", &fault_rank, &fault_site);
  tgx_status local = TGX_OK;
  tgx_error local_err;
  memset(&local_err, 0, sizeof(local_err));
  auto note = [&](tgx_status sdone, const tgx_error &e) {
    if (local == TGX_OK && sdone != TGX_OK) {
      local = sdone;
      local_err = e;
    }
  };
  auto injected = [&](int site, tgx_error *e) -> tgx_status {
    if (fault_rank == R && fault_site == site) return fail(e, TGX_OUT_OF_MEMORY, "injected failure at site %d", site);
    return TGX_OK;
  };
  std::vector<ScanAcc> scan(plan->scan.size());
  auto local_prep = [&](tgx_error *err) -> tgx_status {
    TGX_TRY(injected(1, err));
    // (the small staging buffers of the record exchange: allocated here, where a failure still travels in the facts)
    if (st->device_ready || comm->ops.device_buffers) {
      HIP_TRY(comm->d_small_send.reserve((size_t)W * 16 + 16));
      HIP_TRY(comm->d_small_recv.reserve((size_t)W * 16 + 16));
    }
    if (st->device_ready && !scan.empty()) {
      // (queued in front of the resolve's own read-back, into pinned memory: the two come back with one wait)
      TGX_TRY(pinned_reserve(&comm->h_a, &comm->h_a_cap, scan.size() * sizeof(ScanAcc), err));
      HIP_TRY(hipMemcpyAsync(comm->h_a, st->d_scan_acc.p, scan.size() * sizeof(ScanAcc), hipMemcpyDeviceToHost, s));
    }
    TGX_TRY(distinct_resolve_all(st, err));  // keys outside a sampled bitmap range are brought in first
    if (st->device_ready) {
      HIP_TRY(hipStreamSynchronize(s));
      if (!scan.empty()) memcpy(scan.data(), comm->h_a, scan.size() * sizeof(ScanAcc));
    }
    return TGX_OK;
  };
  {
    tgx_error e;
    memset(&e, 0, sizeof(e));
    note(local_prep(&e), e);
  }
  if (local != TGX_OK || !st->device_ready) {
    for (auto &a : scan) {
      memset(&a, 0, sizeof(a));
      a.min_k = INT64_MAX;
      a.max_k = INT64_MIN;
    }
  }
  // every rank learns of a peer's local failure from a status word it was about to receive anyway
  auto report = [&](int32_t r, uint64_t status, const char *where) -> tgx_status {
    if (r == R && local != TGX_OK) {
      if (err) *err = local_err;
      return local;
    }
    return fail(err, (tgx_status)status, "tgx_allreduce: rank %d failed %s (%s); no rank went on", r, where,
                tgx_status_name((int32_t)status));
  };
  auto peers_ok = [&](const uint8_t *blocks, size_t stride, const char *where) -> tgx_status {
    for (int32_t r = 0; r < W; r++) {
      const Header *hr = (const Header *)(blocks + (size_t)r * stride);
      if (hr->status != 0) return report(r, hr->status, where);
    }
    return TGX_OK;
  };
  // one word per rank, all-gathered: "did what you just did on your own succeed?"
  auto status_round = [&](const char *where) -> tgx_status {
    std::vector<uint64_t> mine_w(1, (uint64_t)local), all_w((size_t)W, 0);
    TGX_TRY(do_allgather_host(comm, s, mine_w.data(), all_w.data(), sizeof(uint64_t), err));
    for (int32_t r = 0; r < W; r++)
      if (all_w[r] != 0) return report(r, all_w[r], where);
    return TGX_OK;
  };
  const size_t facts_bytes = sizeof(Header) + nd * sizeof(TaskFacts);
  std::vector<uint8_t> mine(facts_bytes), all(facts_bytes * (size_t)W);
  Header *hd = (Header *)mine.data();
  hd->magic = kFactsMagic ^ (uint64_t)nd;
  hd->blob_len = hd->blob_need = 0;
  hd->status = (uint64_t)local;
  hd->xchg_cap = st->device_ready ? std::min(comm->d_send.cap, comm->d_recv.cap) : 0;  // (0: "I will have to allocate")
  memcpy(hd->fp_key, plan->fp_key.k, 16);
  TaskFacts *tf = (TaskFacts *)(mine.data() + sizeof(Header));
  for (size_t k = 0; k < nd; k++) {
    const DistinctTask &task = plan->distinct[k];
    const DistinctState &ds = st->distinct[k];
    TaskFacts f;
    f.lo = INT64_MAX;
    f.hi = INT64_MIN;
    f.kind = kKindNone;
    f.wide = (ds.wide || !task.tuple.empty()) ? 1 : 0;
    f.rows = ds.total_rows;
    f.spare_cap = task.multiplicity ? std::min(ds.spare_seen.cap, ds.spare_twice.cap) : ds.spare_seen.cap;
    if (!ds.partitioned && st->device_ready) {
      if (ds.mode == DistinctMode::kBitmap) f.kind = kKindBitmap;
      if (ds.mode == DistinctMode::kHash) f.kind = kKindHash;
    }
    if (f.kind == kKindBitmap) {
      // tightest range known: the declared one, the scan's running MIN / MAX, else what the bitmap can represent
      f.lo = ds.base;
      f.hi = (int64_t)((uint64_t)ds.base + (ds.range - 1));
      if (ds.has_hint) {
        f.lo = ds.hint_lo;
        f.hi = ds.hint_hi;
      } else if (task.scan_slot >= 0) {
        ScanAcc a = scan[task.scan_slot];
        const ScanAcc &h = st->h_scan[task.scan_slot];
        if (h.non_null > 0) {
          a.min_k = std::min(a.min_k, h.min_k);
          a.max_k = std::max(a.max_k, h.max_k);
          a.non_null += h.non_null;
        }
        if (a.non_null > 0 && !a.is_float && a.min_k >= f.lo && a.max_k <= f.hi) {
          f.lo = a.min_k;
          f.hi = a.max_k;
        }
      }
    }
    tf[k] = f;
  }
  TGX_TRY(do_allgather_host(comm, s, mine.data(), all.data(), facts_bytes, err));
  for (int32_t r = 0; r < W; r++)
    if (((const Header *)(all.data() + (size_t)r * facts_bytes))->magic != hd->magic)
      return fail(err, TGX_INVALID_ARGUMENT, "rank %d runs a different plan (or the transport mixed up the blocks)", r);
  TGX_TRY(peers_ok(all.data(), facts_bytes, "while preparing its key sets"));
  auto facts_of = [&](int32_t r, size_t k) -> const TaskFacts & {
    return ((const TaskFacts *)(all.data() + (size_t)r * facts_bytes + sizeof(Header)))[k];
  };
  // string / tuple key sets: one fingerprint key on all ranks (every rank sees the same facts and returns the same)
  {
    bool any_wide = false;
    for (size_t k = 0; k < nd; k++)
      for (int32_t r = 0; r < W; r++) any_wide |= facts_of(r, k).wide != 0 && facts_of(r, k).rows > 0;
    if (any_wide)
      for (int32_t r = 0; r < W; r++)
        if (memcmp(((const Header *)(all.data() + (size_t)r * facts_bytes))->fp_key, hd->fp_key, 16) != 0)
          return fail(err, TGX_INVALID_ARGUMENT,
                      "rank %d holds another fingerprint key than rank %d: string / tuple keys cannot be united -- give "
                      "every rank's plan one key (tgx_plan_set_fingerprint_key) before its first state", r, R);
  }

  ph_facts.end();
  HostPhase ph_exchange(st, "xr_exchange");
  // ---- 2. exact DISTINCT: one exchange of key sets ----------------------------------------------------------------
  struct BitmapPart {
    size_t task;
    int64_t glo;
    uint64_t slice_words, col_words;
    bool mult;
  };
  std::vector<BitmapPart> parts;
  std::vector<size_t> by_records;
  uint64_t row_words = 0;
  for (size_t k = 0; k < nd; k++) {
    int64_t glo = INT64_MAX, ghi = INT64_MIN;
    bool any_set = false, all_bitmap = true, wide = false;
    uint64_t rows = 0;
    for (int32_t r = 0; r < W; r++) {
      const TaskFacts &f = facts_of(r, k);
      glo = std::min(glo, f.lo);
      ghi = std::max(ghi, f.hi);
      any_set |= f.kind != kKindNone;
      all_bitmap &= f.kind != kKindHash;
      wide |= f.wide != 0;
      rows += (uint64_t)f.rows;
    }
    if (!any_set) continue;  // nobody holds keys on a device (already partitioned, host-only or empty states)
    // a pure function of agreed values, so every rank takes the same branch: range bitmaps everywhere, and a global
    // range that is still dense (at most 16 bits per row of the whole table, below 2^34 values)
    bool use_bitmap = all_bitmap && !wide && glo <= ghi;
    if (use_bitmap) {
      const uint64_t width = (uint64_t)ghi - (uint64_t)glo;
      use_bitmap = width < (1ull << 34) && width / 16 <= std::max<uint64_t>(rows, 4096);
    }
    if (use_bitmap) {
      const uint64_t words = (((uint64_t)ghi - (uint64_t)glo) >> 5) + 1;
      BitmapPart p;
      p.task = k;
      p.glo = glo;
      p.slice_words = round_up((words + W - 1) / W, 4);
      p.col_words = row_words;
      p.mult = plan->distinct[k].multiplicity;
      row_words += p.slice_words * (p.mult ? 2 : 1);
      parts.push_back(p);
    } else {
      by_records.push_back(k);
    }
  }
  if ((!parts.empty() || !by_records.empty()) && !st->device_ready) {
    // (a rank that saw no batch still owns a slice / a share of the keys: it needs its device state now; whether a
    //  rank gets here is a function of the facts, and a failure travels in the next status word)
    tgx_error e;
    memset(&e, 0, sizeof(e));
    tgx_status sd = need_device(&e);
    if (sd == TGX_OK) sd = state_init_device(st, &e);
    if (sd == TGX_OK) {
      hipError_t he = comm->d_small_send.reserve((size_t)W * 16 + 16);
      if (he == hipSuccess) he = comm->d_small_recv.reserve((size_t)W * 16 + 16);
      if (he != hipSuccess) sd = fail(&e, TGX_OUT_OF_MEMORY, "exchange staging: %s", hipGetErrorString(he));
    }
    note(sd, e);
    if (st->device_ready) s = st->stream;
  }
  if (!parts.empty()) {
    const size_t total_bytes = (size_t)W * row_words * 4;
    // does ANY rank have to allocate for this exchange?  (a function of the facts: every rank answers alike)
    bool any_alloc = false;
    for (int32_t r = 0; r < W; r++) {
      any_alloc |= ((const Header *)(all.data() + (size_t)r * facts_bytes))->xchg_cap < total_bytes + 16;
      for (const BitmapPart &p : parts) any_alloc |= facts_of(r, p.task).spare_cap < p.slice_words * 4 + 16;
    }
    if (fault_site == 2 || fault_site == 3) any_alloc = true;  // (every rank reads the same environment)
    auto alloc = [&](tgx_error *err) -> tgx_status {
      TGX_TRY(injected(2, err));
      HIP_TRY(comm->d_send.reserve(total_bytes + 16));
      HIP_TRY(comm->d_recv.reserve(total_bytes + 16));
      for (const BitmapPart &p : parts) {
        DistinctState &ds = st->distinct[p.task];
        TGX_TRY(injected(3, err));
        HIP_TRY(ds.spare_seen.reserve(p.slice_words * 4 + 16));
        if (p.mult) HIP_TRY(ds.spare_twice.reserve(p.slice_words * 4 + 16));
      }
      return TGX_OK;
    };
    if (local == TGX_OK) {
      tgx_error e;
      memset(&e, 0, sizeof(e));
      note(alloc(&e), e);
    }
    if (any_alloc) TGX_TRY(status_round("while allocating the buffers of the key-set exchange"));
    // (from here to the blob round nothing of this branch can fail on its own but a queueing call of the runtime: it is
    //  noted and the blob header carries it)
    uint32_t *send = comm->d_send.as<uint32_t>();
    for (const BitmapPart &p : parts) {
      DistinctState &ds = st->distinct[p.task];
      const bool have = ds.mode == DistinctMode::kBitmap && !ds.partitioned;
      // a rank without keys sends zeros (src_words = 0)
      const long long delta = have ? (long long)((uint64_t)p.glo - (uint64_t)ds.base) : 0;
      const uint64_t src_words = have ? ds.bitmap_words : 0;
      launch_bitmap_rebase(have ? ds.seen.as<uint32_t>() : nullptr, src_words, delta, (uint32_t)W, p.slice_words,
                           row_words, p.col_words, send, s);
      if (p.mult)
        launch_bitmap_rebase(have ? ds.twice.as<uint32_t>() : nullptr, src_words, delta, (uint32_t)W, p.slice_words,
                             row_words, p.col_words + p.slice_words, send, s);
    }
    TGX_TRY(do_alltoall(comm, s, comm->d_send.p, comm->d_recv.p, (size_t)row_words * 4, err));
    const uint32_t *recv = comm->d_recv.as<uint32_t>();
    for (const BitmapPart &p : parts) {
      DistinctState &ds = st->distinct[p.task];
      // the owned slice: OR of what every rank sent for it; the key counters are recounted from it, the row counters
      // stay (the old bitmap becomes the spare of the next round, so a state that is reset and refilled every step
      // neither frees nor allocates)
      const hipError_t me = hipMemsetAsync(ds.counters.as<unsigned long long>() + kCntDistinct, 0,
                                           2 * sizeof(unsigned long long), s);
      if (me != hipSuccess) {
        tgx_error e;
        note(fail(&e, TGX_DEVICE_ERROR, "hipMemsetAsync failed: %s", hipGetErrorString(me)), e);
        continue;
      }
      launch_bitmap_adopt(recv + p.col_words, p.mult ? recv + p.col_words + p.slice_words : nullptr, (uint32_t)W,
                          p.slice_words, row_words, ds.spare_seen.as<uint32_t>(),
                          p.mult ? ds.spare_twice.as<uint32_t>() : nullptr, ds.counters.as<unsigned long long>(), s);
      std::swap(ds.seen, ds.spare_seen);
      std::swap(ds.twice, ds.spare_twice);
      ds.capacity = 0;
      ds.mode = DistinctMode::kBitmap;
      ds.col_type = TGX_INT64;
      ds.base = (int64_t)((uint64_t)p.glo + (uint64_t)R * p.slice_words * 32);
      ds.range = p.slice_words * 32;
      ds.bitmap_words = p.slice_words;
      ds.partitioned = true;
    }
  }
  for (size_t k : by_records) {
    DistinctState &ds = st->distinct[k];
    bool wide = false;
    for (int32_t r = 0; r < W; r++) wide |= facts_of(r, k).wide != 0;
    if (ds.mode == DistinctMode::kUndecided) ds.wide = wide;  // a rank that saw no rows still receives keys
    const void *recs = nullptr;
    std::vector<uint64_t> sc((size_t)W, 0), rc((size_t)W, 0);
    if (!ds.partitioned && local == TGX_OK) {
      tgx_error e;
      memset(&e, 0, sizeof(e));
      tgx_status se = injected(4, &e);
      if (se == TGX_OK) se = distinct_export_impl(st, k, (uint32_t)W, &recs, sc.data(), &e);
      note(se, e);
      if (local != TGX_OK) {  // keep taking part with nothing to send; the status beside the counts tells everybody
        recs = nullptr;
        std::fill(sc.begin(), sc.end(), 0);
      }
    }
    // counts first -- 16 bytes per peer: how many records, and whether this rank is still well -- then the records
    std::vector<uint64_t> cs((size_t)W * 2, 0), cr((size_t)W * 2, 0);
    for (int32_t r = 0; r < W; r++) {
      cs[2 * r] = sc[r];
      cs[2 * r + 1] = (uint64_t)local;
    }
    HIP_TRY(hipMemcpyAsync(comm->d_small_send.p, cs.data(), (size_t)W * 16, hipMemcpyHostToDevice, s));
    TGX_TRY(do_alltoall(comm, s, comm->d_small_send.p, comm->d_small_recv.p, 16, err));
    HIP_TRY(hipMemcpyAsync(cr.data(), comm->d_small_recv.p, (size_t)W * 16, hipMemcpyDeviceToHost, s));
    TGX_TRY(deadline_sync(comm, s, "the all-to-all of record counts", err));
    for (int32_t r = 0; r < W; r++) {
      rc[r] = cr[2 * r];
      if (cr[2 * r + 1] != 0) return report(r, cr[2 * r + 1], "while exporting its keys");
    }
    const size_t rec_bytes = wide ? sizeof(KeyRecord128) : sizeof(KeyRecord);
    uint64_t n_recv = 0;
    for (int32_t r = 0; r < W; r++) n_recv += rc[r];
    {  // the receive buffer: sized by what the peers announced, so its allocation is this rank's own business --
       // and the ranks tell each other how it went before anybody sends
      tgx_error e;
      memset(&e, 0, sizeof(e));
      tgx_status sa = injected(5, &e);
      if (sa == TGX_OK) {
        const hipError_t he = comm->d_recv.reserve(std::max<size_t>((size_t)n_recv * rec_bytes, 16));
        if (he != hipSuccess)
          sa = fail(&e, he == hipErrorOutOfMemory ? TGX_OUT_OF_MEMORY : TGX_DEVICE_ERROR,
                    "receive buffer of %llu key records: %s", (unsigned long long)n_recv, hipGetErrorString(he));
      }
      note(sa, e);
    }
    TGX_TRY(status_round("while allocating the receive buffer of its keys"));
    static const uint64_t nothing = 0;
    TGX_TRY(do_alltoallv(comm, s, recs ? recs : (const void *)&nothing, sc.data(), comm->d_recv.p, rc.data(), rec_bytes, err));
    // keep the row counters, replace the key set by the owned keys of all ranks (what fails from here on is noted: the
    // blob round's header carries it, and nothing below needs a peer)
    {
      tgx_error e;
      memset(&e, 0, sizeof(e));
      tgx_status si = injected(6, &e);
      if (si == TGX_OK) {
        const hipError_t me = hipMemsetAsync(ds.counters.as<unsigned long long>() + kCntDistinct, 0,
                                             3 * sizeof(unsigned long long), s);
        if (me != hipSuccess) si = fail(&e, TGX_DEVICE_ERROR, "hipMemsetAsync failed: %s", hipGetErrorString(me));
      }
      if (si == TGX_OK) {
        ds.capacity = 0;
        ds.rows_upper_bound = 0;
        ds.mode = DistinctMode::kHash;
        ds.wide = wide;
        si = distinct_import_records(st, k, comm->d_recv.p, n_recv, wide, &e);
      }
      note(si, e);
    }
    // (d_recv is reused by the next column's exchange: queued behind the import on the same stream -- the host does
    //  not wait here; a transport that takes HOST buffers has waited in do_alltoallv)
    ds.partitioned = true;
  }

  if (loan.on) {  // the state's own stream goes on only after the exchange; the host has not waited for either
    HIP_TRY(hipEventRecord(st->aux_done, st->aux_stream));
    st->stream = loan.own;
    loan.on = false;
    s = st->stream;
    HIP_TRY(hipStreamWaitEvent(st->stream, st->aux_done, 0));
  }
  ph_exchange.end();
  HostPhase ph_pack(st, "xr_pack");
  // ---- 3. the packed partial states: one all-gather, folded in rank order ----------------------------------------
  size_t len = 0;
  std::vector<uint8_t> blob;
  if (local == TGX_OK) {
    tgx_error e;
    memset(&e, 0, sizeof(e));
    tgx_status ss = injected(7, &e);
    // (packing reads the state back: it waits for the exchange queued above -- with a deadline, like every wait
    //  behind a collective)
    if (ss == TGX_OK && comm->ops.device_buffers && st->device_ready && (!parts.empty() || !by_records.empty()))
      ss = deadline_sync(comm, s, "the exchange of the key sets", &e);
    // (packing reads every accumulator back: once -- into a buffer of the capacity the ranks last agreed on -- and a
    //  second time only if the state has outgrown that)
    if (ss == TGX_OK) {
      blob.resize(std::max<size_t>(comm->blob_plan == plan ? comm->blob_cap : 0, 4096));
      ss = tgx_state_serialize(plan, st, blob.data(), blob.size(), &len, &e);
      if (ss == TGX_INVALID_ARGUMENT && len > blob.size()) {  // ("buffer too small": len says how much it takes)
        blob.resize(len);
        memset(&e, 0, sizeof(e));
        ss = tgx_state_serialize(plan, st, blob.data(), blob.size(), &len, &e);
      }
      if (ss == TGX_OK) blob.resize(len);
    }
    note(ss, e);
    if (local != TGX_OK) len = 0;
  }
  if (comm->blob_plan != plan) {
    comm->blob_plan = plan;
    comm->blob_cap = 0;
  }
  ph_pack.end();
  HostPhase ph_gather(st, "xr_gather");
  std::vector<uint8_t> sendbuf, recvbuf;
  for (int round = 0;; round++) {
    // every rank sends header + payload in a buffer of the agreed capacity; a rank whose payload has outgrown it says
    // so in the header, every rank sees that and all repeat the round with the larger capacity (the first round of a
    // plan only carries the sizes)
    const size_t cap = comm->blob_cap;
    const bool fits = cap > 0 && len <= cap;
    sendbuf.assign(sizeof(Header) + cap, 0);
    Header h;
    h.magic = kFactsMagic;
    h.blob_len = fits ? len : 0;
    h.blob_need = len;
    h.status = (uint64_t)local;
    memcpy(sendbuf.data(), &h, sizeof(h));
    if (fits) memcpy(sendbuf.data() + sizeof(Header), blob.data(), len);
    recvbuf.resize(sendbuf.size() * (size_t)W);
    TGX_TRY(do_allgather_host(comm, s, sendbuf.data(), recvbuf.data(), sendbuf.size(), err));
    size_t need = 0;
    bool all_fit = cap > 0;
    for (int32_t r = 0; r < W; r++) {
      const Header *hr = (const Header *)(recvbuf.data() + (size_t)r * sendbuf.size());
      if (hr->magic != kFactsMagic) return fail(err, TGX_INTERNAL, "state gather: bad header from rank %d", r);
      need = std::max<size_t>(need, hr->blob_need);
      all_fit &= hr->blob_need <= cap;
    }
    TGX_TRY(peers_ok(recvbuf.data(), sendbuf.size(), "while exchanging its key sets or packing its state"));
    if (all_fit) break;
    if (round >= 3) return fail(err, TGX_INTERNAL, "state gather: the ranks cannot agree on a capacity");
    comm->blob_cap = round_up(need + need / 2 + 64, 256);  // the same on every rank: all saw the same headers
  }
  ph_gather.end();
  HostPhase ph_merge(st, "xr_merge");
  // the local contribution travels in its own blob like everybody else's: empty this state (device buffers are kept)
  // and fold all W blobs in rank order -- identical arithmetic, hence bit-identical results, on every rank
  TGX_TRY(tgx_state_reset(plan, st, err));
  const size_t stride = sizeof(Header) + comm->blob_cap;
  for (int32_t r = 0; r < W; r++) {
    const uint8_t *p = recvbuf.data() + (size_t)r * stride;
    const Header *hr = (const Header *)p;
    tgx_state *part = nullptr;
    TGX_TRY(tgx_state_deserialize(plan, p + sizeof(Header), (size_t)hr->blob_len, &part, err));
    tgx_status ms = tgx_merge(plan, st, &part, 1, err);
    tgx_state_destroy(part);
    if (ms != TGX_OK) return ms;
  }
  if (has_spearman) spearman_install(st, spearman_results);
  return TGX_OK;
} catch (...) {
  return abi_exception(err);
}
