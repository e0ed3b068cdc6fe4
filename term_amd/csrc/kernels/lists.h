// lists.h -- device templates shared by the fingerprint lists of big Utf8 batches (distinct128.hip: 16-byte records,
// two 64-bit hashes of a value) and the key lists of big sparse Int64 / Float64 batches (distinct.hip: 8-byte records,
// the mixed key).  A record is partitioned by bits of its (first) word -- 8 bits per level -- into kFpFan^2 lists
// that are deduplicated one by one in LDS; see the comment in distinct128.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "distinct_types.h"

namespace tgx {

typedef unsigned long long KeyRec;  // 8-byte record: mix64 of the key (a bijection: equal records <=> equal keys)

__device__ __forceinline__ uint64_t rec_word(const ulonglong2 &r) { return r.x; }
__device__ __forceinline__ uint64_t rec_word(const KeyRec &r) { return r; }
__device__ __forceinline__ bool rec_equal(const ulonglong2 &a, const ulonglong2 &b) { return a.x == b.x && a.y == b.y; }
__device__ __forceinline__ bool rec_equal(const KeyRec &a, const KeyRec &b) { return a == b; }

template <class REC>
struct FpTileLdsT {
  REC stage[kFpTile];         // the tile's records grouped by list (level 1 of the strings: first the value bytes)
  uint32_t hist[kFpFan];      // records per list, then the placement cursors
  uint32_t lbase[kFpFan];     // where the list's run starts in `stage`
  uint32_t delta[kFpFan];     // position in the global list - position in `stage`
  uint32_t wsum[4], dropped;
};

template <class REC>
__device__ __forceinline__ void fp_tile_begin(FpTileLdsT<REC> &s) {
  s.hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) s.dropped = 0;
  __syncthreads();
}

// The tile's records are in registers (bit k of `present`: mine[k] is one) and counted per list in s.hist; a barrier
// has passed since, and nobody reads s.stage any more.  Groups them by list in LDS and appends every run to its
// list with ONE reservation per list.
template <class REC>
__device__ __forceinline__ void fp_tile_scatter(FpTileLdsT<REC> &s, const REC (&mine)[kFpTile / 256], uint32_t present,
                                                uint32_t out_list0, const FpLists &out, int shift,
                                                unsigned long long *counters) {
  constexpr int PER = kFpTile / 256;
  const uint32_t tid = threadIdx.x;
  const uint32_t h = s.hist[tid];
  // (the reservation's round trip runs under the scan and the regrouping: only the stores need its result)
  const uint32_t reserved = h ? atomicAdd(&out.offered[out_list0 + tid], h) : 0u;
  uint32_t incl = h;
#pragma unroll
  for (int dlt = 1; dlt < 64; dlt <<= 1) {
    const uint32_t up = __shfl_up(incl, dlt, 64);
    if ((tid & 63) >= (uint32_t)dlt) incl += up;
  }
  if ((tid & 63) == 63) s.wsum[tid >> 6] = incl;
  __syncthreads();  // wsum is there; everyone has read its count
  uint32_t excl = incl - h;
  for (uint32_t w = 0; w < (tid >> 6); w++) excl += s.wsum[w];
  s.lbase[tid] = excl;
  s.hist[tid] = excl;  // becomes the placement cursor
  const uint32_t total = s.wsum[0] + s.wsum[1] + s.wsum[2] + s.wsum[3];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; k++) {
    if (!((present >> k) & 1u)) continue;
    const uint32_t b = (uint32_t)(rec_word(mine[k]) >> shift) & (kFpFan - 1);
    s.stage[atomicAdd(&s.hist[b], 1u)] = mine[k];
  }
  s.delta[tid] = reserved - excl;
  __syncthreads();
  // runs out (16-byte records: 16 bytes per lane)
  bool lost = false;
  for (uint32_t p = tid; p < total; p += 256) {
    const REC r = s.stage[p];
    const uint32_t b = (uint32_t)(rec_word(r) >> shift) & (kFpFan - 1);
    const uint64_t at = (uint64_t)(uint32_t)(p + s.delta[b]);
    if (at < out.cap)
      ((REC *)out.recs)[(uint64_t)(out_list0 + b) * out.cap + at] = r;
    else
      lost = true;
  }
  if (lost) s.dropped = 1;  // (same value from every writer)
  __syncthreads();
  if (tid == 0 && s.dropped) atomicAdd(&counters[kCntOutOfRange], 1ull);
}

// level 2: a tile of one level-1 list -> the kFpFan lists of bits [48, 56) under it
template <class REC>
__global__ __launch_bounds__(256) void fp_partition_lists_kernel(FpLists in, uint32_t tiles_per_list, FpLists out,
                                                                  unsigned long long *counters) {
  constexpr int PER = kFpTile / 256;
  __shared__ FpTileLdsT<REC> s;
  const uint32_t tid = threadIdx.x;
  // Workgroups go round the 8 XCDs.  XCD x takes the level-1 lists (all kFpXcds of them) of 32 values of the first
  // byte, so everything that lands in one level-2 list comes out of ONE L2 (partial lines meet there before they
  // leave: the pass took 0.80 ms with the plain order, 0.62 ms with this one).
  constexpr uint32_t kPerXcd = kFpFan / kFpXcds;
  const uint32_t xcd = blockIdx.x % kFpXcds, j = blockIdx.x / kFpXcds;
  const uint32_t b1 = xcd * kPerXcd + j / (kFpXcds * tiles_per_list);
  const uint32_t in_list = ((j / tiles_per_list) % kFpXcds) * kFpFan + b1;
  const int64_t first = (int64_t)(j % tiles_per_list) * kFpTile;
  const uint64_t have = in.offered[in_list];
  int64_t count = (int64_t)(have < in.cap ? have : in.cap) - first;
  if (count <= 0) return;
  if (count > kFpTile) count = kFpTile;
  fp_tile_begin(s);
  const REC *src = (const REC *)in.recs + (uint64_t)in_list * in.cap + (uint64_t)first;
  REC mine[PER];
  uint32_t present = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int pos = k * 256 + (int)tid;
    if (pos < count) {
      mine[k] = src[pos];
      present |= 1u << k;
    }
  }
#pragma unroll
  for (int k = 0; k < PER; k++)
    if ((present >> k) & 1u) atomicAdd(&s.hist[(rec_word(mine[k]) >> 48) & (kFpFan - 1)], 1u);
  __syncthreads();
  fp_tile_scatter(s, mine, present, b1 * kFpFan, out, 48, counters);
}

// one workgroup per final list: distinct keys / keys seen twice of the list -> per_list[list]
// gridDim.x < n_lists: a workgroup takes lists blockIdx.x, blockIdx.x + gridDim.x, ... (the 128 KiB table leaves room
// for one workgroup a CU: one per CU that stays, instead of 65 536 that come and go, spares their set-up: 4.84 -> 4.52 ms
// per 1 G keys.  Tried and dropped there: 16-byte loads of two 8-byte records, 4.72 ms; two walks with half the table, 7.2)
// EQ: when are two records the same key?  PlainEq: when they are equal (a record IS its key: a mixed 8-byte key, a
// 128-bit fingerprint).  The exact string / tuple sets (distinct128.hip, ExactUtf8Eq / ExactTupleEq) carry a row number
// in their records and settle equal fingerprints by comparing the rows' bytes.
struct PlainEq {
  template <class REC>
  __device__ __forceinline__ bool operator()(const REC &a, const REC &b) const { return rec_equal(a, b); }
};
template <uint32_t SLOTS, uint32_t THREADS, class REC, class EQ = PlainEq>  // THREADS: 256 for the small table, 1024 for
                                                        // the big ones (one or two workgroups fit a CU then: the waves come from inside)
__global__ __launch_bounds__(THREADS) void fp_count_kernel(FpLists l, int want_mult, uint2 *per_list, uint32_t n_lists,
                                                            EQ same) {
  // a slot is (16 bits of the first word) << 16 | index of the record that owns it: ONE 32-bit compare-and-swap claims
  // it and names the owner (16 KiB of table: eight workgroups a CU).  Equal tags are settled by reading the owner's
  // record back from the list (it has just come through this CU's caches); tag, slot and list together fix 44 bits,
  // so that read is rare unless the values really are equal.
  static_assert(kFpListMax <= 0xFFFFu && SLOTS >= 4096 && (SLOTS & (SLOTS - 1)) == 0, "a record index fits 16 bits");
  __shared__ uint32_t slot[SLOTS];
  __shared__ uint32_t dupw[SLOTS / 32];
  __shared__ uint32_t s_new[THREADS / 64], s_dup[THREADS / 64];
  constexpr uint32_t kFree = 0xFFFFFFFFu;  // (no record has index 0xFFFF)
  const uint32_t tid = threadIdx.x;
  for (uint32_t list = blockIdx.x; list < n_lists; list += gridDim.x) {
  const uint32_t offered = l.offered[list];
  if (offered == 0 || offered > l.cap) {  // (an overflowed list was flagged by the kernel that filled it)
    if (tid == 0) per_list[list] = make_uint2(0, 0);
    continue;
  }
  const REC *recs = (const REC *)l.recs + (uint64_t)list * l.cap;
  constexpr int kAhead = 4;  // records a thread requests before it inserts the first
  REC r[kAhead];
#pragma unroll
  for (int j = 0; j < kAhead; j++)
    if (tid + THREADS * j < offered) r[j] = recs[tid + THREADS * j];
  for (uint32_t k = tid; k < SLOTS; k += THREADS) slot[k] = kFree;
  if (tid < SLOTS / 32) dupw[tid] = 0;
  __syncthreads();
  uint32_t n_new = 0, n_dup = 0;
  for (uint32_t i0 = tid; i0 < offered; i0 += THREADS * kAhead) {
    REC nx[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; j++) {
      const uint32_t i = i0 + THREADS * (kAhead + j);
      if (i < offered) nx[j] = recs[i];
    }
#pragma unroll
    for (int j = 0; j < kAhead; j++) {
      const uint32_t i = i0 + THREADS * j;
      if (i >= offered) break;
      const uint32_t tag = (uint32_t)rec_word(r[j]) & 0xFFFFu;
      const uint32_t mine = (tag << 16) | i;
      uint32_t hs = (uint32_t)(rec_word(r[j]) >> 32) & (SLOTS - 1);
      for (;;) {
        const uint32_t old = atomicCAS(&slot[hs], kFree, mine);
        if (old == kFree) {
          n_new++;
          break;
        }
        if ((old >> 16) == tag) {
          const REC o = recs[old & 0xFFFFu];
          if (same(o, r[j])) {
            if (want_mult) {
              const uint32_t bit = 1u << (hs & 31);
              const uint32_t prev = atomicOr(&dupw[hs >> 5], bit);
              n_dup += (prev & bit) ? 0u : 1u;
            }
            break;
          }
        }
        hs = (hs + 1) & (SLOTS - 1);
      }
    }
#pragma unroll
    for (int j = 0; j < kAhead; j++) r[j] = nx[j];
  }
#pragma unroll
  for (int dlt = 32; dlt >= 1; dlt >>= 1) {
    n_new += __shfl_down(n_new, dlt, 64);
    n_dup += __shfl_down(n_dup, dlt, 64);
  }
  if ((tid & 63) == 0) {
    s_new[tid >> 6] = n_new;
    s_dup[tid >> 6] = n_dup;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t a = 0, b = 0;
    for (uint32_t w = 0; w < THREADS / 64; w++) {
      a += s_new[w];
      b += s_dup[w];
    }
    per_list[list] = make_uint2(a, b);
  }
  __syncthreads();  // (the table and the wave sums are the next list's)
  }
}

// the CUs of the current device (a grid of workgroups that stay)
static inline unsigned fp_resident_grid() {
  static const unsigned cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      n = 0;
    return n > 0 ? (unsigned)n : 256u;
  }();
  return cus;
}

// the batch's counts into the task's counters (valid rows = records offered to the first level)
template <class REC>
__global__ __launch_bounds__(256) void fp_totals_kernel(const uint2 *per_list, uint32_t n_lists, const uint32_t *offered1,
                                                        unsigned long long *counters) {
  __shared__ unsigned long long s[3][4];
  unsigned long long a = 0, b = 0, v = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_lists; i += gridDim.x * 256) {
    const uint2 c = per_list[i];
    a += c.x;
    b += c.y;
  }
  if (blockIdx.x == 0 && offered1)  // (nullptr: the valid rows were counted elsewhere -- tuples)
    for (uint32_t i = threadIdx.x; i < (uint32_t)(kFpXcds * kFpFan); i += 256) v += offered1[i];
#pragma unroll
  for (int dlt = 32; dlt >= 1; dlt >>= 1) {
    a += __shfl_down(a, dlt, 64);
    b += __shfl_down(b, dlt, 64);
    v += __shfl_down(v, dlt, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s[0][threadIdx.x >> 6] = a;
    s[1][threadIdx.x >> 6] = b;
    s[2][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const unsigned long long t = s[threadIdx.x][0] + s[threadIdx.x][1] + s[threadIdx.x][2] + s[threadIdx.x][3];
    const int at = threadIdx.x == 0 ? kCntDistinct : threadIdx.x == 1 ? kCntTwice : kCntValidRows;
    if (t) atomicAdd(&counters[at], t);
  }
}

}  // namespace tgx
