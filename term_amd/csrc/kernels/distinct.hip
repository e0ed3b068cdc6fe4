// distinct.hip -- exact COUNT(DISTINCT col) and GROUP BY col multiplicity for gfx950.
//
//   COUNT(DISTINCT c)                                   TG/constraints/uniqueness.rs:612-617
//   SUM(CASE WHEN cnt = 1 ...) over GROUP BY c          TG/constraints/uniqueness.rs:671-681
//
// Keys are 64-bit patterns (Int64 values; Float64 by bit pattern, as DataFusion hashes floats).
// Two device-resident set representations, chosen per column by the host from the running
// MIN/MAX of the column (tgx_api.cpp):
//   * range bitmap: 1 bit per value of [base, base + range) (+1 "seen twice" bit when multiplicity
//     is wanted).  global atomicOr per row; the bit's previous value says whether the key is new.
//   * open-addressing hash set (linear probing, 64-bit atomicCAS claim), capacity 2^k >= 2 x keys;
//     the all-ones pattern doubles as EMPTY and is tracked by a side counter.
// Device-scope atomics execute at the memory side, so inserts from all 8 XCDs are coherent
// (MI355X_MICROARCH.md "Global float atomics" / SURVEY.md section 7 notes).  Counters are
// block-reduced first: one atomicAdd per block, not per row.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <stdlib.h>

#include "device_types.h"
#include "distinct_types.h"
#include "lists.h"

namespace tgx {

typedef const int64_t __attribute__((address_space(1))) *global_i64_ptr;
typedef const uint8_t __attribute__((address_space(1))) *global_u8_ptr;

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  // splitmix64 finaliser
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}

__device__ __forceinline__ void block_add2(unsigned long long a, unsigned long long b,
                                           unsigned long long *ga, unsigned long long *gb) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    a += __shfl_down(a, d, 64);
    b += __shfl_down(b, d, 64);
  }
  __shared__ unsigned long long sa[16], sb[16];
  const int wave = threadIdx.x >> 6;
  const int n_waves = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) {
    sa[wave] = a;
    sb[wave] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long ta = 0, tb = 0;
    for (int w = 0; w < n_waves; w++) {
      ta += sa[w];
      tb += sb[w];
    }
    if (ta) atomicAdd(ga, ta);
    if (tb) atomicAdd(gb, tb);
  }
}

// returns 1 if the key was new; *became_dup = 1 if this insert is the key's second sighting
__device__ __forceinline__ int hash_insert(const HashSetView &t, uint64_t key, int want_mult,
                                           int weight_two, int *became_dup) {
  uint64_t h = mix64(key) & t.mask;
  for (;;) {
    unsigned long long old =
        atomicCAS((unsigned long long *)&t.keys[h], (unsigned long long)kEmptyKey,
                  (unsigned long long)key);
    if (old == kEmptyKey) {
      if (want_mult && weight_two) {
        // another thread that found this key may already have set the bit: count it once
        const uint32_t bit = 1u << (h & 31);
        uint32_t prev = atomicOr(&t.dup[h >> 5], bit);
        *became_dup = (prev & bit) ? 0 : 1;
      }
      return 1;
    }
    if (old == key) {
      if (want_mult) {
        const uint32_t bit = 1u << (h & 31);
        // plain read first: most duplicates of a hot key find the bit already set
        if (!(__hip_atomic_load(&t.dup[h >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) {
          uint32_t prev = atomicOr(&t.dup[h >> 5], bit);
          *became_dup = (prev & bit) ? 0 : 1;
        }
      }
      return 0;
    }
    h = (h + 1) & t.mask;
  }
}

// One row per lane, grid-stride.  counters: [0] distinct, [1] keys seen at least twice,
// [2] rows whose key is the all-ones pattern (EMPTY stand-in), [3] non-null rows.
__global__ __launch_bounds__(256) void distinct_hash_kernel(DistinctColDesc d, HashSetView t,
                                                             unsigned long long *counters) {
  global_i64_ptr vals = (global_i64_ptr)(uintptr_t)((const int64_t *)d.values + d.offset);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  unsigned long long n_new = 0, n_dup = 0, n_empty = 0, n_valid = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.length; i += stride) {
    bool valid = true;
    if (vbits) {
      int64_t b = d.offset + i;
      valid = (vbits[b >> 3] >> (b & 7)) & 1;
    }
    if (!valid) continue;
    n_valid++;
    uint64_t key = (uint64_t)vals[i];
    if (key == kEmptyKey) {
      n_empty++;
      continue;
    }
    int became_dup = 0;
    n_new += hash_insert(t, key, d.want_multiplicity, 0, &became_dup);
    n_dup += became_dup;
  }
  block_add2(n_new, n_dup, &counters[0], &counters[1]);
  __syncthreads();
  block_add2(n_empty, n_valid, &counters[2], &counters[3]);
}

// Range bitmap: bit (key - base) of `seen`; `twice` marks keys seen again.
__global__ __launch_bounds__(256) void distinct_bitmap_kernel(DistinctColDesc d, BitmapView bm,
                                                               unsigned long long *counters) {
  global_i64_ptr vals = (global_i64_ptr)(uintptr_t)((const int64_t *)d.values + d.offset);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  unsigned long long n_new = 0, n_dup = 0, n_out = 0, n_valid = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.length; i += stride) {
    bool valid = true;
    if (vbits) {
      int64_t b = d.offset + i;
      valid = (vbits[b >> 3] >> (b & 7)) & 1;
    }
    if (!valid) continue;
    n_valid++;
    uint64_t rel = (uint64_t)vals[i] - (uint64_t)bm.base;  // wraps for keys below base
    if (rel >= bm.range) {
      n_out++;  // host guarantees this cannot happen; counted so a violation is detected
      continue;
    }
    const uint32_t bit = 1u << (rel & 31);
    uint32_t prev = atomicOr(&bm.seen[rel >> 5], bit);
    if (!(prev & bit)) {
      n_new++;
    } else if (d.want_multiplicity) {
      uint32_t p2 = atomicOr(&bm.twice[rel >> 5], bit);
      if (!(p2 & bit)) n_dup++;
    }
  }
  block_add2(n_new, n_dup, &counters[0], &counters[1]);
  __syncthreads();
  block_add2(n_out, n_valid, &counters[4], &counters[3]);
}

// ---------------------------------------------------------------------------------------------
// Range-partitioned bitmap population: the fast path for big batches of a dense-range Int64 column.
// A global atomicOr per row runs at ~27 G rows/s on MI355X (memory-side atomics); replaying bucketed
// keys against an LDS-resident slice of the bitmap is bounded by HBM traffic instead:
//   phase 1 reads 8 B/row and writes 4 B/row, phase 2 reads 4 B/row (+ the bitmap once).
//
// Phase 1.  One 1024-thread workgroup takes tiles of 32768 rows.  Each key gets its bucket and an
// in-tile rank from an LDS histogram (ds_add_rtn); a block scan turns the histogram into offsets and
// the tile is counting-sorted by bucket inside LDS.  The workgroup then reserves room in every bucket
// list it touches with ONE global atomicAdd per (tile, bucket) and each wave streams whole runs out,
// padded to 16 slots so that every global store is a full, 64-byte aligned chunk (scattered 4-byte
// stores ran this kernel 7x slower: 11.9 ms vs 1.6 ms without them at 1 G rows).  A list that is full
// (skewed data) spills to the global atomicOr path, so the result is exact for any distribution and
// only the speed depends on the spread.
// loads one tile's keys + validity into registers.  Validity bytes are requested BEFORE the keys so that
// turning them into the `ok` mask only waits for those (vmcnt retires in order) and the 16-byte key loads
// stay in flight.
template <int THREADS, int KPT, bool VALIDITY>
__device__ __forceinline__ void partition_load_tile(const PartitionParams &p, int64_t tile, bool wide,
                                                    int64_t (&key)[KPT], uint32_t &ok) {
  constexpr int kPartitionThreads = THREADS;  // (shadows the namespace constant inside this function)
  constexpr int kTile = kPartitionThreads * KPT;
  global_i64_ptr vals = (global_i64_ptr)(uintptr_t)((const int64_t *)p.values + p.offset);
  // VALIDITY = false: the column has no validity bitmap (its own instance: no byte loads, fewer live registers)
  global_u8_ptr vbits = VALIDITY ? (global_u8_ptr)(uintptr_t)p.validity : (global_u8_ptr) nullptr;
  const uint32_t tid = threadIdx.x;
  const int64_t row0 = tile * kTile;
  const bool full = row0 + kTile <= p.length;
  ok = 0;
  if (full && wide) {
    // lane holds rows row0 + (j/2)*2*T + 2*tid + (j&1): one global_load_dwordx4 per pair
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    typedef const i64x2 __attribute__((address_space(1))) *global_i64x2_ptr;
    global_i64x2_ptr pv = (global_i64x2_ptr)(vals + row0) + tid;
    const bool pair_bytes = vbits && (p.offset & 1) == 0;  // both rows of a pair share a validity byte
    // The validity bytes are requested and folded into `ok` BEFORE the keys are requested: holding 16 byte
    // registers next to the 64 key registers in flight spilled 29 registers per thread (3.6 GB of scratch traffic
    // per 1 G-row column); the price is one short, byte-sized round trip per tile ahead of the key loads.
    ok = (uint32_t)((1ull << KPT) - 1ull);
    if (pair_bytes) {
      uint8_t vb[KPT / 2];
#pragma unroll
      for (int j = 0; j < KPT / 2; j++) {
        const int64_t bit = p.offset + row0 + (int64_t)j * 2 * kPartitionThreads + 2 * tid;
        vb[j] = vbits[bit >> 3];
      }
      ok = 0;
#pragma unroll
      for (int j = 0; j < KPT / 2; j++) {
        const int64_t bit = p.offset + row0 + (int64_t)j * 2 * kPartitionThreads + 2 * tid;
        ok |= (uint32_t)((vb[j] >> (bit & 7)) & 3) << (2 * j);
      }
    } else if (vbits) {
      ok = 0;
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        const int64_t bit = p.offset + row0 + (int64_t)(j / 2) * 2 * kPartitionThreads + 2 * tid + (j & 1);
        ok |= (uint32_t)((vbits[bit >> 3] >> (bit & 7)) & 1) << j;
      }
    }
    if (VALIDITY) asm volatile("" : "+v"(ok));  // (the fold stays ahead of the key loads)
#pragma unroll
    for (int j = 0; j < KPT / 2; j++) {
      i64x2 v = pv[(int64_t)j * kPartitionThreads];
      key[2 * j] = v.x;
      key[2 * j + 1] = v.y;
    }
  } else {
    // ragged last tile / 8-byte aligned buffers: lane holds rows row0 + j*T + tid
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const int64_t i = row0 + (int64_t)j * kPartitionThreads + tid;
      const bool in = i < p.length;
      key[j] = vals[in ? i : p.length - 1];
      bool valid = in;
      if (in && vbits) {
        const int64_t bit = p.offset + i;
        valid = (vbits[bit >> 3] >> (bit & 7)) & 1;
      }
      ok |= (uint32_t)valid << j;
    }
  }
}

// THREADS x KPT keys per tile, up to MAXP = 2 * THREADS buckets, runs padded to PAD slots.  <1024, 32>: one
// workgroup per CU (152 KiB of LDS); <512, 32>: two per CU, so one loads while the other sorts.
// ---- the per-tile body shared by both partition kernels -------------------------------------------------
// THREADS x KPT keys per tile (32768 either way), up to MAXP buckets, runs padded to PAD slots.
//   pass 1   LDS histogram of the tile's keys per bucket
//   scan     exclusive prefix of the counts + ONE global atomicAdd per touched bucket reserving the run
//   pass 2   counting sort of the 20-bit sub-keys into LDS      (hook `mid` runs just before it)
//   stores   each wave streams whole runs out, 16 bytes per lane (hook `before_stores` runs just before)
// (the hooks are where a software-pipelined caller would request the next tile; unused today)
template <int THREADS, int KPT, int MAXP, int PAD, bool KEY16, bool PACK20, bool CLUSTERED, class MidFn, class StoreFn>
__device__ __forceinline__ void partition_process_tile(const PartitionParams &p, uint32_t *sorted, uint32_t *hist,
                                                       uint32_t *toff, uint32_t *gbase, uint32_t *wave_sums,
                                                       uint32_t *long_runs, const uint32_t (&rel)[KPT], uint64_t ok,
                                                       MidFn &&mid,
                                                       StoreFn &&before_stores) {
  constexpr uint32_t NW = THREADS / 64;       // waves per workgroup
  constexpr int BPT = MAXP / THREADS;         // buckets per thread in the scan
  constexpr int NS = MAXP / (NW * 64);        // run-metadata sets per lane in the store phase
  static_assert(MAXP % THREADS == 0 && MAXP % (NW * 64) == 0 && PAD % 4 == 0, "shape");
  // KEY16: buckets of <= 2^16 keys, so a list entry is 2 bytes: half the list traffic.  Runs are padded to 32 slots
  // (64 bytes) by REPEATING their last key (a set union is idempotent; not used with multiplicity), since no
  // 16-bit value is left over as a filler.
  // PACK20 (round 5): 20-bit entries, three to an 8-byte word: a 64-byte line holds 24 of them (2.67 B per key instead
  // of 4); runs are padded to 24 entries, again by repeating their last key
  constexpr uint32_t RPAD = PACK20 ? 24u : KEY16 ? (uint32_t)kRunPad2 : (uint32_t)PAD;
  auto pad_up = [](uint32_t h) -> uint32_t { return PACK20 ? (h + 23u) / 24u * 24u : (h + (RPAD - 1u)) & ~(RPAD - 1u); };
  // six consecutive entries of a run as two words of three.  Past the run's end: its last key again -- or, with
  // multiplicity (a second sighting of a key counts), the filler 0xFFFFF, which no sub-key equals there (two bitmap
  // slices share the LDS: sub_bits <= 19).  Two copies of the loop, picked by a wave-uniform branch per run: the
  // selects of the filler form cost the plain form 8 % of the pass when both shared one body
  const bool fill = p.want_multiplicity != 0;
  auto pack6 = [&](uint32_t o, uint32_t h, uint32_t i) -> uint4 {
    uint64_t w[2];
    if (!fill) {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const uint32_t i0 = i + 3u * q;
        const uint64_t a = sorted[o + (i0 < h ? i0 : h - 1)], b = sorted[o + (i0 + 1 < h ? i0 + 1 : h - 1)],
                       c = sorted[o + (i0 + 2 < h ? i0 + 2 : h - 1)];
        w[q] = a | (b << 20) | (c << 40);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        uint64_t e[3];
#pragma unroll
        for (uint32_t j = 0; j < 3; j++) {
          const uint32_t at = i + 3u * q + j;
          e[j] = at < h ? (uint64_t)sorted[o + at] : 0xFFFFFull;
        }
        w[q] = e[0] | (e[1] << 20) | (e[2] << 40);
      }
    }
    return make_uint4((uint32_t)w[0], (uint32_t)(w[0] >> 32), (uint32_t)w[1], (uint32_t)(w[1] >> 32));
  };
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63, wave = tid >> 6;
  const uint32_t sub_mask = (uint32_t)((1ull << p.sub_bits) - 1);
  for (uint32_t b = tid; b < (uint32_t)MAXP; b += THREADS) hist[b] = 0;
  // CLUSTERED: runs of kLongRun keys and more (a tile of keys in order is one or two runs) are streamed out by the
  // whole workgroup, not by the one wave that owns the bucket: long_runs[0 .. n) are their buckets, [kMaxLongRuns] = n
  constexpr uint32_t kLongRun = 1024, kMaxLongRuns = THREADS * KPT / kLongRun;
  if (CLUSTERED && tid == 0) long_runs[kMaxLongRuns] = 0;
  __syncthreads();  // hist is zero
  // ---- pass 1: count keys per bucket ----
  // Keys that arrive in order (ids that grow with the row number, timestamps) put the 128 consecutive rows a wave
  // holds for one j into ONE bucket: 64 LDS atomics on one address take their turns (a tile of sorted keys cost
  // 180 us instead of 24).  CLUSTERED (the batch looked like that to partition_init_kernel's probe): a wave whose
  // valid lanes agree on the bucket for EVERY j of the tile (`whole`, a wave-uniform fact: one scalar branch picks a
  // straight-line loop) sends one add of the lane count per j, and in pass 2 lines its lanes up behind one cursor
  // bump.  A wave that straddles a bucket boundary takes the plain form for this tile.  (A template parameter and a
  // copy of the tile loop, not a question per tile: next to the plain passes in one loop the extra state spilled
  // ~200 bytes per lane and cost shuffled keys 0.5 - 1.3 ms per 1 G-row column.)
  bool whole = false;
  if (CLUSTERED) {
    uint32_t agreed = 0;
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const bool okj = (ok >> j) & 1;
      const uint32_t b = rel[j] >> p.sub_bits;
      const uint64_t act = __ballot(okj);
      const uint32_t first = act ? (uint32_t)__builtin_ctzll(act) : 0u;
      const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)b, (int)first);
      agreed |= (__ballot(okj && b != b0) == 0 ? 1u : 0u) << j;
    }
    whole = agreed == (uint32_t)((1ull << KPT) - 1ull);
  }
  if (CLUSTERED && whole) {
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const uint64_t act = __ballot((ok >> j) & 1);
      const uint32_t first = act ? (uint32_t)__builtin_ctzll(act) : 64u;
      if (lane == first) atomicAdd(&hist[rel[j] >> p.sub_bits], (uint32_t)__builtin_popcountll(act));
    }
  } else {
#pragma unroll
    for (int j = 0; j < KPT; j++)
      if ((ok >> j) & 1) atomicAdd(&hist[rel[j] >> p.sub_bits], 1u);
  }
  __syncthreads();
  // ---- exclusive scan of the counts (BPT entries per thread) + one global reservation per touched bucket ----
  {
    uint32_t h[BPT], sum = 0;
#pragma unroll
    for (int k = 0; k < BPT; k++) {
      h[k] = hist[BPT * tid + k];
      sum += h[k];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t up = __shfl_up(incl, d, 64);
      if (lane >= (uint32_t)d) incl += up;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    uint32_t excl = incl - sum;
    for (uint32_t w = 0; w < wave; w++) excl += wave_sums[w];
#pragma unroll
    for (int k = 0; k < BPT; k++) {
      const uint32_t b = BPT * tid + k;
      toff[b] = excl;
      hist[b] = excl;  // becomes the placement cursor of pass 2
      uint32_t g = 0;
      if (h[k] && b - p.bucket0 >= p.n_lists) {
        g = 0xFFFFFFFFu;  // no list for this bucket: the run goes straight to the bitmap
      } else if (h[k]) {
        const unsigned long long padded = pad_up(h[k]);
        const unsigned long long at = atomicAdd(&p.cursors[b], padded);
        // cap < 2^32 (checked on the host); a run that does not fit spills as a whole
        if (at + padded > p.cap) {
          // every later reservation fails too, so the list is valid exactly up to the first failure
          atomicMin(&p.cursors[p.n_buckets + b], at);
          g = 0xFFFFFFFFu;
        } else {
          g = (uint32_t)at;
          if (CLUSTERED && h[k] >= kLongRun) long_runs[atomicAdd(&long_runs[kMaxLongRuns], 1u)] = b;
        }
      }
      gbase[b] = g;
      excl += h[k];
    }
    if (tid == THREADS - 1) toff[MAXP] = excl;
  }
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);  // the hooks stay where they are written
  mid();
  __builtin_amdgcn_sched_barrier(0);
  // ---- pass 2: counting sort into LDS ----
  if (CLUSTERED && whole) {
    // one bucket per j for the whole wave: one cursor bump, the lanes line up behind it
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const bool okj = (ok >> j) & 1;
      const uint64_t act = __ballot(okj);
      const uint32_t first = act ? (uint32_t)__builtin_ctzll(act) : 64u;
      uint32_t at = 0;
      if (lane == first) at = atomicAdd(&hist[rel[j] >> p.sub_bits], (uint32_t)__builtin_popcountll(act));
      at = (uint32_t)__builtin_amdgcn_readlane((int)at, (int)(first & 63u));
      const uint32_t before =
          __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
      if (okj) sorted[at + before] = rel[j] & sub_mask;
    }
  } else {
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      if (!((ok >> j) & 1)) continue;
      const uint32_t pos = atomicAdd(&hist[rel[j] >> p.sub_bits], 1u);
      sorted[pos] = rel[j] & sub_mask;
    }
  }
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  before_stores();
  __builtin_amdgcn_sched_barrier(0);
  // ---- each wave streams whole runs out, 16 bytes per lane.  The wave owns buckets wave, wave + NW, ...;
  // lane m keeps the (offset, count, global start) of the (m + 64 s)-th of them in registers, so a step needs
  // no LDS round trip for the bookkeeping.  Four groups of 16 lanes take four buckets per step (a group covers
  // 64 slots per pass; runs average kTile/P keys).  The cost of this phase is per store instruction, not per
  // byte: 4-byte-per-lane stores of the same runs took 2.3 ms instead of 1.0 ms.
  {
    uint32_t m_o[NS], m_h[NS], m_g[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; s2++) {
      const uint32_t b = (lane + 64 * s2) * NW + wave;
      const bool in = b < p.n_buckets;
      const uint32_t o = in ? toff[b] : 0;
      m_o[s2] = o;
      m_h[s2] = in ? toff[b + 1] - o : 0;
      m_g[s2] = in ? gbase[b] : 0;
    }
    const uint32_t n_meta = (p.n_buckets > wave) ? (p.n_buckets - wave + NW - 1) / NW : 0;
    // lanes per bucket and buckets per step: 16 x 4 with 4-byte entries, 4 x 16 with 2-byte entries (a lane always
    // stores 16 bytes; runs average kTile / P keys)
    constexpr uint32_t LPB = (KEY16 || PACK20) ? 4u : 16u, BPS = 64u / LPB, KPL = PACK20 ? 6u : KEY16 ? 8u : 4u;
    const uint32_t grp = lane / LPB, sub = lane % LPB;
#pragma unroll
    for (int s2 = 0; s2 < NS; s2++) {
      const uint32_t m_end = n_meta > 64u * s2 ? (n_meta - 64u * s2 < 64u ? n_meta - 64u * s2 : 64u) : 0;
      for (uint32_t m0 = 0; m0 < m_end; m0 += BPS) {
        const uint32_t m = m0 + grp, src = m & 63;
        const uint32_t o = __shfl(m_o[s2], src, 64), g = __shfl(m_g[s2], src, 64);
        uint32_t h = __shfl(m_h[s2], src, 64);
        if (m >= m_end) h = 0;
        if (CLUSTERED && h >= kLongRun && g != 0xFFFFFFFFu) h = 0;  // (everybody's job, below)
        if (h == 0) continue;
        const uint32_t b = (m + 64 * s2) * NW + wave;
        if (g != 0xFFFFFFFFu) {
          const uint32_t padded = pad_up(h);
          if (PACK20) {
            // cap and g are multiples of 24 entries: the run starts on a 64-byte line
            uint8_t *dst = (uint8_t *)p.lists + ((uint64_t)(b - p.bucket0) * p.cap + g) / 3 * 8;
            for (uint32_t i = KPL * sub; i < padded; i += KPL * LPB) *(uint4 *)(dst + (uint64_t)(i / 3) * 8) = pack6(o, h, i);
          } else if (KEY16) {
            // 16-byte aligned: cap and g are multiples of 32 two-byte slots
            uint16_t *dst = (uint16_t *)p.lists + (uint64_t)(b - p.bucket0) * p.cap + g;
            for (uint32_t i = KPL * sub; i < padded; i += KPL * LPB) {
              uint32_t k8[8];
#pragma unroll
              for (uint32_t j = 0; j < 8; j++) k8[j] = sorted[o + (i + j < h ? i + j : h - 1)];
              uint4 v;
              v.x = k8[0] | (k8[1] << 16);
              v.y = k8[2] | (k8[3] << 16);
              v.z = k8[4] | (k8[5] << 16);
              v.w = k8[6] | (k8[7] << 16);
              *(uint4 *)&dst[i] = v;
            }
          } else {
            uint32_t *dst = p.lists + (uint64_t)(b - p.bucket0) * p.cap + g;  // 16-byte aligned: cap, g multiples of PAD
            for (uint32_t i = KPL * sub; i < padded; i += KPL * LPB) {
              uint4 v;
              v.x = i < h ? sorted[o + i] : kListPad;
              v.y = i + 1 < h ? sorted[o + i + 1] : kListPad;
              v.z = i + 2 < h ? sorted[o + i + 2] : kListPad;
              v.w = i + 3 < h ? sorted[o + i + 3] : kListPad;
              *(uint4 *)&dst[i] = v;
            }
          }
        } else {
          for (uint32_t i = sub; i < h; i += LPB) {
            // spill: straight into the global bitmap
            const uint64_t r = ((uint64_t)b << p.sub_bits) | sorted[o + i];
            const uint32_t bit = 1u << (r & 31);
            const uint32_t prev = atomicOr(&p.seen[r >> 5], bit);
            if ((prev & bit) && p.want_multiplicity) atomicOr(&p.twice[r >> 5], bit);
          }
        }
      }
    }
  }
  if (CLUSTERED) {
    const uint32_t n_long = long_runs[kMaxLongRuns];
    for (uint32_t q = 0; q < n_long; q++) {
      const uint32_t b = long_runs[q];
      const uint32_t o = toff[b], h = toff[b + 1] - o, g = gbase[b];
      const uint32_t padded = pad_up(h);
      if (PACK20) {
        uint8_t *dst = (uint8_t *)p.lists + ((uint64_t)(b - p.bucket0) * p.cap + g) / 3 * 8;
        for (uint32_t i = 6u * tid; i < padded; i += 6u * THREADS) *(uint4 *)(dst + (uint64_t)(i / 3) * 8) = pack6(o, h, i);
      } else if (KEY16) {
        uint16_t *dst = (uint16_t *)p.lists + (uint64_t)(b - p.bucket0) * p.cap + g;
        for (uint32_t i = 8u * tid; i < padded; i += 8u * THREADS) {
          uint32_t k8[8];
#pragma unroll
          for (uint32_t j = 0; j < 8; j++) k8[j] = sorted[o + (i + j < h ? i + j : h - 1)];
          uint4 v;
          v.x = k8[0] | (k8[1] << 16);
          v.y = k8[2] | (k8[3] << 16);
          v.z = k8[4] | (k8[5] << 16);
          v.w = k8[6] | (k8[7] << 16);
          *(uint4 *)&dst[i] = v;
        }
      } else {
        uint32_t *dst = p.lists + (uint64_t)(b - p.bucket0) * p.cap + g;
        for (uint32_t i = 4u * tid; i < padded; i += 4u * THREADS) {
          uint4 v;
          v.x = i < h ? sorted[o + i] : kListPad;
          v.y = i + 1 < h ? sorted[o + i + 1] : kListPad;
          v.z = i + 2 < h ? sorted[o + i + 2] : kListPad;
          v.w = i + 3 < h ? sorted[o + i + 3] : kListPad;
          *(uint4 *)&dst[i] = v;
        }
      }
    }
  }
  __syncthreads();
}

// Keys in order replayed against a slice: G neighbouring lanes hold the 32 keys of ONE bitmap word, and G atomics on one
// LDS address take their turns.  The lanes of such a group merge their bits (a butterfly over the group; merging only
// ever adds bits of the same word, so it is harmless when the group does not agree) and, when the whole group names
// the same word, only its first lane sends the OR.  Returns whether this lane still has to send its own.
template <int G>
__device__ __forceinline__ bool merge_word_group(uint32_t cw, uint32_t &cb) {
  const uint32_t lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < G; d <<= 1) {
    const uint32_t ow = __shfl_xor(cw, d, 64), ob = __shfl_xor(cb, d, 64);
    if (ow == cw) cb |= ob;
  }
  const uint32_t lead = lane & ~(uint32_t)(G - 1);
  const uint64_t agree = __ballot(cw == (uint32_t)__shfl(cw, (int)lead, 64));
  const bool whole = ((agree >> lead) & ((1ull << G) - 1)) == ((1ull << G) - 1);
  return !whole || lane == lead;
}

// key - base of one tile: in-range keys fit 31 bits (n_buckets << sub_bits <= 2^31).  "In range" is key - base < range,
// the test every other kernel of the key set applies (distinct_bitmap_kernel, distinct_outlier_kernel, the exports): NOT
// "inside the last slice" -- a key between the range's end and the slice's would be in the bitmap for this pass and
// an outlier for the repair, i.e. counted twice (groups_once came out short).  Keys outside the range (a
// sampled range that missed them, a later batch of keys that grow, a caller-supplied hint that does not hold) are never
// inserted but counted, so that the repair (distinct_resolve) or tgx_finalize knows; `outm` gets their positions.
template <int KPT>
__device__ __forceinline__ void partition_relative(const PartitionParams &p, const int64_t (&key)[KPT],
                                                   uint32_t (&rel)[KPT], uint32_t &outm) {
  outm = 0;
#pragma unroll
  for (int j = 0; j < KPT; j++) {
    const uint64_t r = (uint64_t)key[j] - (uint64_t)p.base;
    outm |= (r >= p.range ? 1u : 0u) << j;
    rel[j] = (uint32_t)r;  // the 64-bit keys die here
    asm volatile("" : "+v"(rel[j]));  // (keeps the compiler from re-deriving rel from the keys later)
  }
  __builtin_amdgcn_sched_barrier(0);
}

// The same for a batch of keys in no particular order (the plain passes): the few keys outside a sampled range add
// their share of the aggregates through global atomics where they are met.
template <int KPT, bool STATS>
__device__ __forceinline__ void partition_relative_plain(const PartitionParams &p, const int64_t (&key)[KPT],
                                                         uint32_t (&rel)[KPT], uint64_t &ok, unsigned long long &n_out) {
#pragma unroll
  for (int j = 0; j < KPT; j++) {
    const uint64_t r = (uint64_t)key[j] - (uint64_t)p.base;
    if (((ok >> j) & 1) && r >= p.range) {
      ok &= ~(1ull << j);
      n_out++;
      if (STATS) {
        const long long k = (long long)key[j];
        atomicMin(&p.outliers->mn, k);
        atomicMax(&p.outliers->mx, k);
        atomicAdd(&p.outliers->lo32_sum, (unsigned long long)((uint64_t)k & 0xFFFFFFFFull));
        atomicAdd((unsigned long long *)&p.outliers->hi32_sum, (unsigned long long)(k >> 32));
        atomicAdd(&p.outliers->count, 1ull);
      }
    }
    rel[j] = (uint32_t)r;  // the 64-bit keys die here
    asm volatile("" : "+v"(rel[j]));  // (keeps the compiler from re-deriving rel from the keys later)
  }
  __builtin_amdgcn_sched_barrier(0);
}

// The outliers' share of the column's aggregates (STATS), collected in LDS and handed on once by partition_kernel: a
// batch that lies outside the range altogether would otherwise queue five device-wide atomics per key on the same
// five addresses (3.9 ms per 4 Mi keys).  Their keys are read AGAIN here (from L2: the tile has just been loaded) --
// keeping the 64-bit keys until now, or reducing inside partition_relative, spills the tile's registers on every tile
// for the sake of a case that is rare.  MIN / MAX only bother the LDS when they would change it.
template <int THREADS, int KPT>
__device__ __forceinline__ void partition_outlier_stats(const PartitionParams &p, int64_t tile, bool wide, uint32_t outm,
                                                     OutlierStats *lds_out) {
  constexpr int kTile = THREADS * KPT;
  global_i64_ptr vals = (global_i64_ptr)(uintptr_t)((const int64_t *)p.values + p.offset);
  const int64_t row0 = tile * kTile;
  const bool paired = row0 + kTile <= p.length && wide;  // (the layout partition_load_tile chose)
  unsigned long long cnt = 0;
  for (int j = 0; j < KPT; j++) {
    if (!((outm >> j) & 1)) continue;
    const int64_t i = paired ? row0 + (int64_t)(j / 2) * 2 * THREADS + 2 * threadIdx.x + (j & 1)
                             : row0 + (int64_t)j * THREADS + threadIdx.x;
    const long long k = vals[i];
    if (k < *(volatile long long *)&lds_out->mn) atomicMin(&lds_out->mn, k);
    if (k > *(volatile long long *)&lds_out->mx) atomicMax(&lds_out->mx, k);
    atomicAdd(&lds_out->lo32_sum, (unsigned long long)((uint64_t)k & 0xFFFFFFFFull));
    atomicAdd((unsigned long long *)&lds_out->hi32_sum, (unsigned long long)(k >> 32));
    cnt++;
  }
  atomicAdd(&lds_out->count, cnt);
}

// 1024 threads x 32 keys, one workgroup per CU (152 KiB of LDS); any alignment, ragged last tile.
// FORM: the probe's verdict (partition_init_kernel) picks one of two copies of the tile loop for the whole launch; they
// are two KERNELS, launched one behind the other, the one whose form it is not leaving at once: in one kernel they
// shared a register allocation and the form for keys in order paid for it (78 scratch loads per tile and thread)
template <int THREADS, int KPT, int MAXP, int PAD, bool VALIDITY, bool KEY16, bool STATS, bool FORM_CLUSTERED, bool PACK20 = false>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void partition_kernel(
    PartitionParams p, unsigned long long *counters) {
  if ((p.force_form ? p.force_form == 2 : __builtin_amdgcn_readfirstlane((int)p.cursors[2 * p.n_buckets]) != 0) != FORM_CLUSTERED)
    return;
  constexpr int kTile = THREADS * KPT;
  __shared__ uint32_t sorted[kTile];     // the tile, grouped by bucket
  __shared__ uint32_t hist[MAXP];        // pass 1: keys per bucket; pass 2: placement cursors
  __shared__ uint32_t toff[MAXP + 1];    // exclusive prefix of the counts (toff[P] = tile total)
  __shared__ uint32_t gbase[MAXP];       // start of the run in the bucket's global list
  __shared__ uint32_t wave_sums[16];
  __shared__ uint32_t long_runs[kTile / 1024 + 1];
  __shared__ uint32_t t_lo, t_hi;  // (CLUSTERED) the span of the tile's keys
  const bool wide = (((uintptr_t)p.values + (uintptr_t)p.offset * 8) & 15) == 0;  // 16-byte loads legal
  unsigned long long n_valid = 0, n_out = 0;
  const int64_t n_tiles = (p.length + kTile - 1) / kTile;
  uint32_t rel[KPT];
  // STATS: the column's COUNT / MIN / MAX / SUM over the keys inside the range, taken from the 32-bit offsets (the
  // 64-bit keys are gone by then: no register is held across the tile for them) -- per tile a wave reduction and
  // four LDS atomics
  __shared__ uint32_t st_min, st_max;
  __shared__ unsigned long long st_sum, st_cnt;
  __shared__ OutlierStats st_out;  // (STATS) the keys outside the range: their aggregates go out once, at the end
  if (STATS && threadIdx.x == 0) {
    st_min = 0xFFFFFFFFu;
    st_max = 0;
    st_sum = 0;
    st_cnt = 0;
    st_out.mn = INT64_MAX;
    st_out.mx = INT64_MIN;
    st_out.lo32_sum = 0;
    st_out.hi32_sum = 0;
    st_out.count = 0;
  }
  if (STATS) __syncthreads();
  auto tile_loop = [&](auto clustered_tag) __attribute__((always_inline)) {
  constexpr bool CLUSTERED = decltype(clustered_tag)::value;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint64_t ok;
    {
      int64_t key[KPT];
      uint32_t ok32 = 0;
      partition_load_tile<THREADS, KPT, VALIDITY>(p, tile, wide, key, ok32);
      ok = ok32;
      n_valid += __builtin_popcountll(ok);  // every non-NULL row, whether its key lies inside the range or not
      if (CLUSTERED) {
        // (keys in order: a later batch of a column that grows lies outside the range as a whole)
        uint32_t outm;
        partition_relative<KPT>(p, key, rel, outm);
        outm &= ok32;
        ok = ok32 & ~outm;
        if (__ballot(outm != 0)) {
          n_out += __builtin_popcount(outm);
          if (STATS && outm) partition_outlier_stats<THREADS, KPT>(p, tile, wide, outm, &st_out);
        }
      } else {
        partition_relative_plain<KPT, STATS>(p, key, rel, ok, n_out);
      }
    }
    if (STATS) {
      uint32_t tmin = 0xFFFFFFFFu, tmax = 0;
      unsigned long long tsum = 0;
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        const bool in = (ok >> j) & 1;
        tmin = (in && rel[j] < tmin) ? rel[j] : tmin;
        tmax = (in && rel[j] > tmax) ? rel[j] : tmax;
        tsum += in ? rel[j] : 0u;
      }
      unsigned long long tcnt = __builtin_popcountll(ok);
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t omin = __shfl_down(tmin, d, 64), omax = __shfl_down(tmax, d, 64);
        tmin = omin < tmin ? omin : tmin;
        tmax = omax > tmax ? omax : tmax;
        tsum += __shfl_down(tsum, d, 64);
        tcnt += __shfl_down(tcnt, d, 64);
      }
      if ((threadIdx.x & 63) == 0 && tcnt) {
        atomicMin(&st_min, tmin);
        atomicMax(&st_max, tmax);
        atomicAdd(&st_sum, tsum);
        atomicAdd(&st_cnt, tcnt);
      }
    }
    if (CLUSTERED && !p.want_multiplicity) {
      // A tile of keys in order covers a short stretch of the range: when its keys span fewer bits than `sorted` has
      // (2^20: 32 768 consecutive ids span 2^15) the tile is OR-ed into a bitmap of that stretch in LDS and the
      // stretch's non-zero words into the global bitmap -- one device-wide atomic per 32 keys (ids in steps of one),
      // no list written, nothing to replay.  (Not with multiplicity: the second sighting of a key is not seen here.)
      if (threadIdx.x == 0) {
        t_lo = 0xFFFFFFFFu;
        t_hi = 0;
      }
      __syncthreads();
      uint32_t lo = 0xFFFFFFFFu, hi = 0;
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        const bool in = (ok >> j) & 1;
        lo = (in && rel[j] < lo) ? rel[j] : lo;
        hi = (in && rel[j] > hi) ? rel[j] : hi;
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t olo = __shfl_xor(lo, d, 64), ohi = __shfl_xor(hi, d, 64);
        lo = olo < lo ? olo : lo;
        hi = ohi > hi ? ohi : hi;
      }
      if ((threadIdx.x & 63) == 0 && lo <= hi) {
        atomicMin(&t_lo, lo);
        atomicMax(&t_hi, hi);
      }
      __syncthreads();
      const uint32_t tlo = t_lo, thi = t_hi;
      if (tlo > thi) continue;  // (no valid key inside the range in this tile)
      const uint32_t w0 = tlo >> 5, nw = (thi >> 5) - w0 + 1;
      if (nw <= (uint32_t)kTile) {
        for (uint32_t w = threadIdx.x; w < nw; w += THREADS) sorted[w] = 0;
        __syncthreads();
        // a lane's keys 2 j and 2 j + 1 are neighbouring rows; 16 lanes share a word when ids go in steps of one
#pragma unroll
        for (int j = 0; j < KPT; j += 2) {
          uint32_t cw = 0xFFFFFFFFu, cb = 0;
          if ((ok >> j) & 1) {
            cw = (rel[j] >> 5) - w0;
            cb = 1u << (rel[j] & 31);
          }
          if ((ok >> (j + 1)) & 1) {
            const uint32_t w1 = (rel[j + 1] >> 5) - w0, b1 = 1u << (rel[j + 1] & 31);
            if (w1 == cw) {
              cb |= b1;
            } else {
              if (cb) atomicOr(&sorted[cw], cb);
              cw = w1;
              cb = b1;
            }
          }
          const bool send = merge_word_group<16>(cw, cb);
          if (send && cb) atomicOr(&sorted[cw], cb);
        }
        __syncthreads();
        for (uint32_t w = threadIdx.x; w < nw; w += THREADS) {
          const uint32_t v = sorted[w];
          if (v) atomicOr(&p.seen[w0 + w], v);
        }
        __syncthreads();
        continue;
      }
    }
    partition_process_tile<THREADS, KPT, MAXP, PAD, KEY16, PACK20, CLUSTERED>(p, sorted, hist, toff, gbase, wave_sums, long_runs, rel, ok,
                                                                      [] {}, [] {});
  }
  };
  tile_loop(std::integral_constant<bool, FORM_CLUSTERED>{});
  if (STATS) {
    __syncthreads();
    if (threadIdx.x == 0) {
      ScanPartial out;
      out.min_k = INT64_MAX;
      out.max_k = INT64_MIN;
      out.sum_lo = 0;
      out.sum_hi = 0;
      out.non_null = (int64_t)st_cnt;
      out.sum = out.comp = out.s1 = out.s2 = 0.0;
      if (st_cnt) {
        out.min_k = (int64_t)((uint64_t)p.base + st_min);
        out.max_k = (int64_t)((uint64_t)p.base + st_max);
        const __int128 sum = (__int128)st_sum + (__int128)st_cnt * (__int128)p.base;
        out.sum_lo = (uint64_t)sum;
        out.sum_hi = (int64_t)(sum >> 64);
      }
      p.stats[blockIdx.x] = out;
      if (st_out.count) {
        atomicMin(&p.outliers->mn, st_out.mn);
        atomicMax(&p.outliers->mx, st_out.mx);
        atomicAdd(&p.outliers->lo32_sum, st_out.lo32_sum);
        atomicAdd((unsigned long long *)&p.outliers->hi32_sum, (unsigned long long)st_out.hi32_sum);
        atomicAdd(&p.outliers->count, st_out.count);
      }
    }
  }
  block_add2(n_valid, n_out, &counters[kCntValidRows], &counters[kCntOutOfRange]);
}

// ---- the partition pass with the runs' remainders CARRIED OVER (round 6) ------------------------------------------------
// partition_kernel pads every (tile, bucket) run to a whole 64-byte line: half a line per run on average, whatever the
// entry size -- 1.26 of the 3.76 bytes a key of the headline's id column leaves in its list (954 buckets: runs of 34
// entries), written once and read once by the replay.  Here a workgroup -- which walks many tiles -- keeps what a run
// leaves beyond whole lines in LDS, per bucket and PACKED AS THE LIST IS (three 20-bit entries to an 8-byte word, at
// most 23 entries = 8 words a bucket), and the bucket's next run starts with it: only whole lines are ever reserved and
// written, every entry is a real key, and the padding shrinks to the one line per bucket a workgroup flushes when it
// is through.  The carry takes 64 KiB for up to 1024 buckets, so a tile is 16 384 keys (its runs may be short now).
// For the 20-bit lists without multiplicity and keys that do not arrive in order (those take partition_kernel's
// CLUSTERED form, which this kernel leaves to it by returning at once -- the same probe flag).
// MEASURED AND LEFT OFF (TGX_PARTITION_CARRY=1 turns it on): 1 G shuffled ids 3.75 ms against partition_kernel's 2.46,
// the 10^8-value column 3.10 against 2.40; the replay gains what the lists lost (0.72 -> 0.46 ms, 0.55 -> 0.42) but the
// pass loses three times as much: its cost is per tile and per LDS operation, not per byte written -- half the tile
// doubles the barriers, scans and reservations per key, and building the carry is ~16 LDS operations per thread and
// tile next to the 48 of the two counting passes.  With the tile kept at 32 768 keys the carry of 954 buckets does not
// fit beside it (128 + 16 + 61 KiB).
template <int THREADS, int KPT, bool VALIDITY, bool STATS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void partition_carry_kernel(
    PartitionParams p, unsigned long long *counters) {
  if (p.force_form ? p.force_form == 2 : __builtin_amdgcn_readfirstlane((int)p.cursors[2 * p.n_buckets]) != 0) return;
  constexpr int kTile = THREADS * KPT;
  constexpr int MAXP = THREADS;          // one bucket per thread in the scan, the carry update and the flush
  constexpr uint32_t NW = THREADS / 64;  // waves
  constexpr uint32_t kNoList = 0xFFFFFFFFu;
  static_assert(MAXP == (int)(NW * 64), "a lane of every wave names one bucket in the store phase");
  __shared__ uint32_t sorted[kTile];
  __shared__ uint32_t hist[MAXP], toff[MAXP + 1], gbase[MAXP], ccount[MAXP];
  __shared__ unsigned long long carryw[MAXP * 8];
  __shared__ uint32_t wave_sums[16];
  __shared__ uint32_t st_min, st_max;
  __shared__ unsigned long long st_sum, st_cnt;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool wide = (((uintptr_t)p.values + (uintptr_t)p.offset * 8) & 15) == 0;
  const uint32_t sub_mask = (uint32_t)((1ull << p.sub_bits) - 1);
  unsigned long long n_valid = 0, n_out = 0;
  const int64_t n_tiles = (p.length + kTile - 1) / kTile;
  ccount[tid] = 0;
  if (STATS && tid == 0) {
    st_min = 0xFFFFFFFFu;
    st_max = 0;
    st_sum = 0;
    st_cnt = 0;
  }
  // entry i of what bucket b holds: its carry (c entries) first, then the tile's run (from sorted[o])
  auto carry_get = [&](uint32_t b, uint32_t i) -> uint32_t {
    return (uint32_t)(carryw[b * 8 + i / 3] >> (20u * (i % 3u))) & 0xFFFFFu;
  };
  auto entry = [&](uint32_t b, uint32_t c, uint32_t o, uint32_t i) -> uint32_t {
    return i < c ? carry_get(b, i) : sorted[o + i - c];
  };
  auto word_of = [&](uint32_t b, uint32_t c, uint32_t o, uint32_t first) -> unsigned long long {  // entries first .. first + 2
    if (first + 3 <= c) return carryw[b * 8 + first / 3];  // (packed alike)
    return (unsigned long long)entry(b, c, o, first) | ((unsigned long long)entry(b, c, o, first + 1) << 20) |
           ((unsigned long long)entry(b, c, o, first + 2) << 40);
  };
  auto spill = [&](uint32_t b, uint32_t sub_key) {  // straight into the global bitmap (no list, or a full one)
    const uint64_t r = ((uint64_t)b << p.sub_bits) | sub_key;
    atomicOr(&p.seen[r >> 5], 1u << (r & 31));
  };
  uint32_t rel[KPT];
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint64_t ok;
    {
      int64_t key[KPT];
      uint32_t ok32 = 0;
      partition_load_tile<THREADS, KPT, VALIDITY>(p, tile, wide, key, ok32);
      ok = ok32;
      n_valid += __builtin_popcountll(ok);
      partition_relative_plain<KPT, STATS>(p, key, rel, ok, n_out);
    }
    if (STATS) {
      uint32_t tmin = 0xFFFFFFFFu, tmax = 0;
      unsigned long long tsum = 0;
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        const bool in = (ok >> j) & 1;
        tmin = (in && rel[j] < tmin) ? rel[j] : tmin;
        tmax = (in && rel[j] > tmax) ? rel[j] : tmax;
        tsum += in ? rel[j] : 0u;
      }
      unsigned long long tcnt = __builtin_popcountll(ok);
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t omin = __shfl_down(tmin, d, 64), omax = __shfl_down(tmax, d, 64);
        tmin = omin < tmin ? omin : tmin;
        tmax = omax > tmax ? omax : tmax;
        tsum += __shfl_down(tsum, d, 64);
        tcnt += __shfl_down(tcnt, d, 64);
      }
      if (lane == 0 && tcnt) {
        atomicMin(&st_min, tmin);
        atomicMax(&st_max, tmax);
        atomicAdd(&st_sum, tsum);
        atomicAdd(&st_cnt, tcnt);
      }
    }
    hist[tid] = 0;
    __syncthreads();  // hist is zero; the carry of the tile before is in place
#pragma unroll
    for (int j = 0; j < KPT; j++)
      if ((ok >> j) & 1) atomicAdd(&hist[rel[j] >> p.sub_bits], 1u);
    __syncthreads();
    // ---- scan (a bucket per thread) + one reservation per bucket that has whole lines to give ----
    {
      const uint32_t h = hist[tid];
      uint32_t incl = h;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += up;
      }
      if (lane == 63) wave_sums[wave] = incl;
      __syncthreads();
      uint32_t excl = incl - h;
      for (uint32_t w = 0; w < wave; w++) excl += wave_sums[w];
      toff[tid] = excl;
      hist[tid] = excl;  // the placement cursor of pass 2
      if (tid == THREADS - 1) toff[MAXP] = excl + h;
      uint32_t g = 0;
      if (tid - p.bucket0 >= p.n_lists) {
        g = kNoList;  // (the batch was not expected to touch this bucket: its keys go straight to the bitmap)
      } else {
        const uint32_t emit = (ccount[tid] + h) / 24u * 24u;
        if (emit) {
          const unsigned long long at = atomicAdd(&p.cursors[tid], (unsigned long long)emit);
          if (at + emit > p.cap) {
            atomicMin(&p.cursors[p.n_buckets + tid], at);  // (the list is valid up to the first failure)
            g = kNoList;
          } else {
            g = (uint32_t)at;
          }
        }
      }
      gbase[tid] = g;
    }
    __syncthreads();
    // ---- pass 2: counting sort into LDS ----
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      if (!((ok >> j) & 1)) continue;
      const uint32_t pos = atomicAdd(&hist[rel[j] >> p.sub_bits], 1u);
      sorted[pos] = rel[j] & sub_mask;
    }
    __syncthreads();
    // ---- whole lines out: wave w owns buckets w, w + NW, ...; four lanes a bucket, 16 bytes (six entries) a lane ----
    {
      const uint32_t mb = lane * NW + wave;  // the bucket this lane keeps the bookkeeping of
      const uint32_t m_o = toff[mb], m_h = toff[mb + 1] - m_o, m_g = gbase[mb], m_c = ccount[mb];
      const uint32_t grp = lane >> 2, sub = lane & 3;
      const uint32_t n_meta = p.n_buckets > wave ? (p.n_buckets - wave + NW - 1) / NW : 0;
      for (uint32_t m0 = 0; m0 < n_meta; m0 += 16) {
        const uint32_t m = m0 + grp, src = m & 63;
        const uint32_t o = __shfl(m_o, src, 64), g = __shfl(m_g, src, 64), c = __shfl(m_c, src, 64);
        uint32_t h = __shfl(m_h, src, 64);
        if (m >= n_meta) continue;
        const uint32_t b = m * NW + wave;
        if (b - p.bucket0 >= p.n_lists) {  // no list: the tile's keys of the bucket (it never carries)
          for (uint32_t i = sub; i < h; i += 4) spill(b, sorted[o + i]);
          continue;
        }
        const uint32_t emit = (c + h) / 24u * 24u;
        if (emit == 0) continue;
        if (g != kNoList) {
          // cap and g are multiples of 24 entries: a line starts on a 64-byte boundary
          uint8_t *dst = (uint8_t *)p.lists + ((uint64_t)(b - p.bucket0) * p.cap + g) / 3 * 8;
          for (uint32_t i = 6u * sub; i < emit; i += 24u) {
            const unsigned long long w0 = word_of(b, c, o, i), w1 = word_of(b, c, o, i + 3);
            *(uint4 *)(dst + (uint64_t)(i / 3) * 8) = make_uint4((uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32));
          }
        } else {
          for (uint32_t i = sub; i < emit; i += 4) spill(b, entry(b, c, o, i));
        }
      }
    }
    __syncthreads();  // everybody has read the old carry
    // ---- the new carry (a bucket per thread): what the lines left over, or the old carry with the run appended ----
    if (tid - p.bucket0 < p.n_lists) {
      const uint32_t o = toff[tid], h = toff[tid + 1] - o, c = ccount[tid];
      const uint32_t total = c + h, emit = total / 24u * 24u;
      if (emit) {
        const uint32_t left = total - emit, from = o + emit - c;  // (emit >= 24 > c: the carry is all gone)
        for (uint32_t q = 0; 3u * q < left; q++) {
          unsigned long long w = 0;
#pragma unroll
          for (uint32_t j = 0; j < 3; j++)
            if (3u * q + j < left) w |= (unsigned long long)sorted[from + 3u * q + j] << (20u * j);
          carryw[tid * 8 + q] = w;
        }
        ccount[tid] = left;
      } else if (h) {
        for (uint32_t q = c / 3u; 3u * q < total; q++) {
          unsigned long long w = 3u * q < c ? carryw[tid * 8 + q] : 0ull;
#pragma unroll
          for (uint32_t j = 0; j < 3; j++) {
            const uint32_t pos = 3u * q + j;
            if (pos >= c && pos < total) w |= (unsigned long long)sorted[o + pos - c] << (20u * j);
          }
          carryw[tid * 8 + q] = w;
        }
        ccount[tid] = total;
      }
    }
    // (the next tile's first barrier -- or the one below -- comes before anybody reads carry or `sorted` again)
  }
  __syncthreads();
  // ---- what is left of every bucket: one line, padded by its last entry (a set union is idempotent) ----
  if (tid - p.bucket0 < p.n_lists) {
    const uint32_t c = ccount[tid];
    if (c) {
      const unsigned long long at = atomicAdd(&p.cursors[tid], 24ull);
      if (at + 24 > p.cap) {
        atomicMin(&p.cursors[p.n_buckets + tid], at);
        for (uint32_t i = 0; i < c; i++) spill(tid, carry_get(tid, i));
      } else {
        unsigned long long *dst = (unsigned long long *)((uint8_t *)p.lists + ((uint64_t)(tid - p.bucket0) * p.cap + at) / 3 * 8);
        const unsigned long long last = carry_get(tid, c - 1);
        for (uint32_t q = 0; q < 8; q++) {
          unsigned long long w = 0;
#pragma unroll
          for (uint32_t j = 0; j < 3; j++) {
            const uint32_t pos = 3u * q + j;
            w |= (pos < c ? (unsigned long long)carry_get(tid, pos) : last) << (20u * j);
          }
          dst[q] = w;
        }
      }
    }
  }
  if (STATS) {
    __syncthreads();
    if (tid == 0) {
      ScanPartial out;
      out.min_k = INT64_MAX;
      out.max_k = INT64_MIN;
      out.sum_lo = 0;
      out.sum_hi = 0;
      out.non_null = (int64_t)st_cnt;
      out.sum = out.comp = out.s1 = out.s2 = 0.0;
      if (st_cnt) {
        out.min_k = (int64_t)((uint64_t)p.base + st_min);
        out.max_k = (int64_t)((uint64_t)p.base + st_max);
        const __int128 sum = (__int128)st_sum + (__int128)st_cnt * (__int128)p.base;
        out.sum_lo = (uint64_t)sum;
        out.sum_hi = (int64_t)(sum >> 64);
      }
      p.stats[blockIdx.x] = out;
    }
  }
  block_add2(n_valid, n_out, &counters[kCntValidRows], &counters[kCntOutOfRange]);
}

// (the outliers' share of the aggregates -- OutlierStats -- is folded by scan_reduce_kernel, kernels/scan.hip)

// A strided sample of the column (at most 2^16 rows, evenly spread): where its keys lie, before anything has read
// it.  The DISTINCT pass lays its range bitmap out from this estimate (with slack) and takes the column's range
// aggregates along; keys that fall outside after all are counted and repaired later (tgx_api.cpp, distinct_resolve).
__global__ __launch_bounds__(256) void distinct_sample_kernel(DistinctColDesc d, DistinctSample *out) {
  global_i64_ptr vals = (global_i64_ptr)(uintptr_t)((const int64_t *)d.values + d.offset);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  // d.pad != 0: every row (the exact MIN / MAX of a coalesced flush of DEVICE windows), not a sample
  const int64_t want = (d.pad || d.length < 65536) ? d.length : 65536;
  const int64_t step = d.length / want;
  int64_t mn = INT64_MAX, mx = INT64_MIN;
  unsigned long long cnt = 0;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < want; k += (int64_t)gridDim.x * 256) {
    const int64_t i = k * step + ((k * 40503) & 15 ? 0 : (step > 1 ? step / 2 : 0));  // (not a pure stride)
    bool valid = true;
    if (vbits) {
      const int64_t b = d.offset + i;
      valid = (vbits[b >> 3] >> (b & 7)) & 1;
    }
    if (!valid) continue;
    const int64_t v = vals[i];
    mn = v < mn ? v : mn;
    mx = v > mx ? v : mx;
    cnt++;
  }
#pragma unroll
  for (int dlt = 32; dlt >= 1; dlt >>= 1) {
    const int64_t omn = __shfl_down(mn, dlt, 64), omx = __shfl_down(mx, dlt, 64);
    mn = omn < mn ? omn : mn;
    mx = omx > mx ? omx : mx;
    cnt += __shfl_down(cnt, dlt, 64);
  }
  if ((threadIdx.x & 63) == 0 && cnt) {
    atomicMin((long long *)&out->min_v, (long long)mn);
    atomicMax((long long *)&out->max_v, (long long)mx);
    atomicAdd(&out->count, cnt);
  }
}

// The repair of a speculated range: the batch's keys OUTSIDE [base, base + range) -- never inserted into the bitmap
// -- go into the hash set (the bitmap's keys have been moved there first).  Row counters are not touched.
__global__ __launch_bounds__(256) void distinct_outlier_kernel(DistinctColDesc d, int64_t base, uint64_t range,
                                                                HashSetView t, unsigned long long *counters) {
  global_i64_ptr vals = (global_i64_ptr)(uintptr_t)((const int64_t *)d.values + d.offset);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  unsigned long long n_new = 0, n_dup = 0, n_empty = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.length; i += stride) {
    bool valid = true;
    if (vbits) {
      int64_t b = d.offset + i;
      valid = (vbits[b >> 3] >> (b & 7)) & 1;
    }
    if (!valid) continue;
    const uint64_t key = (uint64_t)vals[i];
    if (key - (uint64_t)base < range) continue;  // the bitmap had it
    if (key == kEmptyKey) {
      n_empty++;
      continue;
    }
    int became_dup = 0;
    n_new += hash_insert(t, key, d.want_multiplicity, 0, &became_dup);
    n_dup += became_dup;
  }
  block_add2(n_new, n_dup, &counters[0], &counters[1]);
  __syncthreads();
  block_add2(n_empty, 0ull, &counters[2], &counters[kCntSpare]);
}

// Phase 2.  Workgroup b owns slice b of the bitmap: load it into LDS (it already holds the keys of
// earlier batches and this batch's spills), replay list b with LDS atomics, store it back, and add the
// slice's popcounts to the totals (counters[kCntDistinct] / [kCntTwice] are zeroed before the launch).
template <int LDS_WORDS, bool KEY16 = false, bool PACK20 = false>
__global__ __launch_bounds__(kPartitionThreads) void bucket_apply_kernel(PartitionParams p,
                                                                         unsigned long long *counters) {
  // static LDS: gfx950 lets one workgroup declare up to 160 KiB statically (dynamic LDS is capped lower)
  __shared__ __attribute__((aligned(16))) uint32_t lds[LDS_WORDS];
  const uint32_t tid = threadIdx.x;
  const uint32_t b = blockIdx.x;
  const uint32_t slice_words = (uint32_t)((1ull << p.sub_bits) >> 5);
  uint32_t *l_seen = lds;
  uint32_t *l_twice = lds + slice_words;  // only with multiplicity (host guarantees 2*slice_words fit)
  uint32_t *g_seen = p.seen + (uint64_t)b * slice_words;
  uint32_t *g_twice = p.want_multiplicity ? p.twice + (uint64_t)b * slice_words : nullptr;
  for (uint32_t w = tid * 4; w < slice_words; w += kPartitionThreads * 4) {
    *(uint4 *)&l_seen[w] = *(const uint4 *)&g_seen[w];
    if (g_twice) *(uint4 *)&l_twice[w] = *(const uint4 *)&g_twice[w];
  }
  __syncthreads();
  unsigned long long cnt = p.cursors[b];
  const unsigned long long limit = p.cursors[p.n_buckets + b];  // start of the first run that spilled
  if (cnt > limit) cnt = limit;
  const bool clustered = p.cursors[2 * p.n_buckets] != 0;  // (partition_init_kernel's probe)
  const uint32_t li = b - p.bucket0;  // (a bucket without a list has cnt == 0: its runs spilled)
  if (li >= p.n_lists) cnt = 0;
  if (PACK20) {
    // 20-bit entries, three to an 8-byte word, six per 16-byte load; cnt is a multiple of 24 and runs are padded with
    // repeats of real keys -- every entry counts -- or, with multiplicity, with a filler (the keys-in-order form of the pass takes this plain loop as well: a
    // column that reaches the lists in order has wide steps, and its tiles' words rarely repeat)
    const uint8_t *list = (const uint8_t *)p.lists + (uint64_t)li * p.cap / 3 * 8;
    constexpr uint64_t kStepP = (uint64_t)kPartitionThreads * 6;
    for (uint64_t w0 = (uint64_t)(tid & ~63u) * 6; w0 < cnt; w0 += 4 * kStepP) {
      const uint64_t i0 = w0 + (uint64_t)(tid & 63u) * 6;
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      u32x4 k4[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint64_t i = i0 + q * kStepP;
        k4[q] = u32x4{0, 0, 0, 0};
        if (i < cnt) k4[q] = __builtin_nontemporal_load((const u32x4 *)(list + i / 3 * 8));
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (i0 + q * kStepP >= cnt) continue;
        const uint64_t w[2] = {(uint64_t)k4[q].x | ((uint64_t)k4[q].y << 32), (uint64_t)k4[q].z | ((uint64_t)k4[q].w << 32)};
#pragma unroll
        for (int u = 0; u < 6; u++) {
          const uint32_t k = (uint32_t)(w[u / 3] >> (20 * (u % 3))) & 0xFFFFFu;
          const uint32_t bit = 1u << (k & 31);
          if (g_twice) {  // (multiplicity: runs are padded with the filler, never with a key)
            if (k == 0xFFFFFu) continue;
            const uint32_t prev = atomicOr(&l_seen[k >> 5], bit);
            if (prev & bit) atomicOr(&l_twice[k >> 5], bit);
          } else {
            atomicOr(&l_seen[k >> 5], bit);
          }
        }
      }
    }
  } else if (KEY16) {
    // 2-byte entries, eight per 16-byte load; runs are padded with repeats of real keys, so every entry counts
    const uint16_t *list16 = (const uint16_t *)p.lists + (uint64_t)li * p.cap;
    constexpr uint64_t kStep16 = (uint64_t)kPartitionThreads * 8;
    // (a wave's lanes leave the loop together -- the loop bound is the wave's first entry -- so that the merge of
    //  neighbouring lanes below always finds the whole wave)
    for (uint64_t w0 = (uint64_t)(tid & ~63u) * 8; w0 < cnt; w0 += 4 * kStep16) {
      const uint64_t i0 = w0 + (uint64_t)(tid & 63u) * 8;
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      u32x4 k4[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint64_t i = i0 + q * kStep16;
        k4[q] = u32x4{0, 0, 0, 0};
        if (i < cnt) k4[q] = __builtin_nontemporal_load((const u32x4 *)&list16[i]);
      }
      if (clustered) {
        // (the probe saw keys in order: a lane's eight keys mostly share a bitmap word, four lanes share it too)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const bool in = i0 + q * kStep16 < cnt;
          const uint32_t w4[4] = {k4[q].x, k4[q].y, k4[q].z, k4[q].w};
          uint32_t cw = 0xFFFFFFFFu, cb = 0;
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const uint32_t k = (u & 1) ? w4[u / 2] >> 16 : w4[u / 2] & 0xFFFFu;
            if (!in) continue;
            if ((k >> 5) != cw) {
              if (cb) atomicOr(&l_seen[cw], cb);
              cw = k >> 5;
              cb = 0;
            }
            cb |= 1u << (k & 31);
          }
          const bool send = merge_word_group<4>(cw, cb);
          if (send && cb) atomicOr(&l_seen[cw], cb);
        }
      } else {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (i0 + q * kStep16 >= cnt) continue;
        const uint32_t w4[4] = {k4[q].x, k4[q].y, k4[q].z, k4[q].w};
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t ka = w4[u] & 0xFFFFu, kb = w4[u] >> 16;
          atomicOr(&l_seen[ka >> 5], 1u << (ka & 31));
          atomicOr(&l_seen[kb >> 5], 1u << (kb & 31));
        }
      }
      }
    }
  } else {
  const uint32_t *list = p.lists + (uint64_t)li * p.cap;
  // lists are made of 16-slot aligned runs, so cnt is a multiple of 4; kListPad slots are filler
  // four 16-byte loads in flight per lane before the first LDS atomic (requesting the NEXT four before the atomics
  // of the current ones -- the pipeline that pays in dict.hip / kll.hip -- measured 1.01 ms instead of 0.93 here)
  constexpr uint64_t kStep = (uint64_t)kPartitionThreads * 4;
  for (uint64_t w0 = (uint64_t)(tid & ~63u) * 4; w0 < cnt; w0 += 4 * kStep) {  // (whole waves, as above)
    const uint64_t i0 = w0 + (uint64_t)(tid & 63u) * 4;
    uint4 k4[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint64_t i = i0 + q * kStep;
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      u32x4 v = {kListPad, kListPad, kListPad, kListPad};
      if (i < cnt) v = __builtin_nontemporal_load((const u32x4 *)&list[i]);
      k4[q] = make_uint4(v.x, v.y, v.z, v.w);
    }
    if (clustered && !g_twice) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t ks[4] = {k4[q].x, k4[q].y, k4[q].z, k4[q].w};
        uint32_t cw = 0xFFFFFFFFu, cb = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (ks[u] == kListPad) continue;
          if ((ks[u] >> 5) != cw) {
            if (cb) atomicOr(&l_seen[cw], cb);
            cw = ks[u] >> 5;
            cb = 0;
          }
          cb |= 1u << (ks[u] & 31);
        }
        const bool send = merge_word_group<8>(cw, cb);
        if (send && cb) atomicOr(&l_seen[cw], cb);
      }
      continue;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t ks[4] = {k4[q].x, k4[q].y, k4[q].z, k4[q].w};
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (ks[u] == kListPad) continue;
        const uint32_t bit = 1u << (ks[u] & 31);
        if (g_twice) {
          const uint32_t prev = atomicOr(&l_seen[ks[u] >> 5], bit);
          if (prev & bit) atomicOr(&l_twice[ks[u] >> 5], bit);
        } else {
          atomicOr(&l_seen[ks[u] >> 5], bit);
        }
      }
    }
  }
  }
  __syncthreads();
  unsigned long long n_seen = 0, n_twice = 0;
  for (uint32_t w = tid * 4; w < slice_words; w += kPartitionThreads * 4) {
    const uint4 s4 = *(const uint4 *)&l_seen[w];
    *(uint4 *)&g_seen[w] = s4;
    n_seen += __builtin_popcount(s4.x) + __builtin_popcount(s4.y) + __builtin_popcount(s4.z) +
              __builtin_popcount(s4.w);
    if (g_twice) {
      const uint4 t4 = *(const uint4 *)&l_twice[w];
      *(uint4 *)&g_twice[w] = t4;
      n_twice += __builtin_popcount(t4.x) + __builtin_popcount(t4.y) + __builtin_popcount(t4.z) +
                 __builtin_popcount(t4.w);
    }
  }
  block_add2(n_seen, n_twice, &counters[kCntDistinct], &counters[kCntTwice]);
}

// Re-inserts every key of `src` into `dst` (growth, merge, bitmap -> hash conversion).
__global__ __launch_bounds__(256) void hash_rehash_kernel(HashSetView src, HashSetView dst,
                                                           int want_mult,
                                                           unsigned long long *counters) {
  unsigned long long n_new = 0, n_dup = 0;
  const uint64_t cap = src.mask + 1;
  for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cap;
       s += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t key = src.keys[s];
    if (key == kEmptyKey) continue;
    int two = want_mult ? ((src.dup[s >> 5] >> (s & 31)) & 1) : 0;
    int became_dup = 0;
    int is_new = hash_insert(dst, key, want_mult, two, &became_dup);
    if (!is_new && want_mult && two) {
      // key present on both sides and already a duplicate on the source side: hash_insert has
      // set the bit (or found it set); became_dup tells whether it was newly set
    }
    n_new += is_new;
    n_dup += became_dup;
  }
  block_add2(n_new, n_dup, &counters[0], &counters[1]);
}

__global__ __launch_bounds__(256) void bitmap_to_hash_kernel(BitmapView bm, HashSetView dst,
                                                              int want_mult,
                                                              unsigned long long *counters) {
  unsigned long long n_new = 0, n_dup = 0, n_empty = 0;
  const uint64_t words = (bm.range + 31) >> 5;
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words;
       w += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t seen = bm.seen[w];
    uint32_t twice = want_mult ? bm.twice[w] : 0;
    while (seen) {
      int b = __builtin_ctz(seen);
      seen &= seen - 1;
      uint64_t key = (uint64_t)bm.base + (w << 5) + (uint64_t)b;
      if (key == kEmptyKey) {  // cannot live in the table: goes to the side counter
        n_empty += ((twice >> b) & 1) ? 2 : 1;
        continue;
      }
      int became_dup = 0;
      n_new += hash_insert(dst, key, want_mult, (twice >> b) & 1, &became_dup);
      n_dup += became_dup;
    }
  }
  block_add2(n_new, n_dup, &counters[0], &counters[1]);
  __syncthreads();
  block_add2(n_empty, 0ull, &counters[2], &counters[5]);
}

// Inserts 16-byte {key, count} records (count saturates at 2) -- merge / cross-rank import.
__global__ __launch_bounds__(256) void hash_import_kernel(const KeyRecord *recs, uint64_t n,
                                                           HashSetView dst, int want_mult,
                                                           unsigned long long *counters) {
  unsigned long long n_new = 0, n_dup = 0, n_empty = 0, n_empty_dup = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    KeyRecord r = recs[i];
    if (r.key == kEmptyKey) {
      n_empty += r.count;
      continue;
    }
    int became_dup = 0;
    n_new += hash_insert(dst, r.key, want_mult, r.count >= 2, &became_dup);
    n_dup += became_dup;
  }
  block_add2(n_new, n_dup, &counters[0], &counters[1]);
  __syncthreads();
  block_add2(n_empty, n_empty_dup, &counters[2], &counters[5]);
}

// Export: pass 1 counts records per owner, pass 2 scatters them (owner = mix(key) % world).  Both passes
// aggregate per workgroup in LDS first: one global atomic per (workgroup, owner) instead of one per key
// (all ranks' keys contend on only `world` counters otherwise).
constexpr uint32_t kMaxWorld = 256;

__device__ __forceinline__ uint32_t owner_of(uint64_t key, uint32_t world) {
  return (uint32_t)((mix64(key ^ 0x9e3779b97f4a7c15ULL) >> 32) % world);
}

struct HashSource {
  HashSetView v;
  int want_mult;
  typedef KeyRecord Rec;
  __device__ uint64_t items() const { return v.mask + 1; }
  template <class F>
  __device__ void for_each(uint64_t s, F f) const {
    const uint64_t key = v.keys[s];
    if (key == kEmptyKey) return;
    f(key, (want_mult && ((v.dup[s >> 5] >> (s & 31)) & 1)) ? 2u : 1u);
  }
};

struct BitmapSource {
  BitmapView bm;
  int want_mult;
  typedef KeyRecord Rec;
  __device__ uint64_t items() const { return (bm.range + 31) >> 5; }
  template <class F>
  __device__ void for_each(uint64_t w, F f) const {
    uint32_t seen = bm.seen[w];
    const uint32_t twice = want_mult ? bm.twice[w] : 0;
    while (seen) {
      const int b = __builtin_ctz(seen);
      seen &= seen - 1;
      f((uint64_t)bm.base + (w << 5) + (uint64_t)b, ((twice >> b) & 1) ? 2u : 1u);
    }
  }
};

template <class Src>
__global__ __launch_bounds__(256) void export_count_kernel(Src src, uint32_t world,
                                                            unsigned long long *owner_counts) {
  __shared__ unsigned int cnt[kMaxWorld];
  for (uint32_t t = threadIdx.x; t < world; t += 256) cnt[t] = 0;
  __syncthreads();
  const uint64_t n = src.items();
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
    src.for_each(i, [&](uint64_t key, uint32_t) { atomicAdd(&cnt[owner_of(key, world)], 1u); });
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < world; t += 256)
    if (cnt[t]) atomicAdd(&owner_counts[t], (unsigned long long)cnt[t]);
}

template <class Src>
__global__ __launch_bounds__(256) void export_scatter_kernel(Src src, uint32_t world,
                                                              unsigned long long *cursors, KeyRecord *out) {
  __shared__ unsigned int cnt[kMaxWorld];
  __shared__ unsigned long long pos[kMaxWorld];
  const uint64_t n = src.items();
  const uint64_t step = (uint64_t)gridDim.x * 256;
  const uint64_t rounded = (n + step - 1) / step * step;
  for (uint64_t base = (uint64_t)blockIdx.x * 256; base < rounded; base += step) {
    const uint64_t i = base + threadIdx.x;
    for (uint32_t t = threadIdx.x; t < world; t += 256) cnt[t] = 0;
    __syncthreads();
    if (i < n) src.for_each(i, [&](uint64_t key, uint32_t) { atomicAdd(&cnt[owner_of(key, world)], 1u); });
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < world; t += 256)
      pos[t] = cnt[t] ? atomicAdd(&cursors[t], (unsigned long long)cnt[t]) : 0ull;
    __syncthreads();
    if (i < n)
      src.for_each(i, [&](uint64_t key, uint32_t count) {
        const unsigned long long p = atomicAdd(&pos[owner_of(key, world)], 1ull);
        KeyRecord r;
        r.key = key;
        r.count = count;
        out[p] = r;
      });
    __syncthreads();
  }
}

// Cross-rank reduction of congruent range bitmaps: slice k of the result is the OR of slice k of every
// rank's bitmap; a key is "seen twice" if any rank saw it twice or two ranks saw it at all.
__global__ __launch_bounds__(256) void bitmap_adopt_kernel(const uint32_t *seen_slices,
                                                            const uint32_t *twice_slices, uint32_t n_slices,
                                                            uint64_t slice_words, uint64_t stride, uint32_t *out_seen,
                                                            uint32_t *out_twice,
                                                            unsigned long long *counters) {
  unsigned long long n_seen = 0, n_twice = 0;
  for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < slice_words;
       w += (uint64_t)gridDim.x * 256) {
    uint32_t acc_seen = 0, acc_twice = 0;
    for (uint32_t s = 0; s < n_slices; s++) {
      const uint32_t x = seen_slices[(uint64_t)s * stride + w];
      if (twice_slices) acc_twice |= (acc_seen & x) | twice_slices[(uint64_t)s * stride + w];
      acc_seen |= x;
    }
    out_seen[w] = acc_seen;
    if (out_twice) out_twice[w] = acc_twice;
    n_seen += __builtin_popcount(acc_seen);
    n_twice += __builtin_popcount(acc_twice);
  }
  block_add2(n_seen, n_twice, &counters[kCntDistinct], &counters[kCntTwice]);
}

// The send side of the cross-rank exchange (tgx_allreduce): cuts this rank's range bitmap, whose bit 0 stands for
// key `base` of its own choosing, into `world` slices of `slice_words` words on the grid all ranks agreed on -- bit 0
// of slice p stands for key global_lo + p * slice_words * 32.  delta_bits = global_lo - base, so word w of slice p is
// bits [l, l + 32) of the local bitmap with l = (p * slice_words + w) * 32 + delta_bits, zero outside it.  The
// local bitmaps therefore need not be congruent (no range hint, no second pass): re-basing rides on the copy into
// the all-to-all's send buffer.  Block p of the send buffer starts at p * row_words; this column's slice at col_words.
__global__ __launch_bounds__(256) void bitmap_rebase_kernel(const uint32_t *__restrict__ src, uint64_t src_words,
                                                             long long delta_bits, uint32_t world,
                                                             uint64_t slice_words, uint64_t row_words,
                                                             uint64_t col_words, uint32_t *__restrict__ send) {
  const uint64_t total = (uint64_t)world * slice_words;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
    const uint64_t p = i / slice_words, w = i - p * slice_words;
    const long long l = (long long)(i * 32) + delta_bits;
    const long long wi = l >> 5;  // arithmetic shift: floor
    const uint32_t sh = (uint32_t)(l & 31);
    const uint32_t lo = (wi >= 0 && (uint64_t)wi < src_words) ? src[wi] : 0u;
    const uint32_t hi = (wi + 1 >= 0 && (uint64_t)(wi + 1) < src_words) ? src[wi + 1] : 0u;
    send[p * row_words + col_words + w] = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
  }
}

void launch_partition(const PartitionParams &p, unsigned long long *d_counters, int n_cu,
                      hipStream_t stream) {
  int64_t n_tiles = (p.length + kPartitionTile - 1) / kPartitionTile;
  int grid = (int)(n_tiles < (int64_t)n_cu ? n_tiles : (int64_t)n_cu);  // 152 KiB of LDS: one workgroup per CU
  if (grid < 1) grid = 1;
  // (software-pipelined variants -- next tile requested before this tile's stores -- were measured three times:
  //  with spills 3.3 ms, as 512 threads x 64 keys with 256 registers 6.0 ms, and spill-free 2.9 ms against
  //  2.7 ms without: the load and store phases already run at HBM rate and the other CUs fill the gaps)
  const bool plain = p.force_form != 2, in_order = p.force_form != 1;  // (force_form 0: both, the flag picks)
#define TGX_PART(VAL, K16, ST)                                                                                      \
  do {                                                                                                              \
    if (plain)                                                                                                      \
      hipLaunchKernelGGL((partition_kernel<kPartitionThreads, kPartitionKeysPerThread, (int)kMaxPartitions, kRunPad4, VAL, K16, ST, false>), \
                         dim3(grid), dim3(kPartitionThreads), 0, stream, p, d_counters);                            \
    if (in_order)                                                                                                   \
      hipLaunchKernelGGL((partition_kernel<kPartitionThreads, kPartitionKeysPerThread, (int)kMaxPartitions, kRunPad4, VAL, K16, ST, true>), \
                         dim3(grid), dim3(kPartitionThreads), 0, stream, p, d_counters);                            \
  } while (0)
#define TGX_PART20(VAL, ST)                                                                                         \
  do {                                                                                                              \
    if (plain)                                                                                                      \
      hipLaunchKernelGGL((partition_kernel<kPartitionThreads, kPartitionKeysPerThread, (int)kMaxPartitions, kRunPad4, VAL, false, ST, false, true>), \
                         dim3(grid), dim3(kPartitionThreads), 0, stream, p, d_counters);                            \
    if (in_order)                                                                                                   \
      hipLaunchKernelGGL((partition_kernel<kPartitionThreads, kPartitionKeysPerThread, (int)kMaxPartitions, kRunPad4, VAL, false, ST, true, true>), \
                         dim3(grid), dim3(kPartitionThreads), 0, stream, p, d_counters);                            \
  } while (0)
  // the 20-bit lists without multiplicity, up to 1024 buckets: the runs' remainders are carried from tile to tile
  // (partition_carry_kernel; keys in order keep the CLUSTERED form of partition_kernel) -- an experiment that lost
  // (see the kernel's comment): only with TGX_PARTITION_CARRY=1
  static const bool carry_on = [] {
    const char *e = getenv("TGX_PARTITION_CARRY");
    return e && e[0] == '1';
  }();
  if (p.key16 == 2 && carry_on && !p.want_multiplicity && p.n_buckets <= 1024) {
#define TGX_CARRY(VAL, ST)                                                                                           \
  do {                                                                                                              \
    if (p.force_form != 2)                                                                                          \
      hipLaunchKernelGGL((partition_carry_kernel<kPartitionThreads, 16, VAL, ST>), dim3(grid), dim3(kPartitionThreads), 0, \
                         stream, p, d_counters);                                                                    \
    if (p.force_form != 1)                                                                                          \
      hipLaunchKernelGGL((partition_kernel<kPartitionThreads, kPartitionKeysPerThread, (int)kMaxPartitions, kRunPad4, VAL, false, ST, true, true>), \
                         dim3(grid), dim3(kPartitionThreads), 0, stream, p, d_counters);                            \
  } while (0)
    if (p.stats) {
      if (p.validity) TGX_CARRY(true, true); else TGX_CARRY(false, true);
    } else {
      if (p.validity) TGX_CARRY(true, false); else TGX_CARRY(false, false);
    }
#undef TGX_CARRY
    return;
  }
  if (p.key16 == 2) {  // 20-bit packed entries
    if (p.stats) {
      if (p.validity) TGX_PART20(true, true); else TGX_PART20(false, true);
    } else {
      if (p.validity) TGX_PART20(true, false); else TGX_PART20(false, false);
    }
    return;
  }
  if (p.stats) {
    if (p.key16) {
      if (p.validity) TGX_PART(true, true, true); else TGX_PART(false, true, true);
    } else {
      if (p.validity) TGX_PART(true, false, true); else TGX_PART(false, false, true);
    }
  } else if (p.key16) {
    if (p.validity) TGX_PART(true, true, false); else TGX_PART(false, true, false);
  } else {
    if (p.validity) TGX_PART(true, false, false); else TGX_PART(false, false, false);
  }
#undef TGX_PART
#undef TGX_PART20
}

hipError_t launch_bucket_apply(const PartitionParams &p, unsigned long long *d_counters,
                               hipStream_t stream) {
  const size_t words = ((size_t)1 << p.sub_bits) / 32 * (p.want_multiplicity ? 2 : 1);
  const dim3 grid(p.n_buckets), block(kPartitionThreads);
#define TGX_APPLY(W)                                                                                  \
  if (words <= W) {                                                                                   \
    if (p.key16 == 2)                                                                                 \
      hipLaunchKernelGGL((bucket_apply_kernel<W, false, true>), grid, block, 0, stream, p, d_counters); \
    else if (p.key16)                                                                                 \
      hipLaunchKernelGGL((bucket_apply_kernel<W, true>), grid, block, 0, stream, p, d_counters);      \
    else                                                                                              \
      hipLaunchKernelGGL((bucket_apply_kernel<W, false>), grid, block, 0, stream, p, d_counters);     \
    return hipGetLastError();                                                                         \
  }
  TGX_APPLY(1024)
  TGX_APPLY(2048)
  TGX_APPLY(4096)
  TGX_APPLY(8192)
  TGX_APPLY(16384)
  TGX_APPLY(32768)
#undef TGX_APPLY
  return hipErrorInvalidValue;
}

static inline int grid_for(uint64_t items) {
  uint64_t blocks = (items + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 8) blocks = 256 * 8;
  return (int)blocks;
}

// ---- big batches of keys that have no dense range (sparse Int64 ids, Float64): no global atomic per key -----------
// The hash set takes one 64-byte read-modify-write at the memory side per KEY (~27 G keys/s whatever the table's
// size).  A batch big enough to care goes the way of the big Utf8 batches instead (lists.h, distinct128.hip): every
// key is replaced by its splitmix64 mix -- a bijection, so equal records mean equal keys: the count is EXACT -- and
// the mixes are partitioned twice by 8 bits into kFpFan^2 lists that are deduplicated one by one in LDS.  The lists
// are the key set until something needs the table (key_insert_kernel un-mixes them into it); a list that overflows
// (heavily repeated keys) flags the batch for a redo through the table.
__device__ __forceinline__ uint64_t unmix64(uint64_t x) {  // inverse of mix64
  x ^= (x >> 31) ^ (x >> 62);
  x *= 0x319642b2d24d8ec3ULL;
  x ^= (x >> 27) ^ (x >> 54);
  x *= 0x96de1b173f119089ULL;
  x ^= (x >> 30) ^ (x >> 60);
  return x;
}

// level 1: a tile of the column's keys -> per-XCD lists by the top byte of their mix
__global__ __launch_bounds__(256) void key_partition_values_kernel(DistinctColDesc d, FpLists out,
                                                                    unsigned long long *counters) {
  constexpr int PER = kFpTile / 256;
  __shared__ FpTileLdsT<KeyRec> s;
  const uint32_t tid = threadIdx.x;
  fp_tile_begin(s);
  global_i64_ptr vals = (global_i64_ptr)(uintptr_t)((const int64_t *)d.values + d.offset);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  const int64_t first = (int64_t)blockIdx.x * kFpTile;
  KeyRec mine[PER];
  uint32_t present = 0, n_empty = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int64_t row = first + k * 256 + (int64_t)tid;
    mine[k] = 0;
    if (row < d.length) {
      bool valid = true;
      if (vbits) {
        const int64_t b = d.offset + row;
        valid = (vbits[b >> 3] >> (b & 7)) & 1;
      }
      if (valid) {
        const uint64_t key = (uint64_t)vals[row];
        if (key == kEmptyKey) {
          n_empty++;  // the table's free-slot marker: counted on the side, as everywhere
        } else {
          mine[k] = mix64(key);
          present |= 1u << k;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < PER; k++)
    if ((present >> k) & 1u) atomicAdd(&s.hist[mine[k] >> 56], 1u);
  if (n_empty) {  // (rare)
    atomicAdd(&counters[kCntEmptyRows], (unsigned long long)n_empty);
    atomicAdd(&counters[kCntValidRows], (unsigned long long)n_empty);
  }
  __syncthreads();
  fp_tile_scatter(s, mine, present, (blockIdx.x % kFpXcds) * kFpFan, out, 56, counters);
}

// the lists' records back to keys and into the global table (counted already: no counters)
__global__ __launch_bounds__(256) void key_insert_kernel(FpLists l, HashSetView t, int want_mult) {
  const uint32_t offered = l.offered[blockIdx.x];
  const uint32_t n = offered < l.cap ? offered : (uint32_t)l.cap;
  const KeyRec *recs = (const KeyRec *)l.recs + (uint64_t)blockIdx.x * l.cap;
  for (uint32_t i = threadIdx.x; i < n; i += 256) {
    int became_dup = 0;
    (void)hash_insert(t, unmix64(recs[i]), want_mult, 0, &became_dup);
  }
}

void launch_key_lists(const DistinctColDesc &d, const FpLists &level1, const FpLists &level2, int want_mult,
                      uint2 *per_list, unsigned long long *d_counters, hipStream_t stream) {
  const int64_t tiles = (d.length + kFpTile - 1) / kFpTile;
  hipLaunchKernelGGL(key_partition_values_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, d, level1, d_counters);
  const uint32_t tiles_per_list = (uint32_t)((level1.cap + kFpTile - 1) / kFpTile);
  hipLaunchKernelGGL(fp_partition_lists_kernel<KeyRec>, dim3(kFpXcds * kFpFan * tiles_per_list), dim3(256), 0, stream,
                     level1, tiles_per_list, level2, d_counters);
  const dim3 grid(kFpFan * kFpFan);
  if (level2.cap <= 3072)
    hipLaunchKernelGGL((fp_count_kernel<4096, 256, KeyRec>), grid, dim3(256), 0, stream, level2, want_mult, per_list, (uint32_t)(kFpFan * kFpFan), PlainEq());
  else if (level2.cap <= 12288)
    hipLaunchKernelGGL((fp_count_kernel<16384, 1024, KeyRec>), grid, dim3(1024), 0, stream, level2, want_mult, per_list, (uint32_t)(kFpFan * kFpFan), PlainEq());
  else
    hipLaunchKernelGGL((fp_count_kernel<32768, 1024, KeyRec>), dim3(fp_resident_grid()), dim3(1024), 0, stream, level2, want_mult,
                       per_list, (uint32_t)(kFpFan * kFpFan), PlainEq());  // one workgroup per CU, each walking its share of the lists
  hipLaunchKernelGGL(fp_totals_kernel<KeyRec>, dim3(64), dim3(256), 0, stream, per_list, (uint32_t)(kFpFan * kFpFan),
                     level1.offered, d_counters);
}

void launch_key_insert(const FpLists &level2, const HashSetView &t, int want_mult, hipStream_t stream) {
  hipLaunchKernelGGL(key_insert_kernel, dim3(kFpFan * kFpFan), dim3(256), 0, stream, level2, t, want_mult);
}

void launch_distinct_hash(const DistinctColDesc &d, const HashSetView &t,
                          unsigned long long *d_counters, hipStream_t stream) {
  hipLaunchKernelGGL(distinct_hash_kernel, dim3(grid_for((uint64_t)d.length)), dim3(256), 0, stream,
                     d, t, d_counters);
}

void launch_distinct_bitmap(const DistinctColDesc &d, const BitmapView &bm,
                            unsigned long long *d_counters, hipStream_t stream) {
  hipLaunchKernelGGL(distinct_bitmap_kernel, dim3(grid_for((uint64_t)d.length)), dim3(256), 0,
                     stream, d, bm, d_counters);
}

void launch_hash_rehash(const HashSetView &src, const HashSetView &dst, int want_mult,
                        unsigned long long *d_counters, hipStream_t stream) {
  hipLaunchKernelGGL(hash_rehash_kernel, dim3(grid_for(src.mask + 1)), dim3(256), 0, stream, src,
                     dst, want_mult, d_counters);
}

void launch_bitmap_to_hash(const BitmapView &bm, const HashSetView &dst, int want_mult,
                           unsigned long long *d_counters, hipStream_t stream) {
  hipLaunchKernelGGL(bitmap_to_hash_kernel, dim3(grid_for((bm.range + 31) >> 5)), dim3(256), 0,
                     stream, bm, dst, want_mult, d_counters);
}

void launch_hash_import(const KeyRecord *recs, uint64_t n, const HashSetView &dst, int want_mult,
                        unsigned long long *d_counters, hipStream_t stream) {
  if (n == 0) return;
  hipLaunchKernelGGL(hash_import_kernel, dim3(grid_for(n)), dim3(256), 0, stream, recs, n, dst,
                     want_mult, d_counters);
}

static void check_world(uint32_t world) { (void)world; }

void launch_hash_export_count(const HashSetView &src, uint32_t world, unsigned long long *d_counts,
                              hipStream_t stream) {
  check_world(world);
  HashSource s{src, 0};
  hipLaunchKernelGGL(export_count_kernel<HashSource>, dim3(grid_for(src.mask + 1)), dim3(256), 0, stream, s,
                     world, d_counts);
}

void launch_hash_export_scatter(const HashSetView &src, uint32_t world, int want_mult,
                                unsigned long long *d_cursors, KeyRecord *out, hipStream_t stream) {
  HashSource s{src, want_mult};
  hipLaunchKernelGGL(export_scatter_kernel<HashSource>, dim3(grid_for(src.mask + 1)), dim3(256), 0, stream, s,
                     world, d_cursors, out);
}

void launch_bitmap_export_count(const BitmapView &bm, uint32_t world, unsigned long long *d_counts,
                                hipStream_t stream) {
  BitmapSource s{bm, 0};
  hipLaunchKernelGGL(export_count_kernel<BitmapSource>, dim3(grid_for((bm.range + 31) >> 5)), dim3(256), 0,
                     stream, s, world, d_counts);
}

void launch_bitmap_export_scatter(const BitmapView &bm, uint32_t world, int want_mult,
                                  unsigned long long *d_cursors, KeyRecord *out, hipStream_t stream) {
  BitmapSource s{bm, want_mult};
  hipLaunchKernelGGL(export_scatter_kernel<BitmapSource>, dim3(grid_for((bm.range + 31) >> 5)), dim3(256), 0,
                     stream, s, world, d_cursors, out);
}

void launch_bitmap_rebase(const uint32_t *src, uint64_t src_words, long long delta_bits, uint32_t world,
                          uint64_t slice_words, uint64_t row_words, uint64_t col_words, uint32_t *send,
                          hipStream_t stream) {
  hipLaunchKernelGGL(bitmap_rebase_kernel, dim3(grid_for((uint64_t)world * slice_words)), dim3(256), 0, stream, src,
                     src_words, delta_bits, world, slice_words, row_words, col_words, send);
}

void launch_bitmap_adopt(const uint32_t *seen_slices, const uint32_t *twice_slices, uint32_t n_slices,
                         uint64_t slice_words, uint64_t stride_words, uint32_t *out_seen, uint32_t *out_twice,
                         unsigned long long *d_counters, hipStream_t stream) {
  hipLaunchKernelGGL(bitmap_adopt_kernel, dim3(grid_for(slice_words)), dim3(256), 0, stream, seen_slices,
                     twice_slices, n_slices, slice_words, stride_words, out_seen, out_twice, d_counters);
}


int partition_grid(int64_t length, int n_cu) {
  int64_t n_tiles = (length + kPartitionTile - 1) / kPartitionTile;
  int grid = (int)(n_tiles < (int64_t)n_cu ? n_tiles : (int64_t)n_cu);
  return grid < 1 ? 1 : grid;
}

// identities of the two small accumulators (a kernel, not two copies from pageable host memory: those stall the
// caller for a staging round trip each)
__global__ void distinct_init_kernel(DistinctSample *sample, OutlierStats *outliers) {
  if (threadIdx.x != 0) return;
  if (sample) {
    sample->min_v = INT64_MAX;
    sample->max_v = INT64_MIN;
    sample->count = 0;
    sample->pad = 0;
  }
  if (outliers) {
    outliers->mn = INT64_MAX;
    outliers->mx = INT64_MIN;
    outliers->lo32_sum = 0;
    outliers->hi32_sum = 0;
    outliers->count = 0;
  }
}

// everything a partition pass wants cleared, in ONE launch (four small fills / kernels in a row were 20 us of a
// 100 M-row step per key column): the lists' cursors (zero) and valid-length limits (all-ones), the outliers'
// aggregates, and the two totals the replay recomputes
__global__ __launch_bounds__(1024) void partition_init_kernel(PartitionParams p, unsigned long long *totals) {
  unsigned long long *cursors = p.cursors;
  const uint32_t n_buckets = p.n_buckets;
  OutlierStats *outliers = p.outliers;
  for (uint32_t b = threadIdx.x; b < n_buckets; b += blockDim.x) {
    cursors[b] = 0;
    cursors[n_buckets + b] = ~0ull;
  }
  if (threadIdx.x == 0) {
    if (outliers) {
      outliers->mn = INT64_MAX;
      outliers->mx = INT64_MIN;
      outliers->lo32_sum = 0;
      outliers->hi32_sum = 0;
      outliers->count = 0;
    }
    totals[0] = 0;
    totals[1] = 0;
  }
  // The probe: do keys that sit next to each other in the column fall into the same bucket?  64 groups of 128
  // consecutive rows, evenly spread (what a wave of partition_kernel holds for one j): a group agrees when every valid
  // key inside the range names one bucket.  Three quarters agreeing -> cursors[2 P] = 1 and partition_kernel runs its
  // CLUSTERED passes; shuffled keys never agree, keys in order always do (but for the groups that straddle a boundary).
  __shared__ uint32_t agree, asked;
  if (threadIdx.x == 0) agree = asked = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_groups = p.length / 128;
  global_i64_ptr vals = (global_i64_ptr)(uintptr_t)((const int64_t *)p.values + p.offset);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)p.validity;
  // (one workgroup, all latency: a wave's four groups are requested together, then looked at)
  uint64_t rr[4][2];
  uint32_t vb[4][2];
  bool have[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t g = wave + 16u * k;
    have[k] = (int64_t)g < n_groups;  // (g < 64: sixteen waves)
    const int64_t group = n_groups <= 64 ? (int64_t)g : (int64_t)g * (n_groups / 64);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int64_t i = group * 128 + h * 64 + lane;
      rr[k][h] = have[k] ? (uint64_t)vals[i] - (uint64_t)p.base : ~0ull;
      vb[k][h] = (have[k] && vbits) ? (uint32_t)((vbits[(p.offset + i) >> 3] >> ((p.offset + i) & 7)) & 1) : 1u;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (!have[k]) continue;  // (wave-uniform)
    bool same = true;
    uint32_t b0 = 0xFFFFFFFFu;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint64_t r = rr[k][h];
      const bool okr = r < p.range && vb[k][h] != 0;
      const uint64_t act = __ballot(okr);
      if (act == 0) continue;  // (nothing inside the range: a group of outliers agrees -- CLUSTERED is their path too)
      const uint32_t b = (uint32_t)(r >> p.sub_bits);
      if (b0 == 0xFFFFFFFFu) b0 = (uint32_t)__builtin_amdgcn_readlane((int)b, (int)__builtin_ctzll(act));
      same = same && __ballot(okr && b != b0) == 0;
    }
    if (lane == 0) {
      atomicAdd(&asked, 1u);
      if (same) atomicAdd(&agree, 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long in_order = (p.probe && asked >= 4 && 4 * agree >= 3 * asked) ? 1ull : 0ull;
    cursors[2 * n_buckets] = in_order;
    totals[kCntForm] = 1ull + in_order;  // (the host reads it with the counters: PartitionParams::force_form)
  }
}

void launch_partition_init(const PartitionParams &p, unsigned long long *totals, hipStream_t stream) {
  hipLaunchKernelGGL(partition_init_kernel, dim3(1), dim3(1024), 0, stream, p, totals);
}

void launch_distinct_init(DistinctSample *sample, OutlierStats *outliers, hipStream_t stream) {
  hipLaunchKernelGGL(distinct_init_kernel, dim3(1), dim3(64), 0, stream, sample, outliers);
}

void launch_distinct_sample(const DistinctColDesc &d, DistinctSample *out, hipStream_t stream) {
  const int grid = d.pad ? (int)std::min<int64_t>(1024, (d.length + 4095) / 4096) : 64;
  hipLaunchKernelGGL(distinct_sample_kernel, dim3(grid < 1 ? 1 : grid), dim3(256), 0, stream, d, out);
}

void launch_distinct_outliers(const DistinctColDesc &d, int64_t base, uint64_t range, const HashSetView &t,
                              unsigned long long *d_counters, hipStream_t stream) {
  hipLaunchKernelGGL(distinct_outlier_kernel, dim3(grid_for((uint64_t)d.length)), dim3(256), 0, stream, d, base, range,
                     t, d_counters);
}

}  // namespace tgx
