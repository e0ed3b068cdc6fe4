// distinct128.hip -- exact-with-overwhelming-probability COUNT(DISTINCT) for Utf8 columns on gfx950.
//
// Variable-length values are reduced on the fly to 128-bit fingerprints (fingerprint() below: four 32-bit lanes
// over the bytes and the length).  Small batches deduplicate the fingerprints in an open-addressing table of
// 16-byte slots: a slot is claimed with ONE 64-bit CAS on its first word and the owner publishes the second; a
// thread that meets an equal first word and a different second just keeps probing, so no thread ever waits on
// another.  Big batches never touch the table: the fingerprints are partitioned into lists that are deduplicated in
// LDS (fp_* kernels).  Two distinct values collide only if all 128 bits agree: ~2^-128 per pair on data that was
// not built against the (seedless) function, ~1e-21 for 10^9 distinct values (DESIGN.md "Distinct").
// The same table serves multi-batch updates, merges (records of 32 bytes) and the cross-rank exchange.
#include <hip/hip_runtime.h>
#include <string.h>

#include "distinct_types.h"
#include "lists.h"

namespace tgx {

typedef const uint8_t __attribute__((address_space(1))) *global_u8_ptr;
typedef const uint64_t __attribute__((address_space(1))) *global_u64_ptr;
typedef const int32_t __attribute__((address_space(1))) *global_i32_ptr;
typedef const int64_t __attribute__((address_space(1))) *global_i64_ptr;

__device__ __forceinline__ uint64_t mix64w(uint64_t x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

// ---- the 128-bit fingerprint of a value ------------------------------------------------------------------------
// Four 32-bit lanes in the manner of MurmurHash3's x86_128 variant: all arithmetic is 32-bit (a 64-bit multiply is
// four quarter-rate instructions on this chip; the earlier chain of 64-bit mixers cost ~1800 cycles per wave of 28-byte
// values, more than reading them).  The value is absorbed as LOGICAL little-endian 8-byte words (zero-padded last
// word; the length goes in at the end), even words into lanes 0/1 and odd words into lanes 2/3; every step is a
// bijection of the state for a given word and injective in the word for a given state, and the finish is a bijection
// of the 128-bit state, so values of at most 8 bytes never collide and longer ones do with probability ~2^-128.
struct Fp {
  uint32_t h0, h1, h2, h3;
};
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ void fp_init(Fp &s) {
  s.h0 = 0x9e3779b9u;
  s.h1 = 0x7f4a7c15u;
  s.h2 = 0xc2b2ae3du;
  s.h3 = 0x27d4eb4fu;
}
template <bool ODD>
__device__ __forceinline__ void fp_absorb(Fp &s, uint64_t w) {
  uint32_t k0 = (uint32_t)w, k1 = (uint32_t)(w >> 32);
  k0 *= 0x239b961bu;
  k0 = rotl32(k0, 15);
  k0 *= 0xab0e9789u;
  k1 *= 0x38b34ae5u;
  k1 = rotl32(k1, 17);
  k1 *= 0xa1e38b93u;
  if (!ODD) {
    s.h0 = (rotl32(s.h0 ^ k0, 19) + s.h1) * 5u + 0x561ccd1bu;
    s.h1 = (rotl32(s.h1 ^ k1, 17) + s.h2) * 5u + 0x0bcaa747u;
  } else {
    s.h2 = (rotl32(s.h2 ^ k0, 15) + s.h3) * 5u + 0x96cd1c35u;
    s.h3 = (rotl32(s.h3 ^ k1, 13) + s.h0) * 5u + 0x32ac3b17u;
  }
}
__device__ __forceinline__ void fp_finish(Fp s, uint64_t len, uint64_t *fa, uint64_t *fb) {
  const uint32_t n = (uint32_t)len;
  s.h0 ^= n;
  s.h1 ^= n;
  s.h2 ^= n;
  s.h3 ^= n ^ (uint32_t)(len >> 32);
  s.h0 += s.h1 + s.h2 + s.h3;
  s.h1 += s.h0;
  s.h2 += s.h0;
  s.h3 += s.h0;
  s.h0 = fmix32(s.h0);
  s.h1 = fmix32(s.h1);
  s.h2 = fmix32(s.h2);
  s.h3 = fmix32(s.h3);
  s.h0 += s.h1 + s.h2 + s.h3;
  s.h1 += s.h0;
  s.h2 += s.h0;
  s.h3 += s.h0;
  uint64_t a = (uint64_t)s.h0 | ((uint64_t)s.h1 << 32), b = (uint64_t)s.h2 | ((uint64_t)s.h3 << 32);
  if (a == kEmptyKey) a -= 1;  // (the table's free-slot marker)
  if (b == kEmptyKey) b -= 1;
  *fa = a;
  *fb = b;
}

// fingerprint of bytes [p, p+len) in global memory
__device__ __forceinline__ void fingerprint(uintptr_t p, uint64_t len, uint64_t *fa, uint64_t *fb) {
  Fp s;
  fp_init(s);
  // logical 8-byte words of the VALUE (independent of where it sits in memory), assembled from the one
  // or two aligned words that hold them; bytes outside the value are never part of w
  uint64_t remaining = len;
  auto next = [&]() -> uint64_t {
    const uint32_t nb = remaining < 8 ? (uint32_t)remaining : 8u;
    const uint32_t skip = (uint32_t)(p & 7);
    const uintptr_t base = p & ~(uintptr_t)7;
    uint64_t w = *(global_u64_ptr)base >> (8 * skip);
    if (skip + nb > 8) w |= *(global_u64_ptr)(base + 8) << (8 * (8 - skip));
    if (nb < 8) w &= (1ull << (8 * nb)) - 1;
    p += nb;
    remaining -= nb;
    return w;
  };
  while (remaining > 0) {
    fp_absorb<false>(s, next());
    if (remaining == 0) break;
    fp_absorb<true>(s, next());
  }
  fp_finish(s, len, fa, fb);
}

__device__ __forceinline__ void block_add2w(unsigned long long a, unsigned long long b,
                                            unsigned long long *ga, unsigned long long *gb) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    a += __shfl_down(a, d, 64);
    b += __shfl_down(b, d, 64);
  }
  __shared__ unsigned long long sa[4], sb[4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sa[wave] = a;
    sb[wave] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long ta = sa[0] + sa[1] + sa[2] + sa[3], tb = sb[0] + sb[1] + sb[2] + sb[3];
    if (ta) atomicAdd(ga, ta);
    if (tb) atomicAdd(gb, tb);
  }
}

// returns 1 if (a, b) was new; *became_dup = 1 if this insert marks the key as seen twice
__device__ __forceinline__ int hash_insert128(const HashSetView &t, uint64_t a, uint64_t b, int want_mult,
                                              int weight_two, int *became_dup) {
  // ONE read-modify-write per new key: the CAS on the first word claims the slot, the owner then publishes the
  // second word with a plain (device-scope) store.  Every atomic is executed at the memory side as a 64-byte
  // read-modify-write -- with a CAS on each word the kernel wrote 142 bytes per inserted key.
  // A lane that meets its own first word in a slot whose second word is not there yet goes round the probe loop
  // again WITHOUT moving on (no nested spin: the owner may be a lane of the same wave, which publishes in this same
  // loop body before the wave comes round).
  uint64_t h = a & t.mask;
  for (;;) {
    unsigned long long *w0 = (unsigned long long *)&t.keys[2 * h];
    unsigned long long *w1 = w0 + 1;
    const unsigned long long old0 = atomicCAS(w0, (unsigned long long)kEmptyKey, (unsigned long long)a);
    const uint32_t bit = 1u << (h & 31);
    if (old0 == kEmptyKey) {
      __hip_atomic_store(w1, (unsigned long long)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (want_mult && weight_two) {
        const uint32_t prev = atomicOr(&t.dup[h >> 5], bit);
        *became_dup = (prev & bit) ? 0 : 1;
      }
      return 1;
    }
    if (old0 == a) {
      const unsigned long long old1 = __hip_atomic_load(w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old1 == kEmptyKey) continue;  // claimed, second word on its way: look at this slot again
      if (old1 == b) {
        if (want_mult) {
          if (!(__hip_atomic_load(&t.dup[h >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) {
            const uint32_t prev = atomicOr(&t.dup[h >> 5], bit);
            *became_dup = (prev & bit) ? 0 : 1;
          }
        }
        return 0;
      }
    }
    h = (h + 1) & t.mask;
  }
}

struct Utf8ColDesc {
  const void *offsets;
  const uint8_t *data;
  const uint8_t *validity;
  int64_t offset;
  int64_t length;
  int32_t large_offsets;
  int32_t want_multiplicity;
  const void *views;              // Utf8View: 16-byte views (then offsets / data are unused)
  const uint8_t *const *buffers;  // Utf8View: device array of the data buffers' device pointers
};

// where the value of slot `slot` lies
__device__ __forceinline__ void utf8_value(const Utf8ColDesc &d, int64_t slot, uintptr_t *p, uint64_t *len) {
  if (d.views) {
    global_i32_ptr vw = (global_i32_ptr)((uintptr_t)d.views + (uintptr_t)slot * 16);
    const int32_t n = vw[0];
    *len = (uint64_t)n;
    if (n <= 12) {
      *p = (uintptr_t)d.views + (uintptr_t)slot * 16 + 4;
    } else {
      const int32_t bi = vw[2], bo = vw[3];
      *p = (uintptr_t)d.buffers[bi] + (uintptr_t)(uint32_t)bo;
    }
  } else if (d.large_offsets) {
    global_i64_ptr off = (global_i64_ptr)(uintptr_t)d.offsets;
    const int64_t b = off[slot];
    *p = (uintptr_t)d.data + (uintptr_t)b;
    *len = (uint64_t)(off[slot + 1] - b);
  } else {
    global_i32_ptr off = (global_i32_ptr)(uintptr_t)d.offsets;
    const int32_t b = off[slot];
    *p = (uintptr_t)d.data + (uintptr_t)b;
    *len = (uint64_t)(off[slot + 1] - b);
  }
}

__global__ __launch_bounds__(256) void distinct_utf8_kernel(Utf8ColDesc d, HashSetView t,
                                                             unsigned long long *counters) {
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  unsigned long long n_new = 0, n_dup = 0, n_valid = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.length; i += stride) {
    const int64_t slot = d.offset + i;
    if (vbits && !((vbits[slot >> 3] >> (slot & 7)) & 1)) continue;
    n_valid++;
    uintptr_t p;
    uint64_t len, fa, fb;
    utf8_value(d, slot, &p, &len);
    fingerprint(p, len, &fa, &fb);
    int became_dup = 0;
    n_new += hash_insert128(t, fa, fb, d.want_multiplicity, 0, &became_dup);
    n_dup += became_dup;
  }
  block_add2w(n_new, n_dup, &counters[kCntDistinct], &counters[kCntTwice]);
  __syncthreads();
  block_add2w(n_valid, 0ull, &counters[kCntValidRows], &counters[kCntSpare]);
}

// ---- big batches: no global atomic per value ------------------------------------------------------------------
// A global-table insert is a 64-byte read-modify-write at the memory side per VALUE (the chip does ~20 G of them a
// second whatever the table's size: 5 ms per 100 M values before a byte of string is read).  A batch big enough to
// care is instead reduced to its fingerprints, which are range-partitioned twice -- 8 bits of the first word each
// time, a tile of kFpTile records grouped in LDS so that a list receives whole runs -- into kFpFan^2 lists of a few
// thousand records; fp_count_kernel then deduplicates a list in an LDS table and the counts are summed.  The lists
// ARE the key set until somebody needs the table (a second batch, a merge, an export): fp_insert_kernel then moves
// them in.  A list that receives more records than it can hold (heavily repeated values) drops them and says so in
// kCntOutOfRange: the host redoes the batch through the global table (tgx_api.cpp, fp_resolve).
typedef FpTileLdsT<ulonglong2> FpTileLds;

// 16-byte records carry "none" as kEmptyKey in their first word (no fingerprint's first word: fingerprint())
__device__ __forceinline__ void fp_tile_scatter16(FpTileLds &s, const ulonglong2 (&mine)[kFpTile / 256], uint32_t out_list0,
                                                  const FpLists &out, int shift, unsigned long long *counters) {
  uint32_t present = 0;
#pragma unroll
  for (int k = 0; k < kFpTile / 256; k++) present |= (mine[k].x != kEmptyKey ? 1u : 0u) << k;
  fp_tile_scatter(s, mine, present, out_list0, out, shift, counters);
}

constexpr uint32_t kFpStageBytes = 4080;  // value bytes of 128 consecutive rows a wave stages at a time (255 blocks)
constexpr uint32_t kFpStageAlloc = 4096 + 32;  // what the stage holds: 4 blocks per lane + slack for the read-ahead

// fingerprint() of TWO values staged in LDS (same words, same results), walked in lockstep and without branches in
// the loop body: both chains and all their LDS reads are in flight together.  A value that has ended keeps reading
// (and discarding) what follows it; the addresses are kept inside the stage.
__device__ __forceinline__ void fingerprint_lds2(const uint8_t *stage, uint32_t o0, uint32_t len0, uint32_t o1,
                                                 uint32_t len1, ulonglong2 *f0, ulonglong2 *f1) {
  Fp s0, s1;
  fp_init(s0);
  fp_init(s1);
  uint32_t r0 = len0, r1 = len1;
  auto word = [&](uint32_t &o, uint32_t &rem, bool *live) -> uint64_t {
    *live = rem > 0;
    const uint32_t nb = rem < 8 ? rem : 8u;
    const uint32_t skip = o & 7;
    uint32_t base = o & ~7u;
    base = base < 4096u + 16u ? base : 4096u + 16u;
    const uint64_t lo = *(const uint64_t *)(stage + base), hi = *(const uint64_t *)(stage + base + 8);
    uint64_t w = lo >> (8 * skip);
    if (skip) w |= hi << (8 * (8 - skip));  // (bytes past the value are masked off below)
    if (nb < 8) w &= (1ull << (8 * nb)) - 1;
    o += nb;
    rem -= nb;
    return w;
  };
  while ((r0 | r1) != 0) {
    bool l0, l1;
    uint64_t w0 = word(o0, r0, &l0), w1 = word(o1, r1, &l1);
    Fp t0 = s0, t1 = s1;
    fp_absorb<false>(t0, w0);
    fp_absorb<false>(t1, w1);
    s0.h0 = l0 ? t0.h0 : s0.h0;
    s0.h1 = l0 ? t0.h1 : s0.h1;
    s1.h0 = l1 ? t1.h0 : s1.h0;
    s1.h1 = l1 ? t1.h1 : s1.h1;
    w0 = word(o0, r0, &l0);
    w1 = word(o1, r1, &l1);
    t0 = s0;
    t1 = s1;
    fp_absorb<true>(t0, w0);
    fp_absorb<true>(t1, w1);
    s0.h2 = l0 ? t0.h2 : s0.h2;
    s0.h3 = l0 ? t0.h3 : s0.h3;
    s1.h2 = l1 ? t1.h2 : s1.h2;
    s1.h3 = l1 ? t1.h3 : s1.h3;
  }
  fp_finish(s0, (uint64_t)len0, (uint64_t *)&f0->x, (uint64_t *)&f0->y);
  fp_finish(s1, (uint64_t)len1, (uint64_t *)&f1->x, (uint64_t *)&f1->y);
}

// level 1: a tile of rows -> fingerprints -> the kFpFan lists of bits [56, 64).  A wave takes 128 consecutive rows a
// step, two per lane; their bytes are one span of the value buffer, copied into LDS with 16-byte loads and
// fingerprinted from there (per-lane global loads at a ~28-byte stride read the column at 1.5 TB/s).  The pipeline
// is three steps deep: offsets of step s+2 and bytes of step s+1 (in registers) are in flight while step s is
// fingerprinted.  A span that does not fit the stage is fingerprinted straight from global memory.
__global__ __launch_bounds__(256) void fp_partition_strings_kernel(Utf8ColDesc d, FpLists out,
                                                                    unsigned long long *counters) {
  constexpr int kRowsPerWave = kFpTile / 4, kSteps = kRowsPerWave / 128;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef const u32x4 __attribute__((address_space(1))) *global_u4_ptr;
  __shared__ FpTileLds s;
  static_assert(kFpStageAlloc <= kRowsPerWave * sizeof(ulonglong2), "a wave's value bytes fit its share of the tile");
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the records stay in registers until every wave is through its rows: until then the tile's LDS holds value bytes
  uint8_t *stage = (uint8_t *)&s.stage[wave * kRowsPerWave];
  ulonglong2 mine[kFpTile / 256];
  fp_tile_begin(s);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  const uintptr_t data0 = (uintptr_t)d.data;
  const int64_t wave_first = (int64_t)blockIdx.x * kFpTile + (int64_t)wave * kRowsPerWave;
  auto offset_at = [&](int64_t row) -> int64_t {  // rows past the end read the end offset: they come out empty
    const int64_t slot = d.offset + (row < d.length ? row : d.length);
    return d.large_offsets ? ((global_i64_ptr)(uintptr_t)d.offsets)[slot]
                           : (int64_t)((global_i32_ptr)(uintptr_t)d.offsets)[slot];
  };
  auto valid_at = [&](int64_t row) -> bool {
    if (row >= d.length) return false;
    const int64_t slot = d.offset + row;
    return !vbits || ((vbits[slot >> 3] >> (slot & 7)) & 1);
  };
  struct Step {
    int64_t b0, b1, tail;
    bool v0, v1;
  };
  auto fetch = [&](int step) -> Step {
    Step t;
    const int64_t i0 = wave_first + step * 128 + lane;
    t.b0 = offset_at(i0);
    t.b1 = offset_at(i0 + 64);
    t.tail = offset_at(wave_first + step * 128 + 128);
    t.v0 = valid_at(i0);
    t.v1 = valid_at(i0 + 64);
    return t;
  };
  // the span of a step's values: [base, tail), base rounded down to a 16-byte block by ABSOLUTE address (a block that
  // holds a byte of the buffer lies in the buffer's pages)
  auto span_of = [&](const Step &t, int64_t *base) -> bool {
    const int64_t b_first = __shfl(t.b0, 0, 64);
    *base = b_first - (int64_t)((data0 + (uintptr_t)b_first) & 15);
    return t.tail - *base <= (int64_t)kFpStageBytes;  // wave-uniform
  };
  u32x4 pre[4];
  auto load_bytes = [&](int64_t base, int64_t tail) {
    global_u4_ptr src = (global_u4_ptr)(data0 + (uintptr_t)base);
    const int64_t n16 = (tail - base + 15) >> 4;  // <= 255
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int64_t k = lane + 64 * j;
      pre[j] = k < n16 ? src[k] : (u32x4)0u;
    }
  };
  Step cur = fetch(0), nxt = fetch(1);
  int64_t base_cur = 0;
  bool fit_cur = span_of(cur, &base_cur);
  if (fit_cur) load_bytes(base_cur, cur.tail);
#pragma unroll
  for (int step = 0; step < kSteps; step++) {
    if (fit_cur) {
#pragma unroll
      for (int j = 0; j < 4; j++) *(u32x4 *)(stage + 16 * (lane + 64 * j)) = pre[j];
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    Step after = nxt;
    if (step + 2 < kSteps) after = fetch(step + 2);
    int64_t base_nxt = 0;
    bool fit_nxt = false;
    if (step + 1 < kSteps) {
      fit_nxt = span_of(nxt, &base_nxt);
      if (fit_nxt) load_bytes(base_nxt, nxt.tail);  // lands while this step is fingerprinted
    }
    // (every shuffle with all lanes active: a row's end is the next row's start)
    const int64_t next0 = __shfl_down(cur.b0, 1, 64), next1 = __shfl_down(cur.b1, 1, 64);
    const int64_t first1 = __shfl(cur.b1, 0, 64);
    const int64_t e0 = lane < 63 ? next0 : first1, e1 = lane < 63 ? next1 : cur.tail;
    ulonglong2 r0, r1;
    if (fit_cur) {
      fingerprint_lds2(stage, (uint32_t)(cur.b0 - base_cur), cur.v0 ? (uint32_t)(e0 - cur.b0) : 0u,
                       (uint32_t)(cur.b1 - base_cur), cur.v1 ? (uint32_t)(e1 - cur.b1) : 0u, &r0, &r1);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();  // every lane is done with the stage
    } else {
      r0.x = r1.x = kEmptyKey;
      if (cur.v0) fingerprint(data0 + (uintptr_t)cur.b0, (uint64_t)(e0 - cur.b0), (uint64_t *)&r0.x, (uint64_t *)&r0.y);
      if (cur.v1) fingerprint(data0 + (uintptr_t)cur.b1, (uint64_t)(e1 - cur.b1), (uint64_t *)&r1.x, (uint64_t *)&r1.y);
    }
    if (!cur.v0) r0.x = kEmptyKey;
    if (!cur.v1) r1.x = kEmptyKey;
    if (r0.x != kEmptyKey) atomicAdd(&s.hist[r0.x >> 56], 1u);
    if (r1.x != kEmptyKey) atomicAdd(&s.hist[r1.x >> 56], 1u);
    mine[2 * step] = r0;
    mine[2 * step + 1] = r1;
    cur = nxt;
    nxt = after;
    base_cur = base_nxt;
    fit_cur = fit_nxt;
  }
  __syncthreads();
  fp_tile_scatter16(s, mine, (blockIdx.x % kFpXcds) * kFpFan, out, 56, counters);
}

// fingerprint() of a value of at most 16 bytes held in two registers (the logical words w0 = bytes 0..7, w1 = 8..15)
__device__ __forceinline__ void fingerprint_words(uint64_t w0, uint64_t w1, uint32_t len, uint64_t *fa, uint64_t *fb) {
  Fp s;
  fp_init(s);
  if (len > 0) {
    if (len < 8) w0 &= (1ull << (8 * len)) - 1;
    fp_absorb<false>(s, w0);
  }
  if (len > 8) {
    if (len < 16) w1 &= (1ull << (8 * (len - 8))) - 1;
    fp_absorb<true>(s, w1);
  }
  fp_finish(s, (uint64_t)len, fa, fb);
}

// level 1 for Utf8View columns.  A value is wherever its view says: inline in the 16 view bytes up to 12 bytes (those
// are fingerprinted from the registers that hold the view), else at an offset of one of the data buffers.  Arrow's
// builders append the long values of consecutive rows one after the other, so a wave first looks whether the long
// values of its 128 rows lie in ONE buffer within a span that fits the stage: then the span is copied into LDS with
// 16-byte loads and fingerprinted there like a plain Utf8 column; otherwise every lane reads its own from global
// memory (3.1 -> 2.6 ms per 100 M x 28 B with the span staged).
__global__ __launch_bounds__(256) void fp_partition_views_kernel(Utf8ColDesc d, FpLists out,
                                                                  unsigned long long *counters) {
  constexpr int kRowsPerWave = kFpTile / 4, kSteps = kRowsPerWave / 128;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef const u32x4 __attribute__((address_space(1))) *global_u4_ptr;
  __shared__ FpTileLds s;
  static_assert(kFpStageAlloc <= kRowsPerWave * sizeof(ulonglong2), "a wave's value bytes fit its share of the tile");
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint8_t *stage = (uint8_t *)&s.stage[wave * kRowsPerWave];  // (the records stay in registers: fp_partition_strings_kernel)
  ulonglong2 mine[kFpTile / 256];
  fp_tile_begin(s);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  const int64_t wave_first = (int64_t)blockIdx.x * kFpTile + (int64_t)wave * kRowsPerWave;
  struct Row {
    u32x4 v;     // the view
    bool valid;  // in range and not NULL (the view of a NULL slot is arbitrary: never interpreted)
  };
  auto row_at = [&](int64_t row) -> Row {
    Row r;
    const bool in = row < d.length;
    const int64_t slot = d.offset + (in ? row : d.length - 1);
    r.valid = in && (!vbits || ((vbits[slot >> 3] >> (slot & 7)) & 1));
    r.v = *(global_u4_ptr)((uintptr_t)d.views + (uintptr_t)slot * 16);
    return r;
  };
  auto wave_min = [](uint32_t x) {
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1) {
      const uint32_t o = __shfl_xor(x, dlt, 64);
      x = o < x ? o : x;
    }
    return x;
  };
  auto wave_max = [](uint32_t x) {
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1) {
      const uint32_t o = __shfl_xor(x, dlt, 64);
      x = o > x ? o : x;
    }
    return x;
  };
  Row n0 = row_at(wave_first + lane), n1 = row_at(wave_first + 64 + lane);
#pragma unroll
  for (int step = 0; step < kSteps; step++) {
    const Row r0 = n0, r1 = n1;
    if (step + 1 < kSteps) {  // the next step's views are requested before this step's bytes are staged
      n0 = row_at(wave_first + (step + 1) * 128 + lane);
      n1 = row_at(wave_first + (step + 1) * 128 + 64 + lane);
    }
    const uint32_t len0 = r0.valid ? r0.v.x : 0u, len1 = r1.valid ? r1.v.x : 0u;
    const bool long0 = len0 > 12, long1 = len1 > 12;
    // one buffer, one short span?  (wave-uniform; a step without long values stages nothing)
    const unsigned long long any_long = __builtin_amdgcn_ballot_w64(long0 || long1);
    bool staged = false;
    int64_t base = 0;  // of the staged span, relative to the buffer (up to 15 bytes in front of the first value)
    uint32_t n16 = 0;
    uintptr_t buf = 0;
    if (any_long) {
      const int first_lane = __builtin_ctzll(any_long);
      const uint32_t bi = (uint32_t)__shfl(long0 ? r0.v.z : r1.v.z, first_lane, 64);
      const bool same = (!long0 || r0.v.z == bi) && (!long1 || r1.v.z == bi);
      const uint32_t lo0 = long0 ? r0.v.w : 0xFFFFFFFFu, lo1 = long1 ? r1.v.w : 0xFFFFFFFFu;
      const uint32_t hi0 = long0 ? r0.v.w + len0 : 0u, hi1 = long1 ? r1.v.w + len1 : 0u;  // (< 2^32: both < 2^31)
      const uint32_t lo = wave_min(lo0 < lo1 ? lo0 : lo1), hi = wave_max(hi0 > hi1 ? hi0 : hi1);
      buf = (uintptr_t)d.buffers[bi];
      base = (int64_t)lo - (int64_t)((buf + lo) & 15);  // 16-byte blocks by ABSOLUTE address, as in the plain kernel
      staged = __builtin_amdgcn_ballot_w64(!same) == 0 && (int64_t)hi - base <= (int64_t)kFpStageBytes;
      n16 = (uint32_t)(((int64_t)hi - base + 15) >> 4);  // <= 255 when staged
    }
    ulonglong2 f0, f1;
    f0.x = f1.x = kEmptyKey;
    f0.y = f1.y = 0;
    if (staged) {
      global_u4_ptr src = (global_u4_ptr)(buf + (uintptr_t)base);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t k = lane + 64 * j;
        if (k < n16) *(u32x4 *)(stage + 16 * k) = src[k];
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      fingerprint_lds2(stage, long0 ? (uint32_t)((int64_t)r0.v.w - base) : 0u, long0 ? len0 : 0u,
                       long1 ? (uint32_t)((int64_t)r1.v.w - base) : 0u, long1 ? len1 : 0u, &f0, &f1);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();  // every lane is done with the stage
    } else {
      if (long0) fingerprint((uintptr_t)d.buffers[r0.v.z] + (uintptr_t)r0.v.w, len0, (uint64_t *)&f0.x, (uint64_t *)&f0.y);
      if (long1) fingerprint((uintptr_t)d.buffers[r1.v.z] + (uintptr_t)r1.v.w, len1, (uint64_t *)&f1.x, (uint64_t *)&f1.y);
    }
    // inline values: bytes 4..15 of the view
    if (r0.valid && !long0)
      fingerprint_words((uint64_t)r0.v.y | ((uint64_t)r0.v.z << 32), (uint64_t)r0.v.w, len0, (uint64_t *)&f0.x, (uint64_t *)&f0.y);
    if (r1.valid && !long1)
      fingerprint_words((uint64_t)r1.v.y | ((uint64_t)r1.v.z << 32), (uint64_t)r1.v.w, len1, (uint64_t *)&f1.x, (uint64_t *)&f1.y);
    if (!r0.valid) f0.x = kEmptyKey;
    if (!r1.valid) f1.x = kEmptyKey;
    if (f0.x != kEmptyKey) atomicAdd(&s.hist[f0.x >> 56], 1u);
    if (f1.x != kEmptyKey) atomicAdd(&s.hist[f1.x >> 56], 1u);
    mine[2 * step] = f0;
    mine[2 * step + 1] = f1;
  }
  __syncthreads();
  fp_tile_scatter16(s, mine, (blockIdx.x % kFpXcds) * kFpFan, out, 56, counters);
}

// the lists' records into the global table (counted already: no counters)
__global__ __launch_bounds__(256) void fp_insert_kernel(FpLists l, HashSetView t, int want_mult) {
  const uint32_t offered = l.offered[blockIdx.x];
  const uint32_t n = offered < l.cap ? offered : (uint32_t)l.cap;
  const ulonglong2 *recs = (const ulonglong2 *)l.recs + (uint64_t)blockIdx.x * l.cap;
  for (uint32_t i = threadIdx.x; i < n; i += 256) {
    const ulonglong2 r = recs[i];
    int became_dup = 0;
    (void)hash_insert128(t, r.x, r.y, want_mult, 0, &became_dup);
  }
}

__global__ __launch_bounds__(256) void hash_rehash128_kernel(HashSetView src, HashSetView dst, int want_mult,
                                                              unsigned long long *counters) {
  unsigned long long n_new = 0, n_dup = 0;
  const uint64_t cap = src.mask + 1;
  for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cap;
       s += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t a = src.keys[2 * s], b = src.keys[2 * s + 1];
    if (a == kEmptyKey) continue;
    const int two = want_mult ? ((src.dup[s >> 5] >> (s & 31)) & 1) : 0;
    int became_dup = 0;
    n_new += hash_insert128(dst, a, b, want_mult, two, &became_dup);
    n_dup += became_dup;
  }
  block_add2w(n_new, n_dup, &counters[kCntDistinct], &counters[kCntTwice]);
}

__global__ __launch_bounds__(256) void hash_import128_kernel(const KeyRecord128 *recs, uint64_t n,
                                                              HashSetView dst, int want_mult,
                                                              unsigned long long *counters) {
  unsigned long long n_new = 0, n_dup = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const KeyRecord128 r = recs[i];
    int became_dup = 0;
    n_new += hash_insert128(dst, r.a, r.b, want_mult, r.count >= 2, &became_dup);
    n_dup += became_dup;
  }
  block_add2w(n_new, n_dup, &counters[kCntDistinct], &counters[kCntTwice]);
}

__device__ __forceinline__ uint32_t owner_of128(uint64_t a, uint64_t b, uint32_t world) {
  return (uint32_t)((mix64w(a ^ rotl64(b, 32) ^ 0x9e3779b97f4a7c15ULL) >> 32) % world);
}

__global__ __launch_bounds__(256) void hash_export_count128_kernel(HashSetView src, uint32_t world,
                                                                    unsigned long long *owner_counts) {
  const uint64_t cap = src.mask + 1;
  for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cap;
       s += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t a = src.keys[2 * s];
    if (a == kEmptyKey) continue;
    atomicAdd(&owner_counts[owner_of128(a, src.keys[2 * s + 1], world)], 1ull);
  }
}

__global__ __launch_bounds__(256) void hash_export_scatter128_kernel(HashSetView src, uint32_t world,
                                                                      int want_mult,
                                                                      unsigned long long *cursors,
                                                                      KeyRecord128 *out) {
  const uint64_t cap = src.mask + 1;
  for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cap;
       s += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t a = src.keys[2 * s];
    if (a == kEmptyKey) continue;
    const uint64_t b = src.keys[2 * s + 1];
    const unsigned long long pos = atomicAdd(&cursors[owner_of128(a, b, world)], 1ull);
    KeyRecord128 r;
    r.a = a;
    r.b = b;
    r.count = (want_mult && ((src.dup[s >> 5] >> (s & 31)) & 1)) ? 2 : 1;
    r.pad = 0;
    out[pos] = r;
  }
}

// ---- tuples of columns: COUNT(DISTINCT (a, b, ...)) / GROUP BY a, b, ... ------------------------------------
// Every component is reduced to 128 bits (numeric: its bit pattern; string: the fingerprint above; NULL: a
// marker no value maps to) and the components are chained position by position into the tuple's fingerprint.
__device__ __forceinline__ void tuple_fingerprint(const TupleDesc &d, int64_t i, uint64_t *out_a, uint64_t *out_b,
                                                  bool *all_valid_out) {
  uint64_t fa = 0x6a09e667f3bcc908ULL, fb = 0xbb67ae8584caa73bULL;
  bool all_valid = true;
  for (int c = 0; c < d.n_cols; c++) {
    const TupleCol &col = d.cols[c];
    const int64_t slot = col.offset + i;
    global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)col.validity;
    uint64_t ca, cb;
    if (vbits && !((vbits[slot >> 3] >> (slot & 7)) & 1)) {
      all_valid = false;
      ca = 0x4e554c4c4e554c4cULL;  // "NULLNULL": tagged below so that no value of any type collides with it
      cb = 0;
    } else if (col.kind == 0) {  // Int64 / Float64: the 64 bits themselves
      ca = (uint64_t)((global_i64_ptr)(uintptr_t)col.values)[slot];
      cb = 1;
    } else {
      int64_t b = 0, e = 0;
      uintptr_t base = (uintptr_t)col.data;
      bool null_entry = false;
      if (col.kind == 4) {  // the row's dictionary entry: the component is the entry's string (or NULL), not the index
        const int64_t ds = col.dict_offset + (int64_t)((global_i32_ptr)(uintptr_t)col.values)[slot];
        global_u8_ptr dbits = (global_u8_ptr)(uintptr_t)col.dict_validity;
        if (dbits && !((dbits[ds >> 3] >> (ds & 7)) & 1)) {
          null_entry = true;
        } else if (col.dict_large) {
          global_i64_ptr off = (global_i64_ptr)(uintptr_t)col.offsets;
          b = off[ds];
          e = off[ds + 1];
        } else {
          global_i32_ptr off = (global_i32_ptr)(uintptr_t)col.offsets;
          b = off[ds];
          e = off[ds + 1];
        }
      } else if (col.kind == 3) {
        global_i32_ptr vw = (global_i32_ptr)((uintptr_t)col.values + (uintptr_t)slot * 16);
        const int32_t len = vw[0];
        b = 0;
        e = len;
        if (len <= 12) {
          base = (uintptr_t)col.values + (uintptr_t)slot * 16 + 4;
        } else {
          const int32_t bi = vw[2], bo = vw[3];
          base = (uintptr_t)col.buffers[bi] + (uintptr_t)(uint32_t)bo;
        }
      } else if (col.kind == 2) {
        global_i64_ptr off = (global_i64_ptr)(uintptr_t)col.offsets;
        b = off[slot];
        e = off[slot + 1];
      } else {
        global_i32_ptr off = (global_i32_ptr)(uintptr_t)col.offsets;
        b = off[slot];
        e = off[slot + 1];
      }
      if (null_entry) {
        all_valid = false;
        ca = 0x4e554c4c4e554c4cULL;
        cb = 0;
      } else {
        fingerprint(base + (uintptr_t)b, (uint64_t)(e - b), &ca, &cb);
        cb |= 2;  // (tag space: 0 NULL, 1 numeric, >= 2 string)
      }
    }
    fa = rotl64(fa ^ mix64w(ca + 0x165667b19e3779f9ULL * (uint64_t)(c + 1)), 27) * 0x9fb21c651e98df25ULL + cb;
    fb = rotl64(fb ^ mix64w(cb ^ rotl64(ca, 32) ^ 0x27d4eb2f165667c5ULL), 31) * 0xd6e8feb86659fd93ULL + ca;
  }
  fa = mix64w(fa);
  fb = mix64w(fb ^ rotl64(fa, 17));
  if (fa == kEmptyKey) fa -= 1;
  if (fb == kEmptyKey) fb -= 1;
  *out_a = fa;
  *out_b = fb;
  *all_valid_out = all_valid;
}

__global__ __launch_bounds__(256) void distinct_tuple_kernel(TupleDesc d, HashSetView t, unsigned long long *counters) {
  unsigned long long n_new = 0, n_dup = 0, n_valid = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.length; i += stride) {
    uint64_t fa, fb;
    bool all_valid;
    tuple_fingerprint(d, i, &fa, &fb, &all_valid);
    n_valid += all_valid ? 1 : 0;
    int became_dup = 0;
    n_new += hash_insert128(t, fa, fb, d.want_multiplicity, 0, &became_dup);
    n_dup += became_dup;
  }
  block_add2w(n_new, n_dup, &counters[kCntDistinct], &counters[kCntTwice]);
  __syncthreads();
  block_add2w(n_valid, 0ull, &counters[kCntValidRows], &counters[kCntSpare]);
}

// level 1 of the lists for tuples: EVERY row is a record (a tuple with NULL components is a value of its own); the
// rows whose components are all non-NULL are counted on the side (one add per workgroup)
__global__ __launch_bounds__(256) void fp_partition_tuples_kernel(TupleDesc d, FpLists out, unsigned long long *counters) {
  constexpr int PER = kFpTile / 256;
  __shared__ FpTileLds s;
  __shared__ uint32_t s_valid;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) s_valid = 0;
  fp_tile_begin(s);
  const int64_t first = (int64_t)blockIdx.x * kFpTile;
  uint32_t n_valid = 0;
#pragma unroll 1  // (the fingerprint code once: the records wait in the tile's LDS, row order)
  for (int k = 0; k < PER; k++) {
    const int64_t row = first + k * 256 + (int64_t)tid;
    ulonglong2 r;
    r.x = kEmptyKey;
    r.y = 0;
    if (row < d.length) {
      bool all_valid;
      tuple_fingerprint(d, row, (uint64_t *)&r.x, (uint64_t *)&r.y, &all_valid);
      n_valid += all_valid ? 1u : 0u;
      atomicAdd(&s.hist[r.x >> 56], 1u);
    }
    s.stage[k * 256 + tid] = r;
  }
#pragma unroll
  for (int dlt = 32; dlt >= 1; dlt >>= 1) n_valid += __shfl_down(n_valid, dlt, 64);
  if ((tid & 63) == 0 && n_valid) atomicAdd(&s_valid, n_valid);
  ulonglong2 mine[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) mine[k] = s.stage[k * 256 + tid];  // (its own records: no barrier needed to read them)
  __syncthreads();  // everyone holds its records and has counted them: the tile's LDS is free
  if (tid == 0 && s_valid) atomicAdd(&counters[kCntValidRows], (unsigned long long)s_valid);
  fp_tile_scatter16(s, mine, (blockIdx.x % kFpXcds) * kFpFan, out, 56, counters);
}

// ---- Dictionary<Int32, Utf8> columns (see dict.hip): fingerprints per dictionary entry, inserted with the
// multiplicity the usage pass counted (0 = unreferenced entry, skipped)
__global__ __launch_bounds__(256) void dict_insert_kernel(Utf8ColDesc dict, const uint32_t *seen,
                                                           const uint32_t *twice, HashSetView t,
                                                           unsigned long long *counters) {
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)dict.validity;
  unsigned long long n_new = 0, n_dup = 0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < dict.length; e += (int64_t)gridDim.x * 256) {
    if (!((seen[e >> 5] >> (e & 31)) & 1)) continue;  // unreferenced entry
    const uint32_t u = (dict.want_multiplicity && ((twice[e >> 5] >> (e & 31)) & 1)) ? 2 : 1;
    const int64_t slot = dict.offset + e;
    if (vbits && !((vbits[slot >> 3] >> (slot & 7)) & 1)) continue;  // a NULL dictionary value
    int64_t b, en;
    if (dict.large_offsets) {
      global_i64_ptr off = (global_i64_ptr)(uintptr_t)dict.offsets;
      b = off[slot];
      en = off[slot + 1];
    } else {
      global_i32_ptr off = (global_i32_ptr)(uintptr_t)dict.offsets;
      b = off[slot];
      en = off[slot + 1];
    }
    uint64_t fa, fb;
    fingerprint((uintptr_t)dict.data + (uintptr_t)b, (uint64_t)(en - b), &fa, &fb);
    int became_dup = 0;
    n_new += hash_insert128(t, fa, fb, dict.want_multiplicity, u >= 2, &became_dup);
    n_dup += became_dup;
  }
  block_add2w(n_new, n_dup, &counters[kCntDistinct], &counters[kCntTwice]);
}

void launch_dict_insert(const void *offsets, const uint8_t *data, const uint8_t *validity, int64_t offset,
                        int64_t length, int large_offsets, int want_mult, const uint32_t *seen,
                        const uint32_t *twice, const HashSetView &t, unsigned long long *d_counters, hipStream_t stream) {
  Utf8ColDesc d;
  d.offsets = offsets;
  d.data = data;
  d.views = nullptr;
  d.buffers = nullptr;
  d.validity = validity;
  d.offset = offset;
  d.length = length;
  d.large_offsets = large_offsets;
  d.want_multiplicity = want_mult;
  int64_t blocks = (length + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dict_insert_kernel, dim3((int)blocks), dim3(256), 0, stream, d, seen, twice, t, d_counters);
}

void launch_distinct_tuple(const TupleDesc &d, const HashSetView &t, unsigned long long *d_counters,
                           hipStream_t stream);
static inline int grid_for128(uint64_t items) {
  uint64_t blocks = (items + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 8) blocks = 256 * 8;
  return (int)blocks;
}

void launch_distinct_utf8(const void *offsets, const uint8_t *data, const void *views,
                          const uint8_t *const *buffers, const uint8_t *validity, int64_t offset,
                          int64_t length, int large_offsets, int want_mult, const HashSetView &t,
                          unsigned long long *d_counters, hipStream_t stream) {
  Utf8ColDesc d;
  d.offsets = offsets;
  d.data = data;
  d.views = views;
  d.buffers = buffers;
  d.validity = validity;
  d.offset = offset;
  d.length = length;
  d.large_offsets = large_offsets;
  d.want_multiplicity = want_mult;
  hipLaunchKernelGGL(distinct_utf8_kernel, dim3(grid_for128((uint64_t)length)), dim3(256), 0, stream, d, t,
                     d_counters);
}

void launch_fp_partition_strings(const void *offsets, const uint8_t *data, const uint8_t *validity, int64_t offset,
                                 int64_t length, int large_offsets, const FpLists &level1,
                                 unsigned long long *d_counters, hipStream_t stream) {
  Utf8ColDesc d;
  memset(&d, 0, sizeof(d));
  d.offsets = offsets;
  d.data = data;
  d.validity = validity;
  d.offset = offset;
  d.length = length;
  d.large_offsets = large_offsets;
  const int64_t tiles = (length + kFpTile - 1) / kFpTile;
  hipLaunchKernelGGL(fp_partition_strings_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, d, level1, d_counters);
}

void launch_fp_partition_views(const void *views, const uint8_t *const *buffers, const uint8_t *validity,
                               int64_t offset, int64_t length, const FpLists &level1, unsigned long long *d_counters,
                               hipStream_t stream) {
  Utf8ColDesc d;
  memset(&d, 0, sizeof(d));
  d.views = views;
  d.buffers = buffers;
  d.validity = validity;
  d.offset = offset;
  d.length = length;
  const int64_t tiles = (length + kFpTile - 1) / kFpTile;
  hipLaunchKernelGGL(fp_partition_views_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, d, level1, d_counters);
}

void launch_fp_partition_tuples(const TupleDesc &d, const FpLists &level1, unsigned long long *d_counters,
                                hipStream_t stream) {
  const int64_t tiles = (d.length + kFpTile - 1) / kFpTile;
  hipLaunchKernelGGL(fp_partition_tuples_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, d, level1, d_counters);
}

void launch_fp_partition_lists(const FpLists &level1, const FpLists &level2, unsigned long long *d_counters,
                               hipStream_t stream) {
  const uint32_t tiles_per_list = (uint32_t)((level1.cap + kFpTile - 1) / kFpTile);
  hipLaunchKernelGGL(fp_partition_lists_kernel<ulonglong2>, dim3(kFpXcds * kFpFan * tiles_per_list), dim3(256), 0, stream, level1,
                     tiles_per_list, level2, d_counters);
}

void launch_fp_count(const FpLists &level2, int want_mult, uint2 *per_list, const uint32_t *offered1,
                     unsigned long long *d_counters, hipStream_t stream) {
  // the table holds a list at load <= 3/4: 16 KiB of LDS (eight workgroups a CU) up to 3072 records a list, i.e.
  // batches up to ~157 M rows; 64 / 128 KiB for batches up to ~0.6 / ~1.4 G rows
  const dim3 grid(kFpFan * kFpFan);
  if (level2.cap <= 3072)
    hipLaunchKernelGGL((fp_count_kernel<4096, 256, ulonglong2>), grid, dim3(256), 0, stream, level2, want_mult, per_list, (uint32_t)(kFpFan * kFpFan));
  else if (level2.cap <= 12288)
    hipLaunchKernelGGL((fp_count_kernel<16384, 1024, ulonglong2>), grid, dim3(1024), 0, stream, level2, want_mult, per_list, (uint32_t)(kFpFan * kFpFan));
  else
    hipLaunchKernelGGL((fp_count_kernel<32768, 1024, ulonglong2>), dim3(fp_resident_grid()), dim3(1024), 0, stream, level2, want_mult,
                       per_list, (uint32_t)(kFpFan * kFpFan));  // one workgroup per CU, each walking its share of the lists
  hipLaunchKernelGGL(fp_totals_kernel<ulonglong2>, dim3(64), dim3(256), 0, stream, per_list, (uint32_t)(kFpFan * kFpFan), offered1,
                     d_counters);
}

void launch_fp_insert(const FpLists &level2, const HashSetView &t, int want_mult, hipStream_t stream) {
  hipLaunchKernelGGL(fp_insert_kernel, dim3(kFpFan * kFpFan), dim3(256), 0, stream, level2, t, want_mult);
}

void launch_hash_rehash128(const HashSetView &src, const HashSetView &dst, int want_mult,
                           unsigned long long *d_counters, hipStream_t stream) {
  hipLaunchKernelGGL(hash_rehash128_kernel, dim3(grid_for128(src.mask + 1)), dim3(256), 0, stream, src, dst,
                     want_mult, d_counters);
}

void launch_hash_import128(const KeyRecord128 *recs, uint64_t n, const HashSetView &dst, int want_mult,
                           unsigned long long *d_counters, hipStream_t stream) {
  if (n == 0) return;
  hipLaunchKernelGGL(hash_import128_kernel, dim3(grid_for128(n)), dim3(256), 0, stream, recs, n, dst,
                     want_mult, d_counters);
}

void launch_hash_export_count128(const HashSetView &src, uint32_t world, unsigned long long *d_counts,
                                 hipStream_t stream) {
  hipLaunchKernelGGL(hash_export_count128_kernel, dim3(grid_for128(src.mask + 1)), dim3(256), 0, stream, src,
                     world, d_counts);
}

void launch_hash_export_scatter128(const HashSetView &src, uint32_t world, int want_mult,
                                   unsigned long long *d_cursors, KeyRecord128 *out, hipStream_t stream) {
  hipLaunchKernelGGL(hash_export_scatter128_kernel, dim3(grid_for128(src.mask + 1)), dim3(256), 0, stream, src,
                     world, want_mult, d_cursors, out);
}

void launch_distinct_tuple(const TupleDesc &d, const HashSetView &t, unsigned long long *d_counters,
                           hipStream_t stream) {
  hipLaunchKernelGGL(distinct_tuple_kernel, dim3(grid_for128((uint64_t)d.length)), dim3(256), 0, stream, d, t,
                     d_counters);
}

}  // namespace tgx
