// distinct128.hip -- exact-with-overwhelming-probability COUNT(DISTINCT) for Utf8 columns on gfx950.
//
// Variable-length values are reduced on the fly to 128-bit fingerprints (two independent 64-bit hashes
// of the bytes and the length); the fingerprints are deduplicated in an open-addressing table of
// 16-byte slots.  A slot is claimed with two 64-bit CASes (first word, then second word); a thread that
// finds the first word equal but loses the second to a different fingerprint just keeps probing, so no
// thread ever waits on another (no spinning inside a wave).  Two distinct strings collide only if both
// 64-bit hashes agree: < 2^-64 per pair, ~1e-20 for 10^9 distinct values (DESIGN.md "Distinct").
// The same table serves multi-batch updates, merges (records of 32 bytes) and the cross-rank exchange.
#include <hip/hip_runtime.h>

#include "distinct_types.h"

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

// two independent 64-bit hashes of bytes [p, p+len)
__device__ __forceinline__ void fingerprint(uintptr_t p, uint64_t len, uint64_t *fa, uint64_t *fb) {
  uint64_t a = 0x9e3779b97f4a7c15ULL ^ (len * 0xff51afd7ed558ccdULL);
  uint64_t b = 0xc2b2ae3d27d4eb4fULL ^ (len * 0xc4ceb9fe1a85ec53ULL);
  // logical 8-byte words of the VALUE (independent of where it sits in memory), assembled from the one
  // or two aligned words that hold them; bytes outside the value are never part of w
  uint64_t remaining = len;
  while (remaining > 0) {
    const uint32_t nb = remaining < 8 ? (uint32_t)remaining : 8u;
    const uint32_t skip = (uint32_t)(p & 7);
    const uintptr_t base = p & ~(uintptr_t)7;
    uint64_t w = *(global_u64_ptr)base >> (8 * skip);
    if (skip + nb > 8) w |= *(global_u64_ptr)(base + 8) << (8 * (8 - skip));
    if (nb < 8) w &= (1ull << (8 * nb)) - 1;
    p += nb;
    remaining -= nb;
    a = rotl64(a ^ mix64w(w + 0x165667b19e3779f9ULL), 27) * 0x9fb21c651e98df25ULL + 0x2545f4914f6cdd1dULL;
    b = rotl64(b ^ mix64w(w ^ 0x27d4eb2f165667c5ULL), 31) * 0xd6e8feb86659fd93ULL + 0x85ebca77c2b2ae63ULL;
  }
  a = mix64w(a);
  b = mix64w(b ^ rotl64(a, 17));
  if (a == kEmptyKey) a -= 1;
  if (b == kEmptyKey) b -= 1;
  *fa = a;
  *fb = b;
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

__global__ __launch_bounds__(256) void distinct_utf8_kernel(Utf8ColDesc d, HashSetView t,
                                                             unsigned long long *counters) {
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  unsigned long long n_new = 0, n_dup = 0, n_valid = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.length; i += stride) {
    const int64_t slot = d.offset + i;
    if (vbits && !((vbits[slot >> 3] >> (slot & 7)) & 1)) continue;
    n_valid++;
    int64_t b, e;
    uintptr_t base = (uintptr_t)d.data;
    if (d.views) {
      global_i32_ptr vw = (global_i32_ptr)((uintptr_t)d.views + (uintptr_t)slot * 16);
      const int32_t len = vw[0];
      b = 0;
      e = len;
      if (len <= 12) {
        base = (uintptr_t)d.views + (uintptr_t)slot * 16 + 4;
      } else {
        const int32_t bi = vw[2], bo = vw[3];
        base = (uintptr_t)d.buffers[bi] + (uintptr_t)(uint32_t)bo;
      }
    } else if (d.large_offsets) {
      global_i64_ptr off = (global_i64_ptr)(uintptr_t)d.offsets;
      b = off[slot];
      e = off[slot + 1];
    } else {
      global_i32_ptr off = (global_i32_ptr)(uintptr_t)d.offsets;
      b = off[slot];
      e = off[slot + 1];
    }
    uint64_t fa, fb;
    fingerprint(base + (uintptr_t)b, (uint64_t)(e - b), &fa, &fb);
    int became_dup = 0;
    n_new += hash_insert128(t, fa, fb, d.want_multiplicity, 0, &became_dup);
    n_dup += became_dup;
  }
  block_add2w(n_new, n_dup, &counters[kCntDistinct], &counters[kCntTwice]);
  __syncthreads();
  block_add2w(n_valid, 0ull, &counters[kCntValidRows], &counters[kCntSpare]);
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
__global__ __launch_bounds__(256) void distinct_tuple_kernel(TupleDesc d, HashSetView t, unsigned long long *counters) {
  unsigned long long n_new = 0, n_dup = 0, n_valid = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.length; i += stride) {
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
        int64_t b, e;
        uintptr_t base = (uintptr_t)col.data;
        if (col.kind == 3) {
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
        fingerprint(base + (uintptr_t)b, (uint64_t)(e - b), &ca, &cb);
        cb |= 2;  // (tag space: 0 NULL, 1 numeric, >= 2 string)
      }
      fa = rotl64(fa ^ mix64w(ca + 0x165667b19e3779f9ULL * (uint64_t)(c + 1)), 27) * 0x9fb21c651e98df25ULL + cb;
      fb = rotl64(fb ^ mix64w(cb ^ rotl64(ca, 32) ^ 0x27d4eb2f165667c5ULL), 31) * 0xd6e8feb86659fd93ULL + ca;
    }
    fa = mix64w(fa);
    fb = mix64w(fb ^ rotl64(fa, 17));
    if (fa == kEmptyKey) fa -= 1;
    if (fb == kEmptyKey) fb -= 1;
    n_valid += all_valid ? 1 : 0;
    int became_dup = 0;
    n_new += hash_insert128(t, fa, fb, d.want_multiplicity, 0, &became_dup);
    n_dup += became_dup;
  }
  block_add2w(n_new, n_dup, &counters[kCntDistinct], &counters[kCntTwice]);
  __syncthreads();
  block_add2w(n_valid, 0ull, &counters[kCntValidRows], &counters[kCntSpare]);
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
