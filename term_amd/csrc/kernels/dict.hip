// dict.hip -- checks on Dictionary<Int32, Utf8> columns for gfx950.
//
// A dictionary column is (int32 indices, validity) over a small Utf8 dictionary.  String work happens once per
// dictionary ENTRY, the rows only contribute 4-byte indices:
//   pattern checks : the match kernel runs over the dictionary and records one hit byte per entry
//                    (regex.hip, `hits` output); dict_count_hits_kernel then sums hits[index[row]].
//   DISTINCT       : dict_fingerprint_kernel hashes every entry to the same 128-bit fingerprint a plain Utf8
//                    column would produce (so dictionary and plain batches of one column can be mixed and
//                    per-batch dictionaries need not agree); dict_usage_kernel counts how often each entry is
//                    referenced (saturating at 2, per-workgroup LDS histogram for small dictionaries);
//                    dict_insert_kernel inserts the fingerprints of referenced entries into the state's set.
#include <hip/hip_runtime.h>
#include <string.h>

#include "device_types.h"
#include "distinct_types.h"

namespace tgx {

typedef const uint8_t __attribute__((address_space(1))) *global_u8_ptr;
typedef const int32_t __attribute__((address_space(1))) *global_i32_ptr;


struct DictRowsDesc {
  const int32_t *indices;   // element 0 of the indices buffer
  const uint8_t *validity;  // of the rows, or nullptr
  int64_t offset;
  int64_t length;
  const uint8_t *dict_validity;  // of the dictionary VALUES, or nullptr
  int64_t dict_offset;      // Arrow offset of the dictionary array
  int64_t dict_length;
};

__device__ __forceinline__ void dict_block_add(unsigned long long a, unsigned long long *ga) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) a += __shfl_down(a, d, 64);
  __shared__ unsigned long long sa[4];
  if ((threadIdx.x & 63) == 0) sa[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = sa[0] + sa[1] + sa[2] + sa[3];
    if (t) atomicAdd(ga, t);
  }
}

// Streams the index column once: f(e, valid) for every row.  Rows are taken four at a time with one
// global_load_dwordx4 per lane (after a scalar head up to the first 16-byte aligned index), four loads in flight
// per lane; the four validity bits of a quad come from a 16-bit window of the bitmap.
template <int kQ = 4, class F>  // kQ: quads per lane and batch (8 measured the same as 4)
__device__ __forceinline__ void dict_for_each_row(const DictRowsDesc &d, F &&f) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  typedef const i32x4 __attribute__((address_space(1))) *global_i32x4_ptr;
  global_i32_ptr idx = (global_i32_ptr)(uintptr_t)(d.indices + d.offset);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t head = (int64_t)(((16 - ((uintptr_t)(d.indices + d.offset) & 15)) & 15) >> 2);
  if (head > d.length) head = d.length;
  const int64_t n_quads = (d.length - head) >> 2;
  auto one = [&](int64_t i) {
    bool valid = true;
    if (vbits) valid = (vbits[(d.offset + i) >> 3] >> ((d.offset + i) & 7)) & 1;
    f(idx[i], valid);
  };
  // scalar head and tail (at most 3 + 3 rows)
  if (tid < head) one(tid);
  if (tid < d.length - head - 4 * n_quads) one(head + 4 * n_quads + tid);
  global_i32x4_ptr quads = (global_i32x4_ptr)(idx + head);
  // software pipeline: the loads of the NEXT four quads are requested before the current ones are handed to f (one
  // workgroup per CU and f's LDS work in between: waiting for each batch on its own left the kernel at 3.1-3.7 TB/s)
  struct Batch {
    i32x4 v[kQ];
    uint32_t w[kQ];  // validity bytes covering the quad (low byte first), not shifted yet
  };
  auto issue = [&](int64_t q0, Batch &b) {
#pragma unroll
    for (int u = 0; u < kQ; u++) {
      const int64_t q = q0 + u * stride;
      const bool in = q < n_quads;
      b.w[u] = 0xFFFFu;
      if (in && vbits) {
        const int64_t bit = d.offset + head + 4 * q;
        uint32_t w = vbits[bit >> 3];
        if ((bit & 7) > 4) w |= (uint32_t)vbits[(bit >> 3) + 1] << 8;  // the quad straddles a byte
        b.w[u] = w;
      }
      b.v[u] = __builtin_nontemporal_load(quads + (in ? q : 0));
    }
  };
  auto consume = [&](int64_t q0, const Batch &b) {
#pragma unroll
    for (int u = 0; u < kQ; u++) {
      const int64_t q = q0 + u * stride;
      if (q >= n_quads) continue;
      const uint32_t bits = (b.w[u] >> ((d.offset + head + 4 * q) & 7)) & 0xFu;
      f(b.v[u].x, (bits & 1) != 0);
      f(b.v[u].y, (bits & 2) != 0);
      f(b.v[u].z, (bits & 4) != 0);
      f(b.v[u].w, (bits & 8) != 0);
    }
  };
  Batch cur;
  if (tid < n_quads) issue(tid, cur);
  for (int64_t q0 = tid; q0 < n_quads; q0 += kQ * stride) {
    Batch nxt;
    const int64_t qn = q0 + kQ * stride;
    if (qn < n_quads) issue(qn, nxt);
    consume(q0, cur);
    cur = nxt;
  }
}

constexpr int kDictLdsWords = 32768;  // 128 KiB of LDS bitmaps: 1 Mi entries (one bitmap) or 512 Ki (two)
constexpr int kDictLdsWordsSmall = 16384;  // dictionaries that fit half of that: two workgroups a CU
constexpr int kDictThreads = 1024;    // one workgroup per CU, 16 waves
constexpr int kDictMaxFused = 4;      // pattern checks that can ride on the DISTINCT pass of a column

__device__ __forceinline__ void dict_block_add1024(unsigned long long a, unsigned long long *ga) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) a += __shfl_down(a, d, 64);
  __shared__ unsigned long long sa[16];
  if ((threadIdx.x & 63) == 0) sa[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (uint32_t w = 0; w < blockDim.x / 64; w++) t += sa[w];
    if (t) atomicAdd(ga, t);
  }
}

// 32 per-entry verdict bytes (1 = match; 0 / 2 = not) -> one bitmap word.  `hits` is 16-byte aligned and padded by
// 32 bytes (regex_device.cpp), so the two 16-byte loads of the last word stay inside the allocation.
__device__ __forceinline__ uint32_t dict_pack_hits(const uint8_t *hits, uint32_t w, int64_t dict_length) {
  typedef unsigned long long __attribute__((ext_vector_type(2))) u64x2;
  const u64x2 *p = (const u64x2 *)(hits + 32ull * w);
  const u64x2 a = p[0], b = p[1];
  const unsigned long long ones = 0x0101010101010101ull, gather = 0x0102040810204080ull;
  uint32_t m = (uint32_t)(((a.x & ones) * gather) >> 56) | (uint32_t)(((a.y & ones) * gather) >> 56) << 8 |
               (uint32_t)(((b.x & ones) * gather) >> 56) << 16 | (uint32_t)(((b.y & ones) * gather) >> 56) << 24;
  const int64_t left = dict_length - 32ll * w;
  if (left < 32) m &= left <= 0 ? 0u : ((1u << left) - 1u);
  return m;
}

// matches += hits[index] for valid rows (hit byte: 1 = entry matches, 2 = entry is NULL), null_is_valid for NULL
// rows.  LDS: the per-entry verdicts as a bitmap in LDS (dictionaries without NULL values, <= 1 Mi entries), so
// the per-row lookup is an LDS read instead of a scattered global byte load.
template <int LDSW>  // words of LDS bitmap: 0 = none (verdict bytes from global memory), kDictLdsWordsSmall, kDictLdsWords
__global__ __launch_bounds__(kDictThreads) void dict_count_hits_kernel(DictRowsDesc d, const uint8_t *hits,
                                                                        int null_is_valid,
                                                                        unsigned long long *counters) {
  constexpr bool LDS = LDSW > 0;
  __shared__ uint32_t match_bits[LDS ? LDSW : 1];
  global_u8_ptr h = (global_u8_ptr)(uintptr_t)hits;
  if (LDS) {
    const int64_t words = (d.dict_length + 31) >> 5;
    for (int64_t w = threadIdx.x; w < words; w += kDictThreads)
      match_bits[w] = dict_pack_hits(hits, (uint32_t)w, d.dict_length);
    __syncthreads();
  }
  unsigned long long matches = 0;
  dict_for_each_row(d, [&](int32_t e, bool valid) {
    if (valid) {
      if (e < 0 || e >= d.dict_length) return;
      if (LDS) {
        matches += (match_bits[e >> 5] >> (e & 31)) & 1;
      } else {
        const uint8_t hv = h[e];
        matches += hv == 1 ? 1 : (hv == 2 && null_is_valid) ? 1 : 0;  // 2: the dictionary VALUE is NULL
      }
    } else {
      matches += null_is_valid ? 1 : 0;
    }
  });
  dict_block_add1024(matches, &counters[0]);
}

// Which dictionary entries do the valid rows reference, and which more than once?
// LDS = true : every workgroup keeps private seen / twice bitmaps in LDS and writes them to its slice of
//              `slices` ([workgroup][seen words | twice words]); launch_dict_usage then ORs the slices
//              (two workgroups that each saw an entry once make it "twice") into `seen` / `twice`.
// LDS = false: dictionaries beyond the LDS budget -- test-before-set atomicOr straight into `seen` / `twice`.
// counters[kCntValidRows] += valid rows.
template <int LDSW, bool MULT>
__global__ __launch_bounds__(kDictThreads) void dict_usage_kernel(DictRowsDesc d, uint32_t *seen, uint32_t *twice,
                                                                   uint32_t *slices, unsigned long long *counters) {
  constexpr bool LDS = LDSW > 0;
  __shared__ uint32_t bits[LDS ? LDSW : 1];
  const uint32_t words = (uint32_t)((d.dict_length + 31) >> 5);
  uint32_t *l_seen = bits, *l_twice = bits + words;
  if (LDS) {
    for (uint32_t w = threadIdx.x; w < (MULT ? 2 : 1) * words; w += kDictThreads) bits[w] = 0;
    __syncthreads();
  }
  global_u8_ptr dvbits = (global_u8_ptr)(uintptr_t)d.dict_validity;
  unsigned long long n_valid = 0;
  dict_for_each_row(d, [&](int32_t e, bool valid) {
    if (!valid) return;
    if (e < 0 || e >= d.dict_length) return;  // malformed index: ignored (Arrow validates these)
    if (dvbits && !((dvbits[(d.dict_offset + e) >> 3] >> ((d.dict_offset + e) & 7)) & 1)) return;  // NULL value
    n_valid++;
    const uint32_t bit = 1u << (e & 31), w = (uint32_t)e >> 5;
    uint32_t *s_ = LDS ? l_seen : seen, *t_ = LDS ? l_twice : twice;
    if (MULT) {
      if (!(t_[w] & bit)) {  // once an entry is known to be referenced twice there is nothing left to record
        if (s_[w] & bit) {
          atomicOr(&t_[w], bit);
        } else {
          const uint32_t prev = atomicOr(&s_[w], bit);
          if (prev & bit) atomicOr(&t_[w], bit);
        }
      }
    } else if (!(s_[w] & bit)) {
      atomicOr(&s_[w], bit);
    }
  });
  if (LDS) {
    __syncthreads();
    uint32_t *mine = slices + (size_t)blockIdx.x * (MULT ? 2 : 1) * words;
    for (uint32_t w = threadIdx.x; w < (MULT ? 2 : 1) * words; w += kDictThreads) mine[w] = bits[w];
  }
  dict_block_add1024(n_valid, &counters[kCntValidRows]);
}

// COUNT(*) / COUNT(col) of a dictionary column whose dictionary holds NULL values: a row counts when its index is
// valid AND the value it points at is
__global__ __launch_bounds__(kDictThreads) void dict_count_kernel(DictRowsDesc d, CountAcc *acc) {
  global_u8_ptr dvbits = (global_u8_ptr)(uintptr_t)d.dict_validity;
  unsigned long long n = 0;
  dict_for_each_row(d, [&](int32_t e, bool valid) {
    if (!valid || e < 0 || e >= d.dict_length) return;
    n += (dvbits[(d.dict_offset + e) >> 3] >> ((d.dict_offset + e) & 7)) & 1;
  });
  dict_block_add1024(n, (unsigned long long *)&acc->non_null);
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd((unsigned long long *)&acc->total, (unsigned long long)d.length);
}

// The fused row pass of a dictionary column that carries a DISTINCT check AND pattern / length checks: the indices
// are read ONCE; the lane updates the workgroup's seen / twice bitmaps and looks its entry up in the per-entry
// verdict bitmap of every pattern (all in LDS).  Per pattern, matches go to counters[k][0].
struct DictFuseParams {
  const uint8_t *hits[kDictMaxFused];         // per-entry verdict bytes (1 = match) of pattern k
  unsigned long long *counters[kDictMaxFused];
  int32_t null_is_valid[kDictMaxFused];
  int32_t k;
};

template <bool MULT, int LDSW>  // LDSW = kDictLdsWordsSmall: two workgroups a CU (the compiler keeps to 64 registers)
__global__ __launch_bounds__(kDictThreads) __attribute__((amdgpu_waves_per_eu(LDSW <= kDictLdsWordsSmall ? 8 : 4, LDSW <= kDictLdsWordsSmall ? 8 : 4))) void dict_fused_kernel(
    DictRowsDesc d, DictFuseParams f, uint32_t *slices, unsigned long long *counters) {
  __shared__ uint32_t bits[LDSW];
  const uint32_t words = (uint32_t)((d.dict_length + 31) >> 5);
  const uint32_t usage_words = (MULT ? 2u : 1u) * words;
  uint32_t *l_seen = bits, *l_twice = bits + words, *l_hits = bits + usage_words;
  for (uint32_t w = threadIdx.x; w < usage_words; w += kDictThreads) bits[w] = 0;
  for (int k = 0; k < f.k; k++) {
    for (uint32_t w = threadIdx.x; w < words; w += kDictThreads)
      l_hits[(uint32_t)k * words + w] = dict_pack_hits(f.hits[k], w, d.dict_length);
  }
  __syncthreads();
  uint32_t n_valid = 0, matches[kDictMaxFused] = {0, 0, 0, 0};  // (a lane sees far fewer than 2^32 rows)
  dict_for_each_row<(LDSW <= kDictLdsWordsSmall ? 2 : 4)>(d, [&](int32_t e, bool valid) {
    if (!valid) {
#pragma unroll
      for (int k = 0; k < kDictMaxFused; k++) matches[k] += (k < f.k && f.null_is_valid[k]) ? 1 : 0;
      return;
    }
    if (e < 0 || e >= d.dict_length) return;  // malformed index: ignored (Arrow validates these)
    n_valid++;
    const uint32_t bit = 1u << (e & 31), w = (uint32_t)e >> 5;
#pragma unroll
    for (int k = 0; k < kDictMaxFused; k++)
      if (k < f.k) matches[k] += (l_hits[(uint32_t)k * words + w] >> (e & 31)) & 1;
    if (MULT) {
      if (!(l_twice[w] & bit)) {
        if (l_seen[w] & bit) {
          atomicOr(&l_twice[w], bit);
        } else {
          const uint32_t prev = atomicOr(&l_seen[w], bit);
          if (prev & bit) atomicOr(&l_twice[w], bit);
        }
      }
    } else if (!(l_seen[w] & bit)) {
      atomicOr(&l_seen[w], bit);
    }
  });
  __syncthreads();
  uint32_t *mine = slices + (size_t)blockIdx.x * usage_words;
  for (uint32_t w = threadIdx.x; w < usage_words; w += kDictThreads) mine[w] = bits[w];
  dict_block_add1024(n_valid, &counters[kCntValidRows]);
  for (int k = 0; k < f.k; k++) {
    __syncthreads();
    dict_block_add1024(matches[k], &f.counters[k][0]);
  }
}

// OR of the workgroups' slices; an entry is referenced twice if any workgroup saw it twice or two saw it at all.
// A workgroup owns 64 words; its 16 waves split the slices (a single thread walking all 256 slices of a word was
// 60 us of dependent-latency per column, whatever the dictionary size), then combine through LDS.
constexpr int kReduceThreads = 1024;
__global__ __launch_bounds__(kReduceThreads) void dict_usage_reduce_kernel(const uint32_t *slices, uint32_t n_slices,
                                                                            uint32_t words, int mult, uint32_t *seen,
                                                                            uint32_t *twice) {
  __shared__ uint32_t part_seen[16][64], part_twice[16][64];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t w = blockIdx.x * 64 + lane;
  const uint32_t stride = (mult ? 2 : 1) * words;
  uint32_t acc_seen = 0, acc_twice = 0;
  if (w < words) {
#pragma unroll 4
    for (uint32_t s2 = wave; s2 < n_slices; s2 += 16) {
      const uint32_t x = slices[(size_t)s2 * stride + w];
      if (mult) acc_twice |= (acc_seen & x) | slices[(size_t)s2 * stride + words + w];
      acc_seen |= x;
    }
  }
  part_seen[wave][lane] = acc_seen;
  part_twice[wave][lane] = acc_twice;
  __syncthreads();
  if (wave == 0 && w < words) {
    for (int k = 1; k < 16; k++) {
      const uint32_t x = part_seen[k][lane];
      acc_twice |= (acc_seen & x) | part_twice[k][lane];
      acc_seen |= x;
    }
    seen[w] = acc_seen;
    if (mult) twice[w] = acc_twice;
  }
}
static int dict_grid(int64_t n, int n_cu, size_t lds_words = kDictLdsWords) {
  int64_t b = (n / 16 + kDictThreads - 1) / kDictThreads;  // 16 rows per thread per trip
  if (b < 1) b = 1;
  const int64_t resident = lds_words <= (size_t)kDictLdsWordsSmall ? 2 * (int64_t)n_cu : n_cu;  // 64 / 128 KiB of LDS
  if (b > resident) b = resident;
  return (int)b;
}

void launch_dict_count_hits(const int32_t *indices, const uint8_t *validity, int64_t offset, int64_t length,
                            int64_t dict_length, int dict_has_nulls, const uint8_t *hits, int null_is_valid,
                            unsigned long long *d_counters, int n_cu, hipStream_t stream) {
  DictRowsDesc d{indices, validity, offset, length, nullptr, 0, dict_length};
  const size_t words = (size_t)((dict_length + 31) >> 5);
  const int grid = dict_grid(length, n_cu, words);
  if (!dict_has_nulls && words <= (size_t)kDictLdsWordsSmall)
    hipLaunchKernelGGL(dict_count_hits_kernel<kDictLdsWordsSmall>, dim3(grid), dim3(kDictThreads), 0, stream, d, hits,
                       null_is_valid, d_counters);
  else if (!dict_has_nulls && words <= (size_t)kDictLdsWords)
    hipLaunchKernelGGL(dict_count_hits_kernel<kDictLdsWords>, dim3(grid), dim3(kDictThreads), 0, stream, d, hits,
                       null_is_valid, d_counters);
  else
    hipLaunchKernelGGL(dict_count_hits_kernel<0>, dim3(grid), dim3(kDictThreads), 0, stream, d, hits,
                       null_is_valid, d_counters);
}

void launch_dict_count(const int32_t *indices, const uint8_t *validity, int64_t offset, int64_t length,
                       const uint8_t *dict_validity, int64_t dict_offset, int64_t dict_length, CountAcc *acc, int n_cu,
                       hipStream_t stream) {
  DictRowsDesc d{indices, validity, offset, length, dict_validity, dict_offset, dict_length};
  hipLaunchKernelGGL(dict_count_kernel, dim3(dict_grid(length, n_cu)), dim3(kDictThreads), 0, stream, d, acc);
}

// how many pattern checks (<= 4) fit next to the usage bitmaps in the fused pass; 0: not fusable
int dict_fuse_capacity(int64_t length, int64_t dict_length, int want_mult, int n_cu) {
  const size_t words = (size_t)((dict_length + 31) >> 5);
  const size_t usage = words * (want_mult ? 2 : 1);
  if (words == 0 || usage >= (size_t)kDictLdsWords) return 0;
  const size_t k = ((size_t)kDictLdsWords - usage) / words;
  return (int)(k > (size_t)kDictMaxFused ? (size_t)kDictMaxFused : k);
}

// launch_dict_usage with up to 4 pattern gathers riding on the same pass (see dict_fused_kernel)
void launch_dict_usage_fused(const int32_t *indices, const uint8_t *validity, int64_t offset, int64_t length,
                             int64_t dict_length, int want_mult, int n_patterns, const uint8_t *const *hits,
                             unsigned long long *const *pattern_counters, const int32_t *null_is_valid, uint32_t *seen,
                             uint32_t *twice, uint32_t *scratch, unsigned long long *d_counters, int n_cu,
                             hipStream_t stream) {
  DictRowsDesc d{indices, validity, offset, length, nullptr, 0, dict_length};
  DictFuseParams f;
  memset(&f, 0, sizeof(f));
  f.k = n_patterns;
  for (int k = 0; k < n_patterns; k++) {
    f.hits[k] = hits[k];
    f.counters[k] = pattern_counters[k];
    f.null_is_valid[k] = null_is_valid[k];
  }
  const uint32_t words = (uint32_t)((dict_length + 31) >> 5);
  const size_t lds_words = (size_t)words * ((want_mult ? 2 : 1) + n_patterns);
  const bool small = lds_words <= (size_t)kDictLdsWordsSmall;
  // (the slices were sized by dict_usage_scratch_bytes from the usage words alone: never fewer workgroups than here)
  const int grid = dict_grid(length, n_cu, small ? (size_t)words * (want_mult ? 2 : 1) : (size_t)kDictLdsWords);
  const int grid_used = small ? grid : dict_grid(length, n_cu);
  const dim3 g(grid_used), b(kDictThreads);
  if (small) {
    if (want_mult)
      hipLaunchKernelGGL((dict_fused_kernel<true, kDictLdsWordsSmall>), g, b, 0, stream, d, f, scratch, d_counters);
    else
      hipLaunchKernelGGL((dict_fused_kernel<false, kDictLdsWordsSmall>), g, b, 0, stream, d, f, scratch, d_counters);
  } else {
    if (want_mult)
      hipLaunchKernelGGL((dict_fused_kernel<true, kDictLdsWords>), g, b, 0, stream, d, f, scratch, d_counters);
    else
      hipLaunchKernelGGL((dict_fused_kernel<false, kDictLdsWords>), g, b, 0, stream, d, f, scratch, d_counters);
  }
  hipLaunchKernelGGL(dict_usage_reduce_kernel, dim3((words + 63) / 64), dim3(kReduceThreads), 0, stream, scratch, (uint32_t)grid_used, words, want_mult, seen, twice);
}

size_t dict_usage_words(int64_t dict_length) { return (size_t)((dict_length + 31) >> 5); }

// scratch bytes launch_dict_usage needs for the per-workgroup slices (0: the global-atomics path is used)
size_t dict_usage_scratch_bytes(int64_t length, int64_t dict_length, int want_mult, int n_cu) {
  const size_t words = dict_usage_words(dict_length) * (want_mult ? 2 : 1);
  if (words > (size_t)kDictLdsWords) return 0;
  return (size_t)dict_grid(length, n_cu, words) * words * sizeof(uint32_t);
}

// seen / twice: bitmaps of dict_usage_words(dict_length) words each (twice only written with want_mult);
// the global-atomics path needs them zeroed by the caller
void launch_dict_usage(const int32_t *indices, const uint8_t *validity, int64_t offset, int64_t length,
                       const uint8_t *dict_validity, int64_t dict_offset, int64_t dict_length, int want_mult,
                       uint32_t *seen, uint32_t *twice, uint32_t *scratch, unsigned long long *d_counters, int n_cu,
                       hipStream_t stream) {
  DictRowsDesc d{indices, validity, offset, length, dict_validity, dict_offset, dict_length};
  const size_t lds_words = dict_usage_words(dict_length) * (want_mult ? 2 : 1);
  const bool lds = dict_usage_scratch_bytes(length, dict_length, want_mult, n_cu) != 0;
  const int grid = dict_grid(length, n_cu, lds ? lds_words : (size_t)kDictLdsWords);
  const dim3 g(grid), b(kDictThreads);
  if (lds && lds_words <= (size_t)kDictLdsWordsSmall) {
    if (want_mult)
      hipLaunchKernelGGL((dict_usage_kernel<kDictLdsWordsSmall, true>), g, b, 0, stream, d, seen, twice, scratch, d_counters);
    else
      hipLaunchKernelGGL((dict_usage_kernel<kDictLdsWordsSmall, false>), g, b, 0, stream, d, seen, twice, scratch, d_counters);
    const uint32_t words = (uint32_t)dict_usage_words(dict_length);
    hipLaunchKernelGGL(dict_usage_reduce_kernel, dim3((words + 63) / 64), dim3(kReduceThreads), 0, stream, scratch, (uint32_t)grid, words, want_mult, seen, twice);
  } else if (lds) {
    if (want_mult)
      hipLaunchKernelGGL((dict_usage_kernel<kDictLdsWords, true>), g, b, 0, stream, d, seen, twice, scratch, d_counters);
    else
      hipLaunchKernelGGL((dict_usage_kernel<kDictLdsWords, false>), g, b, 0, stream, d, seen, twice, scratch, d_counters);
    const uint32_t words = (uint32_t)dict_usage_words(dict_length);
    hipLaunchKernelGGL(dict_usage_reduce_kernel, dim3((words + 63) / 64), dim3(kReduceThreads), 0, stream, scratch, (uint32_t)grid, words, want_mult, seen, twice);
  } else {
    if (want_mult)
      hipLaunchKernelGGL((dict_usage_kernel<0, true>), g, b, 0, stream, d, seen, twice, scratch, d_counters);
    else
      hipLaunchKernelGGL((dict_usage_kernel<0, false>), g, b, 0, stream, d, seen, twice, scratch, d_counters);
  }
}

}  // namespace tgx

// the fingerprint + insert kernel lives with the 128-bit set (distinct128.hip) to share its helpers
