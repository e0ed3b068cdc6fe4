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

#include "distinct_types.h"

namespace tgx {

typedef const uint8_t __attribute__((address_space(1))) *global_u8_ptr;
typedef const int32_t __attribute__((address_space(1))) *global_i32_ptr;

constexpr int kDictLdsEntries = 16384;  // usage histogram in LDS (64 KiB) up to this dictionary size

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

// matches += hits[index] for valid rows (hit byte: 1 = entry matches, 2 = entry is NULL), null_is_valid for NULL rows
__global__ __launch_bounds__(256) void dict_count_hits_kernel(DictRowsDesc d, const uint8_t *hits,
                                                               int null_is_valid,
                                                               unsigned long long *counters) {
  global_i32_ptr idx = (global_i32_ptr)(uintptr_t)(d.indices + d.offset);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  global_u8_ptr h = (global_u8_ptr)(uintptr_t)hits;
  unsigned long long matches = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.length; i += (int64_t)gridDim.x * 256) {
    bool valid = true;
    if (vbits) valid = (vbits[(d.offset + i) >> 3] >> ((d.offset + i) & 7)) & 1;
    if (valid) {
      const int32_t e = idx[i];
      const uint8_t hv = (e >= 0 && e < d.dict_length) ? h[e] : 0;
      matches += hv == 1 ? 1 : (hv == 2 && null_is_valid) ? 1 : 0;  // 2: the dictionary VALUE is NULL
    } else {
      matches += null_is_valid ? 1 : 0;
    }
  }
  dict_block_add(matches, &counters[0]);
}

// usage[e] = min(2, number of valid rows referencing entry e); counters[kCntValidRows] += valid rows
__global__ __launch_bounds__(256) void dict_usage_kernel(DictRowsDesc d, uint32_t *usage,
                                                          unsigned long long *counters) {
  __shared__ uint32_t hist[kDictLdsEntries];
  const bool small = d.dict_length <= kDictLdsEntries;
  if (small) {
    for (int64_t e = threadIdx.x; e < d.dict_length; e += 256) hist[e] = 0;
    __syncthreads();
  }
  global_i32_ptr idx = (global_i32_ptr)(uintptr_t)(d.indices + d.offset);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  global_u8_ptr dvbits = (global_u8_ptr)(uintptr_t)d.dict_validity;
  unsigned long long n_valid = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.length; i += (int64_t)gridDim.x * 256) {
    bool valid = true;
    if (vbits) valid = (vbits[(d.offset + i) >> 3] >> ((d.offset + i) & 7)) & 1;
    if (!valid) continue;
    const int32_t e = idx[i];
    if (e < 0 || e >= d.dict_length) continue;  // malformed index: ignored (Arrow validates these)
    if (dvbits && !((dvbits[(d.dict_offset + e) >> 3] >> ((d.dict_offset + e) & 7)) & 1)) continue;  // NULL value
    n_valid++;
    if (small) {
      if (hist[e] < 2) atomicAdd(&hist[e], 1u);
    } else if (usage[e] < 2) {
      atomicAdd(&usage[e], 1u);
    }
  }
  if (small) {
    __syncthreads();
    for (int64_t e = threadIdx.x; e < d.dict_length; e += 256) {
      const uint32_t c = hist[e] > 2 ? 2 : hist[e];
      if (c && usage[e] < 2) atomicAdd(&usage[e], c);
    }
  }
  dict_block_add(n_valid, &counters[kCntValidRows]);
}

static int dict_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

void launch_dict_count_hits(const int32_t *indices, const uint8_t *validity, int64_t offset, int64_t length,
                            int64_t dict_length, const uint8_t *hits, int null_is_valid,
                            unsigned long long *d_counters, hipStream_t stream) {
  DictRowsDesc d{indices, validity, offset, length, nullptr, 0, dict_length};
  hipLaunchKernelGGL(dict_count_hits_kernel, dim3(dict_grid(length)), dim3(256), 0, stream, d, hits, null_is_valid,
                     d_counters);
}

void launch_dict_usage(const int32_t *indices, const uint8_t *validity, int64_t offset, int64_t length,
                       const uint8_t *dict_validity, int64_t dict_offset, int64_t dict_length, uint32_t *usage,
                       unsigned long long *d_counters, hipStream_t stream) {
  DictRowsDesc d{indices, validity, offset, length, dict_validity, dict_offset, dict_length};
  // the LDS histogram is flushed once per workgroup: fewer, fatter workgroups for small dictionaries
  int grid = dict_grid(length);
  if (dict_length <= kDictLdsEntries && grid > 512) grid = 512;
  hipLaunchKernelGGL(dict_usage_kernel, dim3(grid), dim3(256), 0, stream, d, usage, d_counters);
}

}  // namespace tgx

// the fingerprint + insert kernel lives with the 128-bit set (distinct128.hip) to share its helpers
