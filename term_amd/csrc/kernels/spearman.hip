// spearman.hip -- Spearman rank-correlation state on gfx950.
//
// The reference computes it in SQL (TG/analyzers/advanced/correlation.rs:334-350):
//   WITH ranked AS (SELECT RANK() OVER (ORDER BY CAST(x AS DOUBLE)) rank_x, RANK() OVER (ORDER BY CAST(y AS DOUBLE)) rank_y
//                   FROM data WHERE x IS NOT NULL AND y IS NOT NULL)
//   SELECT COUNT(*), SUM(rank_x), SUM(rank_y), SUM(rank_x*rank_x), SUM(rank_y*rank_y), SUM(rank_x*rank_y)
// i.e. two global sorts with min-rank ties and UInt64 arithmetic that wraps (the sums of squares overflow past
// ~3.8 M rows).  Not mergeable (`:103-109`), so the state keeps the (x, y) pairs of every batch and ranks them at
// finalize.  Off the hot path (SURVEY.md section 8d times it separately).  The ranking itself is kernels/sortrank.hip
// (a sample sort that counts ranks bucket by bucket in LDS); this file holds the pair compaction and the pieces of
// the cross-rank ranking.
#include <hip/hip_runtime.h>

#include <cstring>

#include "device_types.h"
#include "sortrank.h"

namespace tgx {

typedef const int64_t __attribute__((address_space(1))) *global_i64_ptr;
typedef const uint8_t __attribute__((address_space(1))) *global_u8_ptr;

__device__ __forceinline__ uint64_t sort_key(int64_t bits, int is_float) {
  const double d = is_float ? __longlong_as_double(bits) : (double)bits;  // CAST(c AS DOUBLE)
  const int64_t k = f64_total_key(__double_as_longlong(d));
  return (uint64_t)k ^ 0x8000000000000000ULL;  // signed total order -> unsigned radix order
}

// appends the (x, y) sort keys of rows where both sides are non-NULL; *count is the running pair count.
// A workgroup takes 2048 consecutive rows per trip, eight per thread: one global atomic (and two barriers) per 2048
// rows -- with 256 rows per trip the kernel ran at 0.7 TB/s, bound by that atomic.  The pairs of a trip are placed by
// ballot: per step the valid lanes of a wave write NEIGHBOURING slots (a thread that placed its own eight pairs one
// behind the other made every store instruction touch 64 lines).
constexpr int kCompactRows = 8;
__global__ __launch_bounds__(256) void spearman_compact_kernel(ComomentColDesc d, uint64_t *kx, uint64_t *ky,
                                                                unsigned long long *count) {
  global_i64_ptr x = (global_i64_ptr)(uintptr_t)((const int64_t *)d.x + d.xoff);
  global_i64_ptr y = (global_i64_ptr)(uintptr_t)((const int64_t *)d.y + d.yoff);
  global_u8_ptr xv = (global_u8_ptr)(uintptr_t)d.xv;
  global_u8_ptr yv = (global_u8_ptr)(uintptr_t)d.yv;
  __shared__ unsigned long long block_base;
  __shared__ uint32_t wave_cnt[4];
  const uint32_t wave = threadIdx.x >> 6;
  constexpr int64_t kTrip = 256 * kCompactRows;
  const int64_t step = (int64_t)gridDim.x * kTrip;
  const int64_t rounded = (d.length + step - 1) / step * step;
  for (int64_t base = (int64_t)blockIdx.x * kTrip; base < rounded; base += step) {
    int64_t vx[kCompactRows], vy[kCompactRows];
    unsigned long long m[kCompactRows];
    uint32_t mine = 0;
#pragma unroll
    for (int u = 0; u < kCompactRows; u++) {
      const int64_t i = base + u * 256 + threadIdx.x;  // coalesced; the order of the pairs does not matter
      bool ok = i < d.length;
      if (ok && xv) ok = (xv[(d.xoff + i) >> 3] >> ((d.xoff + i) & 7)) & 1;
      if (ok && yv) ok = (yv[(d.yoff + i) >> 3] >> ((d.yoff + i) & 7)) & 1;
      vx[u] = i < d.length ? x[i] : 0;
      vy[u] = i < d.length ? y[i] : 0;
      m[u] = __builtin_amdgcn_ballot_w64(ok);
      mine += (uint32_t)__builtin_popcountll(m[u]);  // (the wave's count: the same in every lane)
    }
    if ((threadIdx.x & 63) == 0) wave_cnt[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
      block_base = total ? atomicAdd(count, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    uint64_t off = block_base;
    for (uint32_t w = 0; w < wave; w++) off += wave_cnt[w];
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int u = 0; u < kCompactRows; u++) {
      if ((m[u] >> lane) & 1ull) {
        const uint64_t at = off + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[u] >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)m[u], 0u));
        kx[at] = sort_key(vx[u], d.x_is_float);
        ky[at] = sort_key(vy[u], d.y_is_float);
      }
      off += (uint32_t)__builtin_popcountll(m[u]);
    }
    __syncthreads();
  }
}

// neither column has NULLs: every row is a pair, placed behind the pairs the state already holds (*count is bumped
// by the launch that follows: nobody changes it while this one reads it)
__global__ __launch_bounds__(256) void spearman_convert_kernel(ComomentColDesc d, uint64_t *kx, uint64_t *ky,
                                                                const unsigned long long *count) {
  const int64_t *x = (const int64_t *)d.x + d.xoff, *y = (const int64_t *)d.y + d.yoff;
  const unsigned long long base = *count;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.length; i += (int64_t)gridDim.x * 256) {
    kx[base + i] = sort_key(__builtin_nontemporal_load(x + i), d.x_is_float);
    ky[base + i] = sort_key(__builtin_nontemporal_load(y + i), d.y_is_float);
  }
}
__global__ void spearman_bump_kernel(unsigned long long *count, unsigned long long by) { *count += by; }

// ---- pieces of the cross-rank ranking (spearman_device.cpp, spearman_allreduce) ----
// `count` regular samples of a sorted array
__global__ void sample_sorted_kernel(const uint64_t *sorted, uint64_t n, uint32_t count, uint64_t *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = sorted[(uint64_t)i * n / count];
}
// out[j] = first position whose key is >= splitters[j]
__global__ void lower_bounds_kernel(const uint64_t *sorted, uint64_t n, const uint64_t *splitters, uint32_t k,
                                    uint64_t *out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  const uint64_t v = splitters[j];
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (sorted[mid] < v)
      lo = mid + 1;
    else
      hi = mid;
  }
  out[j] = lo;
}
// out[perm[k]] = vals[k]
__global__ void unsort_kernel(const uint64_t *vals, const uint32_t *perm, uint64_t n, uint64_t *out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    out[perm[i]] = vals[i];
}

__global__ __launch_bounds__(256) void rank_sums_kernel(const uint64_t *rx, const uint64_t *ry, uint64_t n,
                                                         uint64_t plus, RankSums *partials) {
  unsigned long long w[5] = {0, 0, 0, 0, 0};
  unsigned __int128 e[5] = {0, 0, 0, 0, 0};
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long a = rx[i] + plus, b = ry[i] + plus;  // (plus = 1: the arrays hold run starts)
    w[0] += a;
    w[1] += b;
    w[2] += a * a;
    w[3] += b * b;
    w[4] += a * b;
    e[0] += a;
    e[1] += b;
    e[2] += (unsigned __int128)a * a;
    e[3] += (unsigned __int128)b * b;
    e[4] += (unsigned __int128)a * b;
  }
  __shared__ RankSums sh[256 / 64];
  // wave reduce through shuffles of 64-bit halves
  for (int k = 0; k < 5; k++) {
    unsigned long long lo = (unsigned long long)e[k], hi = (unsigned long long)(e[k] >> 64), ww = w[k];
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1) {
      const unsigned long long olo = __shfl_down(lo, dlt, 64), ohi = __shfl_down(hi, dlt, 64);
      ww += __shfl_down(ww, dlt, 64);
      const unsigned long long s = lo + olo;
      hi += ohi + (s < lo ? 1 : 0);
      lo = s;
    }
    if ((threadIdx.x & 63) == 0) {
      sh[threadIdx.x >> 6].wrapped[k] = ww;
      sh[threadIdx.x >> 6].exact_lo[k] = lo;
      sh[threadIdx.x >> 6].exact_hi[k] = hi;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    RankSums r = sh[0];
    for (int wv = 1; wv < 4; wv++)
      for (int k = 0; k < 5; k++) {
        r.wrapped[k] += sh[wv].wrapped[k];
        const unsigned long long s = r.exact_lo[k] + sh[wv].exact_lo[k];
        r.exact_hi[k] += sh[wv].exact_hi[k] + (s < r.exact_lo[k] ? 1 : 0);
        r.exact_lo[k] = s;
      }
    partials[blockIdx.x] = r;
  }
}

static int grid_of(uint64_t n) {
  uint64_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

void launch_spearman_compact(const ComomentColDesc &d, uint64_t *kx, uint64_t *ky, unsigned long long *count,
                             hipStream_t stream) {
  if (!d.xv && !d.yv) {
    hipLaunchKernelGGL(spearman_convert_kernel, dim3(grid_of((uint64_t)d.length)), dim3(256), 0, stream, d, kx, ky, count);
    hipLaunchKernelGGL(spearman_bump_kernel, dim3(1), dim3(1), 0, stream, count, (unsigned long long)d.length);
    return;
  }
  hipLaunchKernelGGL(spearman_compact_kernel, dim3(grid_of((uint64_t)d.length)), dim3(256), 0, stream, d, kx, ky, count);
}

size_t spearman_rank_sums_bytes() { return sizeof(RankSums); }

void launch_sample_sorted(const uint64_t *sorted, uint64_t n, uint32_t count, uint64_t *out, hipStream_t stream) {
  hipLaunchKernelGGL(sample_sorted_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, sorted, n, count, out);
}
void launch_lower_bounds(const uint64_t *sorted, uint64_t n, const uint64_t *splitters, uint32_t k, uint64_t *out,
                         hipStream_t stream) {
  hipLaunchKernelGGL(lower_bounds_kernel, dim3((k + 63) / 64), dim3(64), 0, stream, sorted, n, splitters, k, out);
}
void launch_unsort(const uint64_t *vals, const uint32_t *perm, uint64_t n, uint64_t *out, hipStream_t stream) {
  hipLaunchKernelGGL(unsort_kernel, dim3(grid_of(n)), dim3(256), 0, stream, vals, perm, n, out);
}

int launch_rank_sums(const uint64_t *rx, const uint64_t *ry, uint64_t n, uint64_t plus, void *partials,
                     hipStream_t stream) {
  const int g = grid_of(n);
  hipLaunchKernelGGL(rank_sums_kernel, dim3(g), dim3(256), 0, stream, rx, ry, n, plus, (RankSums *)partials);
  return g;
}

}  // namespace tgx
