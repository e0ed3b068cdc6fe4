// comoments.hip -- raw co-moments of two numeric columns for gfx950.
//
//   n, SUM(x), SUM(y), SUM(x*x), SUM(y*y), SUM(x*y)  over rows where both are non-NULL,
//   every value CAST AS DOUBLE                 (TG/analyzers/advanced/correlation.rs:239-249)
//
// HBM-bound (16 B + 2 validity bits per row).  One row per lane per load, 4 loads in flight per
// column per lane; sums are two-sum compensated per lane so the result does not depend on the
// grid shape beyond the last ulp; partials are folded in a fixed order by a second kernel.
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace tgx {

typedef const int64_t __attribute__((address_space(1))) *global_i64_ptr;
typedef const uint8_t __attribute__((address_space(1))) *global_u8_ptr;

struct ComomentPartial {
  int64_t n;
  double s[5], c[5];
};

__device__ __forceinline__ void cm_two_sum(double &s, double &c, double x) {
  double t = s + x;
  double bp = t - s;
  c += (s - (t - bp)) + (x - bp);
  s = t;
}

__device__ __forceinline__ bool cm_valid(global_u8_ptr v, int64_t bit) {
  return v == nullptr ? true : ((v[bit >> 3] >> (bit & 7)) & 1) != 0;
}

__global__ __launch_bounds__(256) void comoments_kernel(const ComomentLaunch L,
                                                         ComomentPartial *__restrict__ partials) {
  const ComomentColDesc d = L.pairs[blockIdx.y];
  global_i64_ptr x = (global_i64_ptr)(uintptr_t)((const int64_t *)d.x + d.xoff);
  global_i64_ptr y = (global_i64_ptr)(uintptr_t)((const int64_t *)d.y + d.yoff);
  global_u8_ptr xv = (global_u8_ptr)(uintptr_t)d.xv;
  global_u8_ptr yv = (global_u8_ptr)(uintptr_t)d.yv;
  double s[5] = {0, 0, 0, 0, 0}, c[5] = {0, 0, 0, 0, 0};
  int64_t n = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < d.length;
       i0 += 4 * stride) {
    int64_t xb[4], yb[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      int64_t i = i0 + u * stride;
      bool in = i < d.length;
      xb[u] = in ? x[i] : 0;
      yb[u] = in ? y[i] : 0;
      ok[u] = in && cm_valid(xv, d.xoff + (in ? i : 0)) && cm_valid(yv, d.yoff + (in ? i : 0));
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      double a = d.x_is_float ? __longlong_as_double(xb[u]) : (double)xb[u];
      double b = d.y_is_float ? __longlong_as_double(yb[u]) : (double)yb[u];
      a = ok[u] ? a : 0.0;
      b = ok[u] ? b : 0.0;
      n += ok[u] ? 1 : 0;
      cm_two_sum(s[0], c[0], a);
      cm_two_sum(s[1], c[1], b);
      cm_two_sum(s[2], c[2], a * a);
      cm_two_sum(s[3], c[3], b * b);
      cm_two_sum(s[4], c[4], a * b);
    }
  }
  // wave reduce
#pragma unroll
  for (int dlt = 32; dlt >= 1; dlt >>= 1) {
    n += __shfl_down(n, dlt, 64);
#pragma unroll
    for (int k = 0; k < 5; k++) {
      double os = __shfl_down(s[k], dlt, 64), oc = __shfl_down(c[k], dlt, 64);
      c[k] += oc;
      cm_two_sum(s[k], c[k], os);
    }
  }
  __shared__ ComomentPartial sh[kWavesPerBlock];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    sh[wave].n = n;
    for (int k = 0; k < 5; k++) {
      sh[wave].s[k] = s[k];
      sh[wave].c[k] = c[k];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    ComomentPartial r = sh[0];
    for (int w = 1; w < kWavesPerBlock; w++) {
      r.n += sh[w].n;
      for (int k = 0; k < 5; k++) {
        r.c[k] += sh[w].c[k];
        cm_two_sum(r.s[k], r.c[k], sh[w].s[k]);
      }
    }
    partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = r;
  }
}

__global__ __launch_bounds__(64) void comoments_reduce_kernel(
    const ComomentLaunch L, const ComomentPartial *__restrict__ partials, int blocks_per_pair,
    ComomentAcc *__restrict__ accs) {
  const int pair = blockIdx.x;
  if (threadIdx.x != 0) return;
  // serial fixed-order fold: <= 2048 partials, deterministic
  ComomentAcc &a = accs[L.acc_index[pair]];
  a.total += L.pairs[pair].length;
  for (int i = 0; i < blocks_per_pair; i++) {
    const ComomentPartial &p = partials[(size_t)pair * blocks_per_pair + i];
    a.n += p.n;
    for (int k = 0; k < 5; k++) {
      a.c[k] += p.c[k];
      cm_two_sum(a.s[k], a.c[k], p.s[k]);
    }
  }
}

size_t comoments_partial_bytes() { return sizeof(ComomentPartial); }

void launch_comoments(const ComomentLaunch &L, int n_pairs, int blocks_per_pair, void *d_partials,
                      ComomentAcc *d_accs, hipStream_t stream) {
  hipLaunchKernelGGL(comoments_kernel, dim3(blocks_per_pair, n_pairs), dim3(256), 0, stream, L,
                     (ComomentPartial *)d_partials);
  hipLaunchKernelGGL(comoments_reduce_kernel, dim3(n_pairs), dim3(64), 0, stream, L,
                     (const ComomentPartial *)d_partials, blocks_per_pair, d_accs);
}

}  // namespace tgx
