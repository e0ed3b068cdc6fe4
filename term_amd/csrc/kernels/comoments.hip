// comoments.hip -- co-moments of two numeric columns for gfx950.
//
//   n, SUM(x'), SUM(y'), SUM(x'*x'), SUM(y'*y'), SUM(x'*y')  over rows where both are non-NULL,
//   every value CAST AS DOUBLE, x' = x - px, y' = y - py for a per-pair pivot (px, py) picked near the data
//   (como_pivot_kernel).  The host rebuilds the raw sums of TG/analyzers/advanced/correlation.rs:239-249 and the
//   centred moments behind CORR / COVAR_SAMP (TG/constraints/correlation.rs:260-275) from them (tgx_finalize).
//
// HBM-bound (16 B + 2 validity bits per row).  One row per lane per load, 4 loads in flight per
// column per lane; sums are two-sum compensated per lane so the result does not depend on the
// grid shape beyond the last ulp; partials are folded in a fixed order by a second kernel.
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace tgx {

typedef const int64_t __attribute__((address_space(1))) *global_i64_ptr;
typedef const uint8_t __attribute__((address_space(1))) *global_u8_ptr;

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
                                                         ComomentPartial *__restrict__ partials,
                                                         const ComomentAcc *__restrict__ accs) {
  const ComomentColDesc d = L.pairs[blockIdx.y];
  const double px = accs[L.acc_index[blockIdx.y]].px, py = accs[L.acc_index[blockIdx.y]].py;
  global_i64_ptr x = (global_i64_ptr)(uintptr_t)((const int64_t *)d.x + d.xoff);
  global_i64_ptr y = (global_i64_ptr)(uintptr_t)((const int64_t *)d.y + d.yoff);
  global_u8_ptr xv = (global_u8_ptr)(uintptr_t)d.xv;
  global_u8_ptr yv = (global_u8_ptr)(uintptr_t)d.yv;
  double s[5] = {0, 0, 0, 0, 0}, c[5] = {0, 0, 0, 0, 0};
  int64_t n = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  auto fold = [&](int64_t xb, int64_t yb, bool ok) {
    double a = d.x_is_float ? __longlong_as_double(xb) : (double)xb;
    double b = d.y_is_float ? __longlong_as_double(yb) : (double)yb;
    a = ok ? a - px : 0.0;
    b = ok ? b - py : 0.0;
    n += ok ? 1 : 0;
    cm_two_sum(s[0], c[0], a);
    cm_two_sum(s[1], c[1], b);
    cm_two_sum(s[2], c[2], a * a);
    cm_two_sum(s[3], c[3], b * b);
    cm_two_sum(s[4], c[4], a * b);
  };
  const bool wide =
      ((((uintptr_t)((const int64_t *)d.x + d.xoff)) | ((uintptr_t)((const int64_t *)d.y + d.yoff))) & 15) == 0;
  int64_t done = 0;  // rows [0, done) are handled by the wide path
  if (wide) {
    // row pairs: one global_load_dwordx4 per column per pair, four pairs in flight per lane
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    typedef const i64x2 __attribute__((address_space(1))) *global_i64x2_ptr;
    global_i64x2_ptr x2 = (global_i64x2_ptr)x, y2 = (global_i64x2_ptr)y;
    const int64_t n_pairs = d.length >> 1;
    done = 2 * n_pairs;
    const bool x_even = (d.xoff & 1) == 0, y_even = (d.yoff & 1) == 0;
    for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p0 < n_pairs; p0 += 4 * stride) {
      i64x2 xq[4], yq[4];
      bool ok[8];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int64_t p = p0 + u * stride;
        const bool in = p < n_pairs;
        const int64_t q = in ? p : 0;
        // both rows of a pair share a validity byte when the Arrow offset is even: one byte load per column
        uint32_t xb2 = 3, yb2 = 3;
        if (xv) {
          const int64_t b = d.xoff + 2 * q;
          xb2 = x_even ? ((uint32_t)xv[b >> 3] >> (b & 7)) & 3u
                       : (uint32_t)cm_valid(xv, b) | ((uint32_t)cm_valid(xv, b + 1) << 1);
        }
        if (yv) {
          const int64_t b = d.yoff + 2 * q;
          yb2 = y_even ? ((uint32_t)yv[b >> 3] >> (b & 7)) & 3u
                       : (uint32_t)cm_valid(yv, b) | ((uint32_t)cm_valid(yv, b + 1) << 1);
        }
        ok[2 * u] = in && (xb2 & yb2 & 1u);
        ok[2 * u + 1] = in && ((xb2 & yb2) >> 1);
        xq[u] = __builtin_nontemporal_load(x2 + q);
        yq[u] = __builtin_nontemporal_load(y2 + q);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        fold(xq[u].x, yq[u].x, ok[2 * u]);
        fold(xq[u].y, yq[u].y, ok[2 * u + 1]);
      }
    }
  }
  for (int64_t i0 = done + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < d.length; i0 += 4 * stride) {
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
    for (int u = 0; u < 4; u++) fold(xb[u], yb[u], ok[u]);
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
  // fixed-order fold, 64 lanes wide: lane l folds its contiguous slice of the partials in order, lane 0 then folds
  // the 64 lane results in order -- the same association for a given grid shape, so results are reproducible
  __shared__ ComomentPartial lane_sum[64];
  const int lane = threadIdx.x;
  const int per = (blocks_per_pair + 63) / 64;
  ComomentPartial a;
  a.n = 0;
  for (int k = 0; k < 5; k++) a.s[k] = a.c[k] = 0.0;
  for (int i = lane * per; i < (lane + 1) * per && i < blocks_per_pair; i++) {
    const ComomentPartial &p = partials[(size_t)pair * blocks_per_pair + i];
    a.n += p.n;
    for (int k = 0; k < 5; k++) {
      a.c[k] += p.c[k];
      cm_two_sum(a.s[k], a.c[k], p.s[k]);
    }
  }
  lane_sum[lane] = a;
  __syncthreads();
  if (lane != 0) return;
  ComomentAcc &acc = accs[L.acc_index[pair]];
  acc.total += L.pairs[pair].length;
  for (int l = 0; l < 64; l++) {
    const ComomentPartial &p = lane_sum[l];
    acc.n += p.n;
    for (int k = 0; k < 5; k++) {
      acc.c[k] += p.c[k];
      cm_two_sum(acc.s[k], acc.c[k], p.s[k]);
    }
  }
}

// Picks the pivots of a pair: the means of (up to) 4096 evenly spread rows of the batch that have both values, finite.
// Any finite pivot gives the right answer; one near the data keeps S(x'x') - S(x')^2 / n well conditioned.  Pivots are
// fixed once rows have been folded in (acc.n > 0): every batch of a state is summed about the same pair.
__global__ __launch_bounds__(256) void como_pivot_kernel(const ComomentLaunch L, ComomentAcc *__restrict__ accs) {
  const ComomentColDesc d = L.pairs[blockIdx.x];
  ComomentAcc &acc = accs[L.acc_index[blockIdx.x]];
  if (acc.n > 0 || d.length <= 0) return;
  global_i64_ptr x = (global_i64_ptr)(uintptr_t)((const int64_t *)d.x + d.xoff);
  global_i64_ptr y = (global_i64_ptr)(uintptr_t)((const int64_t *)d.y + d.yoff);
  global_u8_ptr xv = (global_u8_ptr)(uintptr_t)d.xv;
  global_u8_ptr yv = (global_u8_ptr)(uintptr_t)d.yv;
  const int64_t samples = d.length < 4096 ? d.length : 4096;
  const int64_t step = d.length / samples;
  double sx = 0.0, sy = 0.0;
  int cnt = 0;
  for (int64_t k = threadIdx.x; k < samples; k += 256) {
    const int64_t i = k * step;
    if (!cm_valid(xv, d.xoff + i) || !cm_valid(yv, d.yoff + i)) continue;
    const double a = d.x_is_float ? __longlong_as_double(x[i]) : (double)x[i];
    const double b = d.y_is_float ? __longlong_as_double(y[i]) : (double)y[i];
    if (a - a != 0.0 || b - b != 0.0) continue;  // inf / NaN
    sx += a;
    sy += b;
    cnt++;
  }
  __shared__ double s_x[256], s_y[256];
  __shared__ int s_n[256];
  s_x[threadIdx.x] = sx;
  s_y[threadIdx.x] = sy;
  s_n[threadIdx.x] = cnt;
  __syncthreads();
  if (threadIdx.x != 0) return;
  double tx = 0.0, ty = 0.0;
  int k = 0;
  for (int i = 0; i < 256; i++) {
    tx += s_x[i];
    ty += s_y[i];
    k += s_n[i];
  }
  if (k == 0) return;
  tx /= (double)k;
  ty /= (double)k;
  if (tx - tx != 0.0 || ty - ty != 0.0) return;  // the sample's sum overflowed: stay with what is there
  acc.px = tx;
  acc.py = ty;
  acc.pivot_set = 1;
}

void launch_como_pivot(const ComomentLaunch &L, int n_pairs, ComomentAcc *d_accs, hipStream_t stream) {
  hipLaunchKernelGGL(como_pivot_kernel, dim3(n_pairs), dim3(256), 0, stream, L, d_accs);
}

size_t comoments_partial_bytes() { return sizeof(ComomentPartial); }

// the fold alone: scan_pair_kernel leaves the same per-block partials
void launch_comoments_reduce(const ComomentLaunch &L, int n_pairs, int blocks_per_pair, const void *d_partials,
                             ComomentAcc *d_accs, hipStream_t stream) {
  hipLaunchKernelGGL(comoments_reduce_kernel, dim3(n_pairs), dim3(64), 0, stream, L,
                     (const ComomentPartial *)d_partials, blocks_per_pair, d_accs);
}

void launch_comoments(const ComomentLaunch &L, int n_pairs, int blocks_per_pair, void *d_partials,
                      ComomentAcc *d_accs, hipStream_t stream) {
  hipLaunchKernelGGL(comoments_kernel, dim3(blocks_per_pair, n_pairs), dim3(256), 0, stream, L,
                     (ComomentPartial *)d_partials, (const ComomentAcc *)d_accs);
  hipLaunchKernelGGL(comoments_reduce_kernel, dim3(n_pairs), dim3(64), 0, stream, L,
                     (const ComomentPartial *)d_partials, blocks_per_pair, d_accs);
}

}  // namespace tgx
