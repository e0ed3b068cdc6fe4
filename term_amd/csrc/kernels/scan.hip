// scan.hip -- fused numeric column scan for gfx950 (MI355X).
//
// One pass over (validity, values) of every numeric column of a batch produces
//   COUNT(col), MIN, MAX, SUM (exact 128-bit for Int64 / two-sum compensated for Float64),
//   and optionally the shifted moments for STDDEV / VARIANCE
// i.e. the aggregates behind term-guard's completeness / has_min / has_max / has_mean / has_sum
// checks (TG/constraints/completeness.rs:158-163, TG/constraints/statistics.rs:45-74).
//
// HBM-bound: 8 B of value + 1 bit of validity per row, nothing is re-read.  Layout of the hot loop:
//   * a wave consumes a 512-row tile per iteration = 4 x nontemporal global_load_dwordx4 per lane (4 KiB per
//     wave in flight, 16 B/lane coalesced),
//   * the tile's 8 validity words are wave-uniform and fetched through the scalar cache
//     (s_load), COUNT(col) is an s_bcnt1 on them -- no per-lane work for the null count,
//   * per-lane accumulators, DPP/shuffle wave reduce, LDS block reduce, one partial per block;
//     a second tiny kernel folds the partials in a fixed order (bitwise reproducible: no float
//     atomics, MI355X_MICROARCH.md "Global float atomics").
// grid = (blocks per column, columns): every column of the batch is covered by one launch.
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace tgx {

typedef long long i64x2 __attribute__((ext_vector_type(2)));
// constant address space: uniform addresses are fetched with s_load_* (scalar cache)
typedef const uint64_t __attribute__((address_space(4))) *const_u64_ptr;
// descriptor pointers are loaded from memory, so the compiler only knows them as flat pointers;
// these casts give global_load_* instead of flat_load_*
typedef const i64x2 __attribute__((address_space(1))) *global_i64x2_ptr;
typedef const int64_t __attribute__((address_space(1))) *global_i64_ptr;
typedef const uint8_t __attribute__((address_space(1))) *global_u8_ptr;
typedef const uint64_t __attribute__((address_space(1))) *global_u64_ptr;

struct LaneAcc {
  int64_t mn, mx;
  uint64_t lo;
  int64_t hi;
  double s, c;
  double s1, s2;
  int64_t cnt;  // rows counted per lane (ragged path only; tiles count with s_bcnt1)
};

__device__ __forceinline__ void acc_init(LaneAcc &a) {
  a.mn = INT64_MAX;
  a.mx = INT64_MIN;
  a.lo = 0;
  a.hi = 0;
  a.s = 0.0;
  a.c = 0.0;
  a.s1 = 0.0;
  a.s2 = 0.0;
  a.cnt = 0;
}

__device__ __forceinline__ void two_sum_add(double &s, double &c, double x) {
  // Knuth two-sum: s + x = t + e exactly; the error accumulates in c
  double t = s + x;
  double bp = t - s;
  double e = (s - (t - bp)) + (x - bp);
  s = t;
  c += e;
}

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef const i32x2 __attribute__((address_space(1))) *global_i32x2_ptr;
typedef const int32_t __attribute__((address_space(1))) *global_i32_ptr;

// an Int32 as the Int64 it stands for / the bits of a Float32 as the bits of the Float64 it stands for
template <bool IS_FLOAT>
__device__ __forceinline__ int64_t widen32(int32_t raw) {
  if (IS_FLOAT) return __double_as_longlong((double)__int_as_float(raw));
  return (int64_t)raw;
}

// The validity bit enters every select as a LANE MASK: ballot(valid) & ballot(compare) is one scalar AND, and
// inverse_ballot hands the result to v_cndmask as its mask operand.  Written as `(valid && x < mn) ? x : mn` the
// compiler selects twice (compare, then validity): 11 instead of 7 vector instructions per value for min + max, in a
// kernel that is 87 % VALU-busy next to 78-83 % of the HBM peak.
template <bool IS_FLOAT, bool VAR>
__device__ __forceinline__ void acc_value(LaneAcc &a, int64_t bits, bool valid, double pivot) {
  const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
  const bool is_valid = __builtin_amdgcn_inverse_ballot_w64(vm);
  if (IS_FLOAT) {
    const int64_t k = f64_total_key(bits);
    const bool lt = __builtin_amdgcn_inverse_ballot_w64(vm & __builtin_amdgcn_ballot_w64(k < a.mn));
    const bool gt = __builtin_amdgcn_inverse_ballot_w64(vm & __builtin_amdgcn_ballot_w64(k > a.mx));
    a.mn = lt ? k : a.mn;
    a.mx = gt ? k : a.mx;
    double x = is_valid ? __longlong_as_double(bits) : 0.0;
    two_sum_add(a.s, a.c, x);
    if (VAR) {
      double d = is_valid ? (__longlong_as_double(bits) - pivot) : 0.0;
      a.s1 += d;
      a.s2 += d * d;
    }
  } else {
    const bool lt = __builtin_amdgcn_inverse_ballot_w64(vm & __builtin_amdgcn_ballot_w64(bits < a.mn));
    const bool gt = __builtin_amdgcn_inverse_ballot_w64(vm & __builtin_amdgcn_ballot_w64(bits > a.mx));
    a.mn = lt ? bits : a.mn;
    a.mx = gt ? bits : a.mx;
    // the 128-bit sum is one add / add-with-carry chain
    const int64_t v = is_valid ? bits : 0;
    unsigned __int128 sum = ((unsigned __int128)a.hi << 64) | (unsigned __int128)a.lo;
    sum += (unsigned __int128)(__int128)v;
    a.lo = (uint64_t)sum;
    a.hi = (uint64_t)(sum >> 64);
    if (VAR) {
      double d = is_valid ? ((double)bits - pivot) : 0.0;
      a.s1 += d;
      a.s2 += d * d;
    }
  }
}

template <bool IS_FLOAT, bool VAR>
__device__ __forceinline__ void acc_pair(LaneAcc &a, i64x2 v, uint32_t two_bits, double pivot) {
  acc_value<IS_FLOAT, VAR>(a, v.x, (two_bits & 1u) != 0, pivot);
  acc_value<IS_FLOAT, VAR>(a, v.y, (two_bits & 2u) != 0, pivot);
}

// rows [r0, r1) one row per lane per step: ragged head / tail and columns whose buffers are not
// 16-byte / 64-bit aligned for the tile path.
template <bool IS_FLOAT, bool VAR>
__device__ void scan_ragged(const ScanColDesc &c, int64_t r0, int64_t r1, int lane, int stride,
                            LaneAcc &a, double pivot) {
  global_i64_ptr vals = (global_i64_ptr)(uintptr_t)((const int64_t *)c.values + c.offset);
  global_i32_ptr vals32 = (global_i32_ptr)(uintptr_t)((const int32_t *)c.values + c.offset);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)c.validity;
  for (int64_t i = r0 + lane; i < r1; i += stride) {
    bool valid = true;
    if (c.validity) {
      int64_t b = c.offset + i;
      valid = (vbits[b >> 3] >> (b & 7)) & 1;
    }
    const int64_t bits = c.elem32 ? widen32<IS_FLOAT>(vals32[i]) : vals[i];
    acc_value<IS_FLOAT, VAR>(a, bits, valid, pivot);
    a.cnt += valid ? 1 : 0;
  }
}

template <int VARIANT>
__device__ __forceinline__ i64x2 tile_load(global_i64x2_ptr p) {
  if (VARIANT & 1) return __builtin_nontemporal_load(p);
  return *p;
}

// 4-byte columns (Int32 / Date32 / Float32): the same pair of rows comes from ONE 8-byte load and is widened in
// registers -- the column is read once at 4 bytes per row instead of being widened through a staging buffer first
// (4.7 ms per 1 G-row column that way)
template <bool IS_FLOAT, int VARIANT>
__device__ __forceinline__ i64x2 tile_load32(global_i32x2_ptr p) {
  const i32x2 v = (VARIANT & 1) ? __builtin_nontemporal_load(p) : *p;
  i64x2 r;
  r.x = widen32<IS_FLOAT>(v.x);
  r.y = widen32<IS_FLOAT>(v.y);
  return r;
}

template <bool IS_FLOAT, bool VAR, int VARIANT>
__device__ __forceinline__ void scan_tiles(const ScanColDesc &c, int64_t wave_global,
                                           int64_t n_waves, int lane, LaneAcc &a,
                                           int64_t &tile_count, double pivot) {
  global_i64x2_ptr vp =
      (global_i64x2_ptr)(uintptr_t)((const int64_t *)c.values + c.offset + c.head);
  global_i32x2_ptr vp32 =
      (global_i32x2_ptr)(uintptr_t)((const int32_t *)c.values + c.offset + c.head);
  const bool e32 = c.elem32 != 0;  // uniform over the workgroup
  auto load = [&](int64_t pair) -> i64x2 {
    return e32 ? tile_load32<IS_FLOAT, VARIANT>(vp32 + pair) : tile_load<VARIANT>(vp + pair);
  };
  const bool has_validity = c.validity != nullptr;
  const_u64_ptr vw = (const_u64_ptr)(uintptr_t)(c.validity + ((c.offset + c.head) >> 3));
  const uint32_t sh = 2u * (uint32_t)(lane & 31);
  const bool upper = lane >= 32;
  int64_t cnt = 0;
  constexpr bool kPrefetch = (VARIANT & 2) != 0;
  i64x2 n0, n1, n2, n3;
  int64_t t = wave_global;
  if (kPrefetch && t < c.n_tiles) {
    const int64_t p = t * (kTileRows / 2) + lane;
    n0 = load(p);
    n1 = load(p + 64);
    n2 = load(p + 128);
    n3 = load(p + 192);
  }
  for (; t < c.n_tiles; t += n_waves) {
    i64x2 v0, v1, v2, v3;
    if (kPrefetch) {
      v0 = n0; v1 = n1; v2 = n2; v3 = n3;
      const int64_t tn = t + n_waves;
      if (tn < c.n_tiles) {
        const int64_t pn = tn * (kTileRows / 2) + lane;
        n0 = load(pn);
        n1 = load(pn + 64);
        n2 = load(pn + 128);
        n3 = load(pn + 192);
      }
    } else {
      const int64_t p = t * (kTileRows / 2) + lane;
      v0 = load(p);
      v1 = load(p + 64);
      v2 = load(p + 128);
      v3 = load(p + 192);
    }
    uint64_t w0 = ~0ull, w1 = ~0ull, w2 = ~0ull, w3 = ~0ull, w4 = ~0ull, w5 = ~0ull, w6 = ~0ull,
             w7 = ~0ull;
    if (has_validity) {
      const_u64_ptr q = vw + t * (kTileRows / 64);
      w0 = q[0]; w1 = q[1]; w2 = q[2]; w3 = q[3];
      w4 = q[4]; w5 = q[5]; w6 = q[6]; w7 = q[7];
      cnt += __builtin_popcountll(w0) + __builtin_popcountll(w1) + __builtin_popcountll(w2) +
             __builtin_popcountll(w3) + __builtin_popcountll(w4) + __builtin_popcountll(w5) +
             __builtin_popcountll(w6) + __builtin_popcountll(w7);
    } else {
      cnt += kTileRows;
    }
    acc_pair<IS_FLOAT, VAR>(a, v0, (uint32_t)((upper ? w1 : w0) >> sh) & 3u, pivot);
    acc_pair<IS_FLOAT, VAR>(a, v1, (uint32_t)((upper ? w3 : w2) >> sh) & 3u, pivot);
    acc_pair<IS_FLOAT, VAR>(a, v2, (uint32_t)((upper ? w5 : w4) >> sh) & 3u, pivot);
    acc_pair<IS_FLOAT, VAR>(a, v3, (uint32_t)((upper ? w7 : w6) >> sh) & 3u, pivot);
  }
  tile_count = cnt;
}

__device__ __forceinline__ int64_t shfl_down_i64(int64_t v, int d) {
  return __shfl_down(v, d, 64);
}

// merge b into a (both lanes' partials)
__device__ __forceinline__ void acc_merge(LaneAcc &a, const LaneAcc &b, bool is_float) {
  a.mn = b.mn < a.mn ? b.mn : a.mn;
  a.mx = b.mx > a.mx ? b.mx : a.mx;
  uint64_t lo = a.lo + b.lo;
  a.hi += b.hi + (lo < a.lo ? 1 : 0);
  a.lo = lo;
  double c = a.c + b.c;
  two_sum_add(a.s, c, b.s);
  a.c = c;
  a.s1 += b.s1;
  a.s2 += b.s2;
  a.cnt += b.cnt;
}

__device__ __forceinline__ LaneAcc acc_shfl_down(const LaneAcc &a, int d) {
  LaneAcc b;
  b.mn = shfl_down_i64(a.mn, d);
  b.mx = shfl_down_i64(a.mx, d);
  b.lo = (uint64_t)shfl_down_i64((int64_t)a.lo, d);
  b.hi = shfl_down_i64(a.hi, d);
  b.s = __shfl_down(a.s, d, 64);
  b.c = __shfl_down(a.c, d, 64);
  b.s1 = __shfl_down(a.s1, d, 64);
  b.s2 = __shfl_down(a.s2, d, 64);
  b.cnt = shfl_down_i64(a.cnt, d);
  return b;
}

// merges one batch's folded partial `a` of a column into its running state (Chan's pairwise merge for the moments)
__device__ __forceinline__ void scan_fold(ScanAcc &s, const LaneAcc &a, const ScanColDesc &c) {
  s.is_float = c.is_float;
  s.total += c.length;
  s.non_null += a.cnt;
  s.min_k = a.mn < s.min_k ? a.mn : s.min_k;
  s.max_k = a.mx > s.max_k ? a.mx : s.max_k;
  uint64_t lo = s.sum_lo + a.lo;
  s.sum_hi += a.hi + (lo < s.sum_lo ? 1 : 0);
  s.sum_lo = lo;
  double comp = s.comp + a.c;
  two_sum_add(s.sum, comp, a.s);
  s.comp = comp;
  if (c.want_variance && a.cnt > 0) {
    // batch moments about the pivot -> (n, mean, M2), then Chan's pairwise merge
    const double pivot = c.pivot ? *c.pivot : 0.0;
    const double nb = (double)a.cnt;
    const double mean_b = pivot + a.s1 / nb;
    double m2_b = a.s2 - a.s1 * a.s1 / nb;
    if (m2_b < 0.0) m2_b = 0.0;
    if (s.var_n == 0) {
      s.var_n = a.cnt;
      s.var_mean = mean_b;
      s.var_m2 = m2_b;
    } else {
      const double na = (double)s.var_n;
      const double delta = mean_b - s.var_mean;
      const double n = na + nb;
      s.var_mean = s.var_mean + delta * nb / n;
      s.var_m2 = s.var_m2 + m2_b + delta * delta * na * nb / n;
      s.var_n += a.cnt;
    }
  }
}

template <bool IS_FLOAT, bool VAR, int VARIANT>
__device__ __forceinline__ void scan_body(const ScanColDesc &c, ScanPartial *out, ScanAcc *direct, int wave,
                                          int lane) {
  LaneAcc a;
  acc_init(a);
  const double pivot = (VAR && c.pivot) ? *c.pivot : 0.0;
  const int64_t wave_global = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
  int64_t tile_count = 0;
  if (c.n_tiles > 0) {
    scan_tiles<IS_FLOAT, VAR, VARIANT>(c, wave_global, n_waves, lane, a, tile_count, pivot);
    // ragged edges belong to the last block (it has the least tile work when tiles % grid != 0)
    if (blockIdx.x == gridDim.x - 1) {
      const int64_t tail0 = c.head + c.n_tiles * kTileRows;
      if (wave == 0) scan_ragged<IS_FLOAT, VAR>(c, 0, c.head, lane, 64, a, pivot);
      if (wave == 1) scan_ragged<IS_FLOAT, VAR>(c, tail0, c.length, lane, 64, a, pivot);
    }
  } else {
    // unaligned or short column: every wave strides over rows
    scan_ragged<IS_FLOAT, VAR>(c, wave_global * 64, c.length, lane, (int)(n_waves * 64), a,
                               pivot);
  }
  // wave reduce
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    LaneAcc b = acc_shfl_down(a, d);
    acc_merge(a, b, IS_FLOAT);
  }
  __shared__ LaneAcc s_acc[kWavesPerBlock];
  __shared__ int64_t s_cnt[kWavesPerBlock];
  if (lane == 0) {
    s_acc[wave] = a;
    s_cnt[wave] = tile_count;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    LaneAcc r = s_acc[0];
    int64_t cnt = s_cnt[0];
    for (int w = 1; w < kWavesPerBlock; w++) {
      acc_merge(r, s_acc[w], IS_FLOAT);
      cnt += s_cnt[w];
    }
    if (direct) {
      // one workgroup per column (small batches): fold straight into the running state -- no reduce launch.  The
      // reduce kernel would do exactly this with the one partial (merging identities is a no-op), so the results
      // are the same bit for bit.
      r.cnt += cnt;
      scan_fold(*direct, r, c);
      return;
    }
    ScanPartial p;
    p.non_null = cnt + r.cnt;
    p.min_k = r.mn;
    p.max_k = r.mx;
    p.sum_lo = r.lo;
    p.sum_hi = r.hi;
    p.sum = r.s;
    p.comp = r.c;
    p.s1 = r.s1;
    p.s2 = r.s2;
    *out = p;
  }
}

template <int VARIANT>
__global__ __launch_bounds__(kScanBlock) void scan_kernel(const ScanLaunch L,
                                                           ScanPartial *__restrict__ partials,
                                                           ScanAcc *__restrict__ accs) {
  const ScanColDesc c = L.cols[blockIdx.y];
  ScanAcc *direct = (accs && gridDim.x == 1) ? accs + L.acc_index[blockIdx.y] : nullptr;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  ScanPartial *out = partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (c.is_float) {
    if (c.want_variance)
      scan_body<true, true, VARIANT>(c, out, direct, wave, lane);
    else
      scan_body<true, false, VARIANT>(c, out, direct, wave, lane);
  } else {
    if (c.want_variance)
      scan_body<false, true, VARIANT>(c, out, direct, wave, lane);
    else
      scan_body<false, false, VARIANT>(c, out, direct, wave, lane);
  }
}

// Folds the per-block partials of each column (fixed order => bitwise reproducible) and merges
// the batch into the running per-column state.  grid = columns, block = 64 (one wave).
__global__ __launch_bounds__(64) void scan_reduce_kernel(const ScanLaunch L,
                                                          const ScanPartial *__restrict__ partials,
                                                          int blocks_per_col,
                                                          ScanAcc *__restrict__ accs) {
  const int col = blockIdx.x;
  const int lane = threadIdx.x;
  const ScanColDesc c = L.cols[col];
  const ScanPartial *p = partials + (size_t)col * blocks_per_col;
  LaneAcc a;
  acc_init(a);
  for (int i = lane; i < blocks_per_col; i += 64) {
    LaneAcc b;
    b.mn = p[i].min_k;
    b.mx = p[i].max_k;
    b.lo = p[i].sum_lo;
    b.hi = p[i].sum_hi;
    b.s = p[i].sum;
    b.c = p[i].comp;
    b.s1 = p[i].s1;
    b.s2 = p[i].s2;
    b.cnt = p[i].non_null;
    acc_merge(a, b, c.is_float);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    LaneAcc b = acc_shfl_down(a, d);
    acc_merge(a, b, c.is_float);
  }
  if (lane == 0) scan_fold(accs[L.acc_index[col]], a, c);
}

// Picks the variance pivot of a column: the mean of (up to) the first 256 valid values of the first
// batch.  Any finite pivot gives the right answer; one near the data keeps s2 - s1^2/n well
// conditioned.  A pivot that is already set (later batches) is kept.
__global__ __launch_bounds__(256) void scan_pivot_kernel(const ScanLaunch L,
                                                          double *__restrict__ pivots,
                                                          int32_t *__restrict__ pivot_set) {
  const ScanColDesc c = L.cols[blockIdx.x];
  if (!c.want_variance) return;
  const int slot = L.acc_index[blockIdx.x];
  if (pivot_set[slot]) return;
  __shared__ double s_sum[256];
  __shared__ int s_cnt[256];
  const int64_t n = c.length < 4096 ? c.length : 4096;
  double sum = 0.0;
  int cnt = 0;
  global_i64_ptr vals = (global_i64_ptr)(uintptr_t)((const int64_t *)c.values + c.offset);
  global_i32_ptr vals32 = (global_i32_ptr)(uintptr_t)((const int32_t *)c.values + c.offset);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)c.validity;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    bool valid = true;
    if (c.validity) {
      int64_t b = c.offset + i;
      valid = (vbits[b >> 3] >> (b & 7)) & 1;
    }
    if (!valid) continue;
    double x;
    if (c.elem32)
      x = c.is_float ? (double)__int_as_float(vals32[i]) : (double)vals32[i];
    else
      x = c.is_float ? __longlong_as_double(vals[i]) : (double)vals[i];
    if (x - x != 0.0) continue;  // skip inf / nan
    sum += x;
    cnt++;
  }
  s_sum[threadIdx.x] = sum;
  s_cnt[threadIdx.x] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    int k = 0;
    for (int i = 0; i < 256; i++) {
      t += s_sum[i];
      k += s_cnt[i];
    }
    if (k > 0) {
      pivots[slot] = t / (double)k;
      pivot_set[slot] = 1;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// COUNT(*) / COUNT(col) for columns whose values are not needed: popcount of the validity bits.
// grid = (blocks, columns).  Reads ceil(n/8) bytes per column.
__global__ __launch_bounds__(256) void count_kernel(const CountLaunch L,
                                                     unsigned long long *__restrict__ block_counts,
                                                     CountAcc *__restrict__ direct) {
  const CountColDesc c = L.cols[blockIdx.y];
  const int64_t bit0 = c.offset, bit1 = c.offset + c.length;
  // words of the 8-byte aligned bitmap view that intersect [bit0, bit1)
  const uintptr_t base = (uintptr_t)c.validity;
  const int64_t mis = (int64_t)(base & 7);  // bytes the buffer starts past an 8-byte boundary
  global_u64_ptr words = (global_u64_ptr)(base - mis);
  const int64_t b0 = bit0 + mis * 8, b1 = bit1 + mis * 8;
  const int64_t w0 = b0 >> 6, w1 = (b1 + 63) >> 6;
  int64_t cnt = 0;
  for (int64_t w = w0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < w1;
       w += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = words[w];
    if (w == w0) x &= ~0ull << (b0 & 63);
    if (w == w1 - 1 && (b1 & 63)) x &= ~0ull >> (64 - (b1 & 63));
    cnt += __builtin_popcountll(x);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cnt += shfl_down_i64(cnt, d);
  __shared__ int64_t s[kWavesPerBlock];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int64_t total = s[0] + s[1] + s[2] + s[3];
    if (direct && gridDim.x == 1) {  // one workgroup per column: no reduce launch (small batches)
      CountAcc &a = direct[L.acc_index[blockIdx.y]];
      a.total += c.length;
      a.non_null += total;
    } else {
      block_counts[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (unsigned long long)total;
    }
  }
}

__global__ __launch_bounds__(64) void count_reduce_kernel(const CountLaunch L,
                                                           const unsigned long long *__restrict__ bc,
                                                           int blocks_per_col,
                                                           CountAcc *__restrict__ accs) {
  const int col = blockIdx.x;
  int64_t cnt = 0;
  for (int i = threadIdx.x; i < blocks_per_col; i += 64)
    cnt += (int64_t)bc[(size_t)col * blocks_per_col + i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cnt += shfl_down_i64(cnt, d);
  if (threadIdx.x == 0) {
    CountAcc &a = accs[L.acc_index[col]];
    a.total += L.cols[col].length;
    a.non_null += cnt;
  }
}

// ---------------------------------------------------------------------------------------------
// Int32 -> Int64 / Float32 -> Float64 over the rows a batch views (TGX_INT32 / TGX_FLOAT32 columns): four values per
// lane and trip (one 16-byte load when the source is 16-byte aligned, two 16-byte stores).
__global__ __launch_bounds__(256) void widen32_kernel(const void *__restrict__ src, void *__restrict__ dst, int64_t n,
                                                      int is_float) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool aligned = (((uintptr_t)src) & 15) == 0;
  const int64_t n4 = aligned ? n >> 2 : 0;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
    if (is_float) {
      const float4 v = ((const float4 *)src)[q];
      double2 *o = (double2 *)dst + 2 * q;
      o[0] = make_double2((double)v.x, (double)v.y);
      o[1] = make_double2((double)v.z, (double)v.w);
    } else {
      const int4 v = ((const int4 *)src)[q];
      longlong2 *o = (longlong2 *)dst + 2 * q;
      o[0] = make_longlong2((long long)v.x, (long long)v.y);
      o[1] = make_longlong2((long long)v.z, (long long)v.w);
    }
  }
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (is_float)
      ((double *)dst)[i] = (double)((const float *)src)[i];
    else
      ((long long *)dst)[i] = (long long)((const int *)src)[i];
  }
}

void launch_widen32(const void *src, void *dst, int64_t n, int is_float, int n_cu, hipStream_t stream) {
  if (n <= 0) return;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > (int64_t)n_cu * 16) blocks = (int64_t)n_cu * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(widen32_kernel, dim3((int)blocks), dim3(256), 0, stream, src, dst, n, is_float);
}

// ---------------------------------------------------------------------------------------------
// host launchers (called from tgx_api.cpp); `n_cols` <= kMaxColsPerLaunch
void launch_scan_pivot(const ScanLaunch &L, int n_cols, double *d_pivots, int32_t *d_pivot_set,
                       hipStream_t stream) {
  hipLaunchKernelGGL(scan_pivot_kernel, dim3(n_cols), dim3(256), 0, stream, L, d_pivots, d_pivot_set);
}

// d_accs != nullptr and blocks_per_col == 1: the kernel folds into the running states itself (no reduce launch)
void launch_scan_main_only(const ScanLaunch &L, int n_cols, int blocks_per_col, ScanPartial *d_partials,
                           ScanAcc *d_accs, hipStream_t stream) {
  // VARIANT bit 0 = nontemporal loads, bit 1 = next tile requested before the current one is consumed.
  // Measured at 1 G rows x 16 columns (ms per launch): 0: 21.11, 1: 20.95, 2: 21.24, 3: 20.74 -> 3.
  hipLaunchKernelGGL(scan_kernel<3>, dim3(blocks_per_col, n_cols), dim3(kScanBlock), 0, stream, L, d_partials,
                     blocks_per_col == 1 ? d_accs : nullptr);
}

void launch_scan_reduce_only(const ScanLaunch &L, int n_cols, int blocks_per_col, ScanPartial *d_partials,
                             ScanAcc *d_accs, hipStream_t stream) {
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(n_cols), dim3(64), 0, stream, L, d_partials, blocks_per_col,
                     d_accs);
}

void launch_count(const CountLaunch &L, int n_cols, int blocks_per_col, unsigned long long *d_block_counts,
                  CountAcc *d_accs, hipStream_t stream) {
  hipLaunchKernelGGL(count_kernel, dim3(blocks_per_col, n_cols), dim3(256), 0, stream, L, d_block_counts,
                     blocks_per_col == 1 ? d_accs : nullptr);
  if (blocks_per_col > 1)
    hipLaunchKernelGGL(count_reduce_kernel, dim3(n_cols), dim3(64), 0, stream, L, d_block_counts,
                       blocks_per_col, d_accs);
}

}  // namespace tgx
