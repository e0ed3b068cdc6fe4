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
#include "distinct_types.h"

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

// ---------------------------------------------------------------------------------------------
// HyperLogLog lane (scan_hll_kernel): the workgroup's 2^14 one-byte registers live in 16 KiB of LDS.  A register only
// ever grows, and after the first few thousand values of a workgroup almost no value raises its register any more: a
// value READS its register (ds_read_u8) and only a value that would raise it enters the compare-and-swap on the word
// that holds it -- LDS has no byte atomics, and a 32-bit max over packed bytes is not a byte-wise max.
__device__ __forceinline__ void hll_rank(int64_t bits, bool valid, uint32_t &idx, uint32_t &rho) {
  uint32_t a, b;
  hll_hash((uint64_t)bits, &a, &b);
  idx = a & (uint32_t)(kHllRegisters - 1);
  rho = (uint32_t)__builtin_clz(b | 1u) + (b ? 1u : 2u);  // clz(b) + 1; b == 0 -> 33
  rho = valid ? rho : 0u;                                  // a NULL raises nothing
}
__device__ __forceinline__ void hll_raise(uint32_t *regs, uint32_t idx, uint32_t rho) {
  const uint32_t sh = (idx & 3u) * 8u;
  uint32_t *word = regs + (idx >> 2);
  uint32_t old = *word;
  while (((old >> sh) & 0xFFu) < rho) {
    const uint32_t want = (old & ~(0xFFu << sh)) | (rho << sh);
    const uint32_t seen = atomicCAS(word, old, want);
    if (seen == old) break;
    old = seen;
  }
}
__device__ __forceinline__ void hll_update(uint32_t *regs, int64_t bits, bool valid) {
  uint32_t idx, rho;
  hll_rank(bits, valid, idx, rho);
  if (((const uint8_t *)regs)[idx] < rho) hll_raise(regs, idx, rho);
}
// a tile's eight values of a lane together: the eight hash chains interleave (a 32-bit multiply is a quarter-rate
// instruction with a long latency), the eight register reads are in flight at once and waited for once, and the
// compare-and-swap path is entered only by a wave in which some value does raise its register -- value by value,
// each behind its own read and its own branch, the lane ran at LDS latency: 2.0 ms per 1 G-row column
__device__ __forceinline__ void hll_update8(uint32_t *regs, const int64_t (&bits)[8], const bool (&valid)[8]) {
  uint32_t idx[8], rho[8], cur[8];
#pragma unroll
  for (int k = 0; k < 8; k++) hll_rank(bits[k], valid[k], idx[k], rho[k]);
#pragma unroll
  for (int k = 0; k < 8; k++) cur[k] = ((const uint8_t *)regs)[idx[k]];
  bool need = false;
#pragma unroll
  for (int k = 0; k < 8; k++) need |= cur[k] < rho[k];
  if (__builtin_amdgcn_ballot_w64(need) != 0) {  // (uniform)
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (cur[k] < rho[k]) hll_raise(regs, idx[k], rho[k]);
  }
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
template <bool IS_FLOAT, bool VAR, bool HLL = false>
__device__ void scan_ragged(const ScanColDesc &c, int64_t r0, int64_t r1, int lane, int stride,
                            LaneAcc &a, double pivot, uint32_t *hll = nullptr) {
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
    if (!HLL || !c.skip_stats) acc_value<IS_FLOAT, VAR>(a, bits, valid, pivot);
    a.cnt += valid ? 1 : 0;
    if (HLL) hll_update(hll, bits, valid);
  }
}

// ---------------------------------------------------------------------------------------------
// The KLL sampler riding on the scan (kll_types.h, ScanKll): every wave keeps the values of its own row stream that
// have not been grouped yet in a private LDS ring -- at most 2^top - 1 pending ones plus the 512 a tile brings --
// (the ring wraps: nothing is ever moved) and, whenever 2^top of them are complete, keeps one member of the group
// chosen by a counter hash.  Per value this
// costs a ballot, an mbcnt and one LDS store next to the column's other accumulators; the column crosses HBM once.
struct KllLane {
  double mn, mx;           // NaN-ignoring MIN / MAX (KllSketch::update drops NaN, kll_sketch.rs:197-199)
  uint32_t head;           // ring slot of the oldest pending value, wave-uniform
  uint32_t pending;        // values in the ring, wave-uniform
  uint32_t groups;         // groups emitted so far, wave-uniform
  uint32_t rsize;          // slots of the ring: 2^top + kTileRows (pending < 2^top before a tile, < rsize after it)
};

__device__ __forceinline__ void kll_lane_init(KllLane &K, int top = 0) {
  K.mn = __longlong_as_double(0x7FF0000000000000LL);
  K.mx = -K.mn;
  K.head = 0;
  K.pending = 0;
  K.groups = 0;
  K.rsize = (1u << top) + (uint32_t)kTileRows;
}

// the ring wraps: slot of the i-th pending value (i < rsize, head < rsize: one conditional subtraction)
__device__ __forceinline__ uint32_t kll_slot(const KllLane &K, uint32_t i) {
  const uint32_t p = K.head + i;
  return p >= K.rsize ? p - K.rsize : p;
}

__device__ __forceinline__ uint64_t scan_mix(uint64_t x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}

// LDS traffic between the lanes of ONE wave: program order is enough for the hardware (a wave's LDS instructions
// execute in order), the fence keeps the compiler from moving the accesses across it
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// one value per lane: `ok` = non-NULL and not NaN.  TRACK = false: the caller keeps the MIN / MAX itself (the pair
// kernel's extreme filter)
template <bool TRACK = true>
__device__ __forceinline__ void kll_push(KllLane &K, double *ring, double x, bool ok) {
  const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
  const uint32_t pos = kll_slot(K, K.pending + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)));
  if (ok) ring[pos] = x;
  K.pending += (uint32_t)__builtin_popcountll(m);
  if (TRACK) {
    const double xn = ok ? x : __longlong_as_double(0x7FF8000000000000LL);
    K.mn = __builtin_fmin(K.mn, xn);  // v_min_f64 / v_max_f64 return the other operand for a NaN
    K.mx = __builtin_fmax(K.mx, xn);
  }
}

// emits the complete groups of the ring; the rest stays where it is (the ring wraps: nothing is moved)
__device__ __forceinline__ void kll_drain(KllLane &K, double *ring, const ScanKll &q, uint32_t wave_slot, int lane) {
  const uint32_t top = (uint32_t)q.top;
  const uint32_t ng = K.pending >> top;
  if (ng == 0) return;  // uniform
  wave_lds_fence();
  double *out = q.picks + (size_t)wave_slot * (size_t)q.cap;
  for (uint32_t g0 = 0; g0 < ng; g0 += 64) {
    const uint32_t g = g0 + (uint32_t)lane;
    if (g < ng && K.groups + g < (uint32_t)q.cap) {
      const uint64_t h = scan_mix(q.salt ^ ((uint64_t)wave_slot << 36) ^ (uint64_t)(K.groups + g));
      out[K.groups + g] = ring[kll_slot(K, (g << top) + ((uint32_t)(h >> 20) & ((1u << top) - 1u)))];
    }
  }
  wave_lds_fence();  // (the slots of the emitted groups are free for the next tile's values from here on)
  K.head = kll_slot(K, ng << top);
  K.groups += ng;
  K.pending &= (1u << top) - 1u;
}

// end of the wave's stream: the pending values leave as weight-1 leftovers, unused slots as NaN
__device__ __forceinline__ void kll_finish_wave(KllLane &K, double *ring, const ScanKll &q, uint32_t wave_slot, int lane) {
  kll_drain(K, ring, q, wave_slot, lane);
  wave_lds_fence();
  const double nan = __longlong_as_double(0x7FF8000000000000LL);
  double *left = q.left + ((size_t)wave_slot << q.top);
  for (uint32_t i = (uint32_t)lane; i < (1u << q.top); i += 64) left[i] = i < K.pending ? ring[kll_slot(K, i)] : nan;
  double *out = q.picks + (size_t)wave_slot * (size_t)q.cap;
  for (uint32_t g = K.groups + (uint32_t)lane; g < (uint32_t)q.cap; g += 64) out[g] = nan;
  double mn = K.mn, mx = K.mx;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    mn = __builtin_fmin(mn, __shfl_down(mn, d, 64));
    mx = __builtin_fmax(mx, __shfl_down(mx, d, 64));
  }
  if (lane == 0) {
    KllWaveMeta m;
    m.count = ((unsigned long long)K.groups << q.top) + K.pending;
    m.min_v = mn;
    m.max_v = mx;
    q.meta[wave_slot] = m;
  }
}

// the value a column hands its sketch: doubles as they are, Int64 cast (KllSketch takes f64)
template <bool IS_FLOAT>
__device__ __forceinline__ double kll_value(int64_t bits) {
  return IS_FLOAT ? __longlong_as_double(bits) : (double)bits;
}

template <bool IS_FLOAT>
__device__ __forceinline__ void kll_push_pair(KllLane &K, double *ring, i64x2 v, uint32_t two_bits) {
  const double a = kll_value<IS_FLOAT>(v.x), b = kll_value<IS_FLOAT>(v.y);
  kll_push(K, ring, a, (two_bits & 1u) != 0 && a == a);
  kll_push(K, ring, b, (two_bits & 2u) != 0 && b == b);
}

// the same rows with the KLL sampler on: whole waves step together (the ring is a wave's), 64 rows per step
template <bool IS_FLOAT, bool VAR>
__device__ void scan_ragged_kll(const ScanColDesc &c, int64_t r0, int64_t r1, int lane, int stride, LaneAcc &a,
                                double pivot, KllLane &K, double *ring, uint32_t wave_slot) {
  global_i64_ptr vals = (global_i64_ptr)(uintptr_t)((const int64_t *)c.values + c.offset);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)c.validity;
  for (int64_t i0 = r0; i0 < r1; i0 += stride) {  // uniform trip count
    const int64_t i = i0 + lane;
    const bool in = i < r1;
    bool valid = in;
    if (in && c.validity) {
      int64_t b = c.offset + i;
      valid = (vbits[b >> 3] >> (b & 7)) & 1;
    }
    const int64_t bits = in ? vals[i] : 0;
    acc_value<IS_FLOAT, VAR>(a, bits, valid, pivot);
    a.cnt += valid ? 1 : 0;
    const double x = kll_value<IS_FLOAT>(bits);
    kll_push(K, ring, x, valid && x == x);
    kll_drain(K, ring, c.kll, wave_slot, lane);
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

template <bool IS_FLOAT, bool VAR, int VARIANT, bool KLL = false, bool HLL = false>
__device__ __forceinline__ void scan_tiles(const ScanColDesc &c, int64_t wave_global,
                                           int64_t n_waves, int lane, LaneAcc &a,
                                           int64_t &tile_count, double pivot, KllLane *K = nullptr,
                                           double *ring = nullptr, uint32_t wave_slot = 0, uint32_t *hll = nullptr) {
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
    const uint32_t b0 = (uint32_t)((upper ? w1 : w0) >> sh) & 3u, b1 = (uint32_t)((upper ? w3 : w2) >> sh) & 3u;
    const uint32_t b2 = (uint32_t)((upper ? w5 : w4) >> sh) & 3u, b3 = (uint32_t)((upper ? w7 : w6) >> sh) & 3u;
    if (!HLL || !c.skip_stats) {  // (uniform)
      acc_pair<IS_FLOAT, VAR>(a, v0, b0, pivot);
      acc_pair<IS_FLOAT, VAR>(a, v1, b1, pivot);
      acc_pair<IS_FLOAT, VAR>(a, v2, b2, pivot);
      acc_pair<IS_FLOAT, VAR>(a, v3, b3, pivot);
    }
    if (HLL) {
      const int64_t hb[8] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y, v3.x, v3.y};
      const bool hv[8] = {(b0 & 1u) != 0, (b0 & 2u) != 0, (b1 & 1u) != 0, (b1 & 2u) != 0,
                          (b2 & 1u) != 0, (b2 & 2u) != 0, (b3 & 1u) != 0, (b3 & 2u) != 0};
      hll_update8(hll, hb, hv);
    }
    if (KLL) {
      kll_push_pair<IS_FLOAT>(*K, ring, v0, b0);
      kll_push_pair<IS_FLOAT>(*K, ring, v1, b1);
      kll_push_pair<IS_FLOAT>(*K, ring, v2, b2);
      kll_push_pair<IS_FLOAT>(*K, ring, v3, b3);
      kll_drain(*K, ring, c.kll, wave_slot, lane);
    }
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

extern __shared__ double scan_dyn_lds[];  // KLL rings: one per (wave, sampled column) of the workgroup

template <bool IS_FLOAT, bool VAR, int VARIANT, bool KLL = false, bool HLL = false>
__device__ __forceinline__ void scan_body(const ScanColDesc &c, ScanPartial *out, ScanAcc *direct, int wave,
                                          int lane, uint32_t *hll = nullptr) {
  LaneAcc a;
  acc_init(a);
  const double pivot = (VAR && c.pivot) ? *c.pivot : 0.0;
  const int64_t wave_global = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
  int64_t tile_count = 0;
  KllLane K;
  double *ring = nullptr;
  if (KLL) {
    kll_lane_init(K, c.kll.top);
    ring = scan_dyn_lds + (size_t)wave * ((1u << c.kll.top) + kTileRows);
  }
  const uint32_t wave_slot = (uint32_t)wave_global;
  if (c.n_tiles > 0) {
    scan_tiles<IS_FLOAT, VAR, VARIANT, KLL, HLL>(c, wave_global, n_waves, lane, a, tile_count, pivot, &K, ring, wave_slot,
                                                 hll);
    // ragged edges belong to the last block (it has the least tile work when tiles % grid != 0)
    if (blockIdx.x == gridDim.x - 1) {
      const int64_t tail0 = c.head + c.n_tiles * kTileRows;
      if (KLL) {
        if (wave == 0) scan_ragged_kll<IS_FLOAT, VAR>(c, 0, c.head, lane, 64, a, pivot, K, ring, wave_slot);
        if (wave == 1) scan_ragged_kll<IS_FLOAT, VAR>(c, tail0, c.length, lane, 64, a, pivot, K, ring, wave_slot);
      } else {
        if (wave == 0) scan_ragged<IS_FLOAT, VAR, HLL>(c, 0, c.head, lane, 64, a, pivot, hll);
        if (wave == 1) scan_ragged<IS_FLOAT, VAR, HLL>(c, tail0, c.length, lane, 64, a, pivot, hll);
      }
    }
  } else if (KLL) {
    scan_ragged_kll<IS_FLOAT, VAR>(c, wave_global * 64, c.length, lane, (int)(n_waves * 64), a, pivot, K, ring,
                                   wave_slot);
  } else {
    // unaligned or short column: every wave strides over rows
    scan_ragged<IS_FLOAT, VAR, HLL>(c, wave_global * 64, c.length, lane, (int)(n_waves * 64), a, pivot, hll);
  }
  if (KLL) kll_finish_wave(K, ring, c.kll, wave_slot, lane);
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


// the same columns with their KLL samplers on (dynamic LDS: one ring per wave); never the small-batch direct fold
template <int VARIANT>
__global__ __launch_bounds__(kScanBlock) void scan_kll_kernel(const ScanLaunch L, ScanPartial *__restrict__ partials) {
  const ScanColDesc c = L.cols[blockIdx.y];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  ScanPartial *out = partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (c.is_float) {
    if (c.want_variance)
      scan_body<true, true, VARIANT, true>(c, out, nullptr, wave, lane);
    else
      scan_body<true, false, VARIANT, true>(c, out, nullptr, wave, lane);
  } else {
    if (c.want_variance)
      scan_body<false, true, VARIANT, true>(c, out, nullptr, wave, lane);
    else
      scan_body<false, false, VARIANT, true>(c, out, nullptr, wave, lane);
  }
}

// the same columns with the HyperLogLog lane on (APPROX_DISTINCT): the workgroup's registers in 16 KiB of LDS, written
// out as one row of the column's [workgroup][register] bytes; hll_reduce_kernel folds the rows into the running
// registers with a byte-wise max (a HyperLogLog merge).  skip_stats: the column's MIN / MAX / SUM were not asked for.
template <int VARIANT>
__global__ __launch_bounds__(kScanBlock) void scan_hll_kernel(const ScanLaunch L, ScanPartial *__restrict__ partials) {
  __shared__ uint32_t s_regs[kHllRegisters / 4];
  const ScanColDesc c = L.cols[blockIdx.y];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < kHllRegisters / 4; i += kScanBlock) s_regs[i] = 0;
  __syncthreads();
  ScanPartial *out = partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (c.is_float)
    scan_body<true, false, VARIANT, false, true>(c, out, nullptr, wave, lane, s_regs);
  else
    scan_body<false, false, VARIANT, false, true>(c, out, nullptr, wave, lane, s_regs);
  __syncthreads();
  uint4 *row = (uint4 *)(c.hll + (size_t)blockIdx.x * kHllRegisters);
  for (int i = threadIdx.x; i < kHllRegisters / 16; i += kScanBlock) row[i] = ((const uint4 *)s_regs)[i];
}

// byte-wise max of two words of four registers: the even and the odd bytes as two pairs of 16-bit lanes
__device__ __forceinline__ uint32_t hll_max4(uint32_t a, uint32_t b) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  union U {
    uint32_t w;
    u16x2 v;
  };
  U ae, be, ao, bo;
  ae.w = a & 0x00FF00FFu;
  be.w = b & 0x00FF00FFu;
  ao.w = (a >> 8) & 0x00FF00FFu;
  bo.w = (b >> 8) & 0x00FF00FFu;
  U e, o;
  e.v = __builtin_elementwise_max(ae.v, be.v);
  o.v = __builtin_elementwise_max(ao.v, bo.v);
  return e.w | (o.w << 8);
}

// grid = (16 word blocks x slices, columns): thread t owns four registers of its column over ONE slice of the
// workgroups' rows (eight loads in flight), and folds its word into the running registers with a compare-and-swap.
// (One thread per word walking all ~1500 rows one dependent load after the other took 0.52 ms of a 2.1 ms lane.)
__global__ __launch_bounds__(256) void hll_reduce_kernel(const ScanLaunch L, int blocks_per_col, int rows_per_slice) {
  const ScanColDesc c = L.cols[blockIdx.y];
  constexpr int kWordBlocks = kHllRegisters / 4 / 256;
  const int w = (int)(blockIdx.x % kWordBlocks) * 256 + (int)threadIdx.x;  // word of four registers
  const int b0 = (int)(blockIdx.x / kWordBlocks) * rows_per_slice;
  const int b1 = b0 + rows_per_slice < blocks_per_col ? b0 + rows_per_slice : blocks_per_col;
  const uint32_t *rows = (const uint32_t *)c.hll;
  uint32_t m = 0;
  int b = b0;
  for (; b + 8 <= b1; b += 8) {
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = rows[(size_t)(b + k) * (kHllRegisters / 4) + w];
#pragma unroll
    for (int k = 0; k < 8; k++) m = hll_max4(m, v[k]);
  }
  for (; b < b1; b++) m = hll_max4(m, rows[(size_t)b * (kHllRegisters / 4) + w]);
  if (m == 0) return;
  uint32_t *dst = (uint32_t *)c.hll_regs + w;
  uint32_t old = *dst;
  for (;;) {
    const uint32_t want = hll_max4(old, m);
    if (want == old) break;
    const uint32_t got = atomicCAS(dst, old, want);
    if (got == old) break;
    old = got;
  }
}

// ---------------------------------------------------------------------------------------------
// Two columns per workgroup (ScanPairDesc): each column's own aggregates exactly as scan_body computes them, the
// co-moments of the pair about its pivots over the rows where both are non-NULL (device_types.h, ComomentAcc;
// TG/analyzers/advanced/correlation.rs:239-249: every value CAST AS DOUBLE), and either column's KLL sampler -- a correlation check next to range / quantile checks on
// the same columns reads them once.
struct ComoLane {
  double s[5], c[5];
};

__device__ __forceinline__ void como_fold(ComoLane &m, double a, double b, bool both) {
  const unsigned long long bm = __builtin_amdgcn_ballot_w64(both);
  const bool ok = __builtin_amdgcn_inverse_ballot_w64(bm);
  a = ok ? a : 0.0;
  b = ok ? b : 0.0;
  two_sum_add(m.s[0], m.c[0], a);
  two_sum_add(m.s[1], m.c[1], b);
  two_sum_add(m.s[2], m.c[2], a * a);
  two_sum_add(m.s[3], m.c[3], b * b);
  two_sum_add(m.s[4], m.c[4], a * b);
}

// The tile form: a lane's eight rows of a tile are summed with plain adds (seven roundings of a partial sum no larger
// than eight terms: a relative error of a few 1e-16 of the partial), the partial then enters the running sum through
// the two-sum.  The kernel is bound by instruction issue, and 5 x 7 operations per row for the compensation were a
// third of them; the result stays far inside the 1e-6 the aggregates are held to (it moves in the 15th digit with
// the tiling, not with the grid).
struct ComoTile {
  double t[5];
};
__device__ __forceinline__ void como_tile_add(ComoTile &q, double a, double b, bool both) {
  const unsigned long long bm = __builtin_amdgcn_ballot_w64(both);
  const bool ok = __builtin_amdgcn_inverse_ballot_w64(bm);
  a = ok ? a : 0.0;
  b = ok ? b : 0.0;
  q.t[0] += a;
  q.t[1] += b;
  // (the file is built with -ffp-contract=off: the fused form is asked for by name -- three instructions fewer per
  //  row in an issue-bound loop, and the product enters the sum unrounded)
  q.t[2] = __builtin_fma(a, a, q.t[2]);
  q.t[3] = __builtin_fma(b, b, q.t[3]);
  q.t[4] = __builtin_fma(a, b, q.t[4]);
}
__device__ __forceinline__ void como_tile_flush(ComoLane &m, ComoTile &q) {
#pragma unroll
  for (int k = 0; k < 5; k++) {
    two_sum_add(m.s[k], m.c[k], q.t[k]);
    q.t[k] = 0.0;
  }
}

// ---- the pair kernel's tile path.  It is bound by instruction issue, not by HBM (two columns' aggregates, the
// co-moments and up to two samplers per row), so the per-value work is cut to what a value can change:
//   * MIN / MAX: a value matters only if it lies outside what the WAVE has seen so far.  The wave keeps uniform
//     bounds; a value is tested against them with two compares, and only a tile in which some lane holds a value
//     outside them (after the first few tiles: almost none) runs the exact update -- IEEE totalOrder keys for
//     doubles, as the single-column scan does -- and refreshes the bounds.  Doubles: NaN fails both compares and a
//     bound that is a zero is nudged one denormal inward, so NaN and -0 / +0 always take the exact path.
//   * SUM of a Float64 column: a lane's eight rows of a tile are added plainly, the partial enters the running sum
//     through the two-sum (as the co-moments do, see ComoTile).
//   * the samplers' NaN-ignoring MIN / MAX come out of the same exact path (KllLane::mn / mx are not touched per row).
template <bool F>
struct ColFast {
  int64_t lo, hi;   // wave-uniform filter bounds: !F the Int64 values, F the bit patterns of the (nudged) doubles
  double dmn, dmx;  // per lane: NaN-ignoring MIN / MAX over the values of the tiles that took the exact path
  double tsum;      // F: plain partial sum of the tile
};

template <bool F>
__device__ __forceinline__ void col_fast_init(ColFast<F> &c) {
  c.dmn = __longlong_as_double(0x7FF0000000000000LL);
  c.dmx = -c.dmn;
  c.tsum = 0.0;
  if (F) {
    c.lo = 0x7FF0000000000000LL;                   // +inf: nothing is inside yet
    c.hi = (int64_t)0xFFF0000000000000ULL;         // -inf
  } else {
    c.lo = INT64_MAX;
    c.hi = INT64_MIN;
  }
}

// the per-row part: range test (returns the lanes whose value lies outside the wave's bounds) and the sum
template <bool F>
__device__ __forceinline__ unsigned long long col_fast_value(ColFast<F> &c, LaneAcc &a, int64_t bits, bool valid) {
  bool inside;
  if (F) {
    const double x = __longlong_as_double(bits);
    inside = x >= __longlong_as_double(c.lo) && x <= __longlong_as_double(c.hi);
    c.tsum += valid ? x : 0.0;
  } else {
    inside = bits >= c.lo && bits <= c.hi;
    const int64_t v = valid ? bits : 0;
    unsigned __int128 sum = ((unsigned __int128)a.hi << 64) | (unsigned __int128)a.lo;
    sum += (unsigned __int128)(__int128)v;
    a.lo = (uint64_t)sum;
    a.hi = (uint64_t)(sum >> 64);
  }
  return __builtin_amdgcn_ballot_w64(valid && !inside);
}

// the exact MIN / MAX update of one value (what acc_value does, without the sums)
template <bool F>
__device__ __forceinline__ void col_exact_value(ColFast<F> &c, LaneAcc &a, int64_t bits, bool valid) {
  const int64_t k = F ? f64_total_key(bits) : bits;
  a.mn = (valid && k < a.mn) ? k : a.mn;
  a.mx = (valid && k > a.mx) ? k : a.mx;
  const double x = F ? __longlong_as_double(bits) : (double)bits;
  const double xn = valid ? x : __longlong_as_double(0x7FF8000000000000LL);
  c.dmn = __builtin_fmin(c.dmn, xn);
  c.dmx = __builtin_fmax(c.dmx, xn);
}

// new bounds from the lanes' exact state
template <bool F>
__device__ __forceinline__ void col_refresh_bounds(ColFast<F> &c, const LaneAcc &a) {
  if (F) {
    double lo = c.dmn, hi = c.dmx;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      lo = __builtin_fmin(lo, __shfl_xor(lo, d, 64));
      hi = __builtin_fmax(hi, __shfl_xor(hi, d, 64));
    }
    // a zero bound moves one denormal inward: -0 / +0 then always take the exact path (totalOrder tells them apart)
    if (lo == 0.0) lo = __longlong_as_double(1LL);
    if (hi == 0.0) hi = __longlong_as_double((long long)0x8000000000000001ULL);
    const long long lb = __double_as_longlong(lo), hb = __double_as_longlong(hi);
    c.lo = ((int64_t)__builtin_amdgcn_readfirstlane((int)(lb >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)lb);
    c.hi = ((int64_t)__builtin_amdgcn_readfirstlane((int)(hb >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)hb);
  } else {
    int64_t lo = a.mn, hi = a.mx;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const int64_t ol = __shfl_xor(lo, d, 64), oh = __shfl_xor(hi, d, 64);
      lo = ol < lo ? ol : lo;
      hi = oh > hi ? oh : hi;
    }
    c.lo = ((int64_t)__builtin_amdgcn_readfirstlane((int)(lo >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
    c.hi = ((int64_t)__builtin_amdgcn_readfirstlane((int)(hi >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)hi);
  }
}

template <bool XF, bool YF, bool KLL>
__device__ __forceinline__ void pair_rows(const ScanPairDesc &P, LaneAcc &ax, LaneAcc &ay, ColFast<XF> &fx,
                                          ColFast<YF> &fy, unsigned long long &out_x, unsigned long long &out_y,
                                          ComoTile &m, i64x2 vx, i64x2 vy, uint32_t bx, uint32_t by, KllLane &Kx,
                                          KllLane &Ky, double *rx, double *ry, double pvx, double pvy) {
  out_x |= col_fast_value<XF>(fx, ax, vx.x, (bx & 1u) != 0);
  out_x |= col_fast_value<XF>(fx, ax, vx.y, (bx & 2u) != 0);
  out_y |= col_fast_value<YF>(fy, ay, vy.x, (by & 1u) != 0);
  out_y |= col_fast_value<YF>(fy, ay, vy.y, (by & 2u) != 0);
  const double x0 = kll_value<XF>(vx.x), x1 = kll_value<XF>(vx.y);
  const double y0 = kll_value<YF>(vy.x), y1 = kll_value<YF>(vy.y);
  como_tile_add(m, x0 - pvx, y0 - pvy, (bx & by & 1u) != 0);  // about the pair's pivot (device_types.h, ComomentAcc)
  como_tile_add(m, x1 - pvx, y1 - pvy, (bx & by & 2u) != 0);
  if (KLL) {
    if (P.x.kll.picks) {  // uniform
      kll_push<false>(Kx, rx, x0, (bx & 1u) != 0 && x0 == x0);
      kll_push<false>(Kx, rx, x1, (bx & 2u) != 0 && x1 == x1);
    }
    if (P.y.kll.picks) {
      kll_push<false>(Ky, ry, y0, (by & 1u) != 0 && y0 == y0);
      kll_push<false>(Ky, ry, y1, (by & 2u) != 0 && y1 == y1);
    }
  }
}

// (Out of line -- its code and live ranges next to the hot loop's -- this ran the kernel at two waves per SIMD
// instead of three: 22.6 ms instead of 20.3 ms for the C4 scan.  It stays inline.)
template <bool F>
__device__ __forceinline__ void col_exact_tile(ColFast<F> &c, LaneAcc &a, i64x2 v0, i64x2 v1, i64x2 v2, i64x2 v3,
                                               uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {
  col_exact_value<F>(c, a, v0.x, (b0 & 1u) != 0);
  col_exact_value<F>(c, a, v0.y, (b0 & 2u) != 0);
  col_exact_value<F>(c, a, v1.x, (b1 & 1u) != 0);
  col_exact_value<F>(c, a, v1.y, (b1 & 2u) != 0);
  col_exact_value<F>(c, a, v2.x, (b2 & 1u) != 0);
  col_exact_value<F>(c, a, v2.y, (b2 & 2u) != 0);
  col_exact_value<F>(c, a, v3.x, (b3 & 1u) != 0);
  col_exact_value<F>(c, a, v3.y, (b3 & 2u) != 0);
  col_refresh_bounds<F>(c, a);
}

// one row per lane: ragged edges and pairs whose buffers do not allow tiles
template <bool XF, bool YF, bool KLL>
__device__ void pair_ragged(const ScanPairDesc &P, int64_t r0, int64_t r1, int lane, int stride, LaneAcc &ax,
                            LaneAcc &ay, ComoLane &m, int64_t &n_both, KllLane &Kx, KllLane &Ky, double *rx, double *ry,
                            uint32_t wave_slot, double pvx, double pvy) {
  global_i64_ptr xs = (global_i64_ptr)(uintptr_t)((const int64_t *)P.x.values + P.x.offset);
  global_i64_ptr ys = (global_i64_ptr)(uintptr_t)((const int64_t *)P.y.values + P.y.offset);
  global_u8_ptr xv = (global_u8_ptr)(uintptr_t)P.x.validity, yv = (global_u8_ptr)(uintptr_t)P.y.validity;
  for (int64_t i0 = r0; i0 < r1; i0 += stride) {  // uniform trip count: the KLL rings are a wave's
    const int64_t i = i0 + lane;
    const bool in = i < r1;
    bool vx = in, vy = in;
    if (in && xv) {
      const int64_t b = P.x.offset + i;
      vx = (xv[b >> 3] >> (b & 7)) & 1;
    }
    if (in && yv) {
      const int64_t b = P.y.offset + i;
      vy = (yv[b >> 3] >> (b & 7)) & 1;
    }
    const int64_t xb = in ? xs[i] : 0, yb = in ? ys[i] : 0;
    acc_value<XF, false>(ax, xb, vx, 0.0);
    acc_value<YF, false>(ay, yb, vy, 0.0);
    ax.cnt += vx ? 1 : 0;
    ay.cnt += vy ? 1 : 0;
    const double a = kll_value<XF>(xb), b = kll_value<YF>(yb);
    como_fold(m, a - pvx, b - pvy, vx && vy);
    n_both += (int64_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(vx && vy));  // wave-uniform, like the tiles'

    if (KLL) {
      if (P.x.kll.picks) {
        kll_push(Kx, rx, a, vx && a == a);
        kll_drain(Kx, rx, P.x.kll, wave_slot, lane);
      }
      if (P.y.kll.picks) {
        kll_push(Ky, ry, b, vy && b == b);
        kll_drain(Ky, ry, P.y.kll, wave_slot, lane);
      }
    }
  }
}

__device__ __forceinline__ void write_partial(ScanPartial *out, const LaneAcc &r, int64_t cnt) {
  ScanPartial p;
  p.non_null = cnt;
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

template <bool XF, bool YF, bool KLL>
__device__ __forceinline__ void pair_body(const ScanPairDesc &P, ScanPartial *out_x, ScanPartial *out_y,
                                          ComomentPartial *out_c, int wave, int lane, double pvx, double pvy) {
  LaneAcc ax, ay;
  acc_init(ax);
  acc_init(ay);
  ComoLane m;
#pragma unroll
  for (int k = 0; k < 5; k++) m.s[k] = m.c[k] = 0.0;
  KllLane Kx, Ky;
  kll_lane_init(Kx, P.x.kll.top);
  kll_lane_init(Ky, P.y.kll.top);
  double *rx = nullptr, *ry = nullptr;
  if (KLL) {
    // rings of the sampled columns of this wave, x first
    const size_t rx_n = P.x.kll.picks ? ((size_t)1 << P.x.kll.top) + kTileRows : 0;
    const size_t ry_n = P.y.kll.picks ? ((size_t)1 << P.y.kll.top) + kTileRows : 0;
    rx = scan_dyn_lds + (size_t)wave * (rx_n + ry_n);
    ry = rx + rx_n;
  }
  const int64_t wave_global = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
  const uint32_t wave_slot = (uint32_t)wave_global;
  const ScanColDesc &cx = P.x, &cy = P.y;
  int64_t cnt_x = 0, cnt_y = 0, n_both = 0;
  uint32_t lane_cnt_x = 0, lane_cnt_y = 0, lane_both = 0;  // tile path: per-lane counts (a lane sees < 2^32 rows)
  ColFast<XF> fx;
  ColFast<YF> fy;
  col_fast_init<XF>(fx);
  col_fast_init<YF>(fy);
  if (cx.n_tiles > 0) {
    global_i64x2_ptr px = (global_i64x2_ptr)(uintptr_t)((const int64_t *)cx.values + cx.offset + cx.head);
    global_i64x2_ptr py = (global_i64x2_ptr)(uintptr_t)((const int64_t *)cy.values + cy.offset + cy.head);
    const bool hx = cx.validity != nullptr, hy = cy.validity != nullptr;
    global_u8_ptr vbx = (global_u8_ptr)(uintptr_t)(cx.validity + ((cx.offset + cx.head) >> 3));
    global_u8_ptr vby = (global_u8_ptr)(uintptr_t)(cy.validity + ((cy.offset + cy.head) >> 3));
    const int32_t n_tiles32 = (int32_t)cx.n_tiles, step32 = (int32_t)n_waves;  // (< 2^31 tiles: 10^12 rows)
    for (int32_t t = (int32_t)wave_global; t < n_tiles32; t += step32) {
      const int64_t p = (int64_t)t * (kTileRows / 2) + lane;
      const i64x2 x0 = __builtin_nontemporal_load(px + p), x1 = __builtin_nontemporal_load(px + p + 64);
      const i64x2 x2 = __builtin_nontemporal_load(px + p + 128), x3 = __builtin_nontemporal_load(px + p + 192);
      const i64x2 y0 = __builtin_nontemporal_load(py + p), y1 = __builtin_nontemporal_load(py + p + 64);
      const i64x2 y2 = __builtin_nontemporal_load(py + p + 128), y3 = __builtin_nontemporal_load(py + p + 192);
      // the lane's validity bits: rows 2 lane, 2 lane + 1 of each 128-row slot share a byte -- one byte load per slot
      // and column.  (The single-column scan fetches the tile's words through the scalar cache; here the wave's
      // scalar registers are the scarce resource -- 16 words held next to the bounds, the samplers' state and two
      // column descriptors were spilled to lanes, a fifth of the loop's vector instructions.)
      uint32_t bx0 = 3, bx1 = 3, bx2 = 3, bx3 = 3, by0 = 3, by1 = 3, by2 = 3, by3 = 3;
      {
        const int64_t vbyte = (int64_t)t * (kTileRows / 8) + (lane >> 2);
        const uint32_t sh2 = 2u * (uint32_t)(lane & 3);
        if (hx) {
          const uint32_t v0 = vbx[vbyte], v1 = vbx[vbyte + 16], v2 = vbx[vbyte + 32], v3 = vbx[vbyte + 48];
          bx0 = (v0 >> sh2) & 3u;
          bx1 = (v1 >> sh2) & 3u;
          bx2 = (v2 >> sh2) & 3u;
          bx3 = (v3 >> sh2) & 3u;
        }
        if (hy) {
          const uint32_t v0 = vby[vbyte], v1 = vby[vbyte + 16], v2 = vby[vbyte + 32], v3 = vby[vbyte + 48];
          by0 = (v0 >> sh2) & 3u;
          by1 = (v1 >> sh2) & 3u;
          by2 = (v2 >> sh2) & 3u;
          by3 = (v3 >> sh2) & 3u;
        }
      }
      ComoTile q;
#pragma unroll
      for (int k = 0; k < 5; k++) q.t[k] = 0.0;
      // COUNT(x), COUNT(y) and the rows both have: per lane from its own bits (the wave's scalar registers are the
      // scarce resource of this kernel: three 64-bit counters and the a[k] & b[k] words were spilled to lanes)
      lane_cnt_x += (bx0 & 1u) + (bx0 >> 1) + (bx1 & 1u) + (bx1 >> 1) + (bx2 & 1u) + (bx2 >> 1) + (bx3 & 1u) + (bx3 >> 1);
      lane_cnt_y += (by0 & 1u) + (by0 >> 1) + (by1 & 1u) + (by1 >> 1) + (by2 & 1u) + (by2 >> 1) + (by3 & 1u) + (by3 >> 1);
      {
        const uint32_t c0 = bx0 & by0, c1 = bx1 & by1, c2 = bx2 & by2, c3 = bx3 & by3;
        lane_both += (c0 & 1u) + (c0 >> 1) + (c1 & 1u) + (c1 >> 1) + (c2 & 1u) + (c2 >> 1) + (c3 & 1u) + (c3 >> 1);
      }
      unsigned long long out_x = 0, out_y = 0;  // lanes holding a value outside the wave's bounds (uniform)
      pair_rows<XF, YF, KLL>(P, ax, ay, fx, fy, out_x, out_y, q, x0, y0, bx0, by0, Kx, Ky, rx, ry, pvx, pvy);
      pair_rows<XF, YF, KLL>(P, ax, ay, fx, fy, out_x, out_y, q, x1, y1, bx1, by1, Kx, Ky, rx, ry, pvx, pvy);
      pair_rows<XF, YF, KLL>(P, ax, ay, fx, fy, out_x, out_y, q, x2, y2, bx2, by2, Kx, Ky, rx, ry, pvx, pvy);
      pair_rows<XF, YF, KLL>(P, ax, ay, fx, fy, out_x, out_y, q, x3, y3, bx3, by3, Kx, Ky, rx, ry, pvx, pvy);
      if (out_x) col_exact_tile<XF>(fx, ax, x0, x1, x2, x3, bx0, bx1, bx2, bx3);
      if (out_y) col_exact_tile<YF>(fy, ay, y0, y1, y2, y3, by0, by1, by2, by3);
      como_tile_flush(m, q);
      if (XF) {
        two_sum_add(ax.s, ax.c, fx.tsum);
        fx.tsum = 0.0;
      }
      if (YF) {
        two_sum_add(ay.s, ay.c, fy.tsum);
        fy.tsum = 0.0;
      }
      if (KLL) {
        if (P.x.kll.picks) kll_drain(Kx, rx, P.x.kll, wave_slot, lane);
        if (P.y.kll.picks) kll_drain(Ky, ry, P.y.kll, wave_slot, lane);
      }
    }
    if (blockIdx.x == gridDim.x - 1) {
      const int64_t tail0 = cx.head + cx.n_tiles * kTileRows;
      if (wave == 0)
        pair_ragged<XF, YF, KLL>(P, 0, cx.head, lane, 64, ax, ay, m, n_both, Kx, Ky, rx, ry, wave_slot, pvx, pvy);
      if (wave == 1)
        pair_ragged<XF, YF, KLL>(P, tail0, cx.length, lane, 64, ax, ay, m, n_both, Kx, Ky, rx, ry, wave_slot, pvx, pvy);
    }
  } else {
    pair_ragged<XF, YF, KLL>(P, wave_global * 64, cx.length, lane, (int)(n_waves * 64), ax, ay, m, n_both, Kx, Ky,
                             rx, ry, wave_slot, pvx, pvy);
  }
  if (KLL) {
    // the samplers' MIN / MAX: what the exact tiles saw, and what the ragged rows added to K.mn / K.mx themselves
    Kx.mn = __builtin_fmin(Kx.mn, fx.dmn);
    Kx.mx = __builtin_fmax(Kx.mx, fx.dmx);
    Ky.mn = __builtin_fmin(Ky.mn, fy.dmn);
    Ky.mx = __builtin_fmax(Ky.mx, fy.dmx);
    if (P.x.kll.picks) kll_finish_wave(Kx, rx, P.x.kll, wave_slot, lane);
    if (P.y.kll.picks) kll_finish_wave(Ky, ry, P.y.kll, wave_slot, lane);
  }
  ax.cnt += lane_cnt_x;
  ay.cnt += lane_cnt_y;
  {
    int64_t both = lane_both;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) both += shfl_down_i64(both, d);
    n_both += __shfl(both, 0, 64);
  }
  // wave reduce
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    LaneAcc bx = acc_shfl_down(ax, d), by = acc_shfl_down(ay, d);
    acc_merge(ax, bx, XF);
    acc_merge(ay, by, YF);
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const double os = __shfl_down(m.s[k], d, 64), oc = __shfl_down(m.c[k], d, 64);
      m.c[k] += oc;
      two_sum_add(m.s[k], m.c[k], os);
    }
  }
  __shared__ LaneAcc s_x[kWavesPerBlock], s_y[kWavesPerBlock];
  __shared__ ComoLane s_m[kWavesPerBlock];
  __shared__ int64_t s_n[kWavesPerBlock][3];
  if (lane == 0) {
    s_x[wave] = ax;
    s_y[wave] = ay;
    s_m[wave] = m;
    s_n[wave][0] = cnt_x;
    s_n[wave][1] = cnt_y;
    s_n[wave][2] = n_both;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    LaneAcc rxa = s_x[0], rya = s_y[0];
    ComomentPartial c;
    c.n = s_n[0][2];
    int64_t cx_n = s_n[0][0], cy_n = s_n[0][1];
#pragma unroll
    for (int k = 0; k < 5; k++) {
      c.s[k] = s_m[0].s[k];
      c.c[k] = s_m[0].c[k];
    }
    for (int w = 1; w < kWavesPerBlock; w++) {
      acc_merge(rxa, s_x[w], XF);
      acc_merge(rya, s_y[w], YF);
      cx_n += s_n[w][0];
      cy_n += s_n[w][1];
      c.n += s_n[w][2];
      for (int k = 0; k < 5; k++) {
        c.c[k] += s_m[w].c[k];
        two_sum_add(c.s[k], c.c[k], s_m[w].s[k]);
      }
    }
    write_partial(out_x, rxa, cx_n + rxa.cnt);
    write_partial(out_y, rya, cy_n + rya.cnt);
    *out_c = c;
  }
}

// grid = (blocks per pair, pairs); partials: [2 * pair + {0, 1}][block] for the columns, como: [pair][block].
// One kernel per (x type, y type, samplers on / off): inside ONE kernel the four type combinations shared a register
// allocation sized for the worst of them (the Int64 paths convert every value to a double) and spilled scalars to
// lanes in every loop; a launch takes the pairs of one combination (`first`: its first pair in the launch's table).
template <bool XF, bool YF, bool KLL>
__global__ __launch_bounds__(kScanBlock) void scan_pair_kernel(const ScanPairLaunch L, int first,
                                                                ScanPartial *__restrict__ partials,
                                                                ComomentPartial *__restrict__ como,
                                                                const ComomentAcc *__restrict__ como_accs) {
  const int pair = first + (int)blockIdx.y;
  const ScanPairDesc &P = L.pairs[pair];
  const double pvx = como_accs[P.como_acc].px, pvy = como_accs[P.como_acc].py;  // the pair's pivots (uniform)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  ScanPartial *ox = partials + ((size_t)2 * pair) * gridDim.x + blockIdx.x;
  ScanPartial *oy = partials + ((size_t)2 * pair + 1) * gridDim.x + blockIdx.x;
  ComomentPartial *oc = como + (size_t)pair * gridDim.x + blockIdx.x;
  pair_body<XF, YF, KLL>(P, ox, oy, oc, wave, lane, pvx, pvy);
}

// Folds the per-block partials of each column (fixed order => bitwise reproducible) and merges
// the batch into the running per-column state.  grid = columns, block = 64 (one wave).
// `outliers` (a key column's uniqueness pass took its aggregates along, kernels/distinct.hip: one column per launch):
// the share of the keys that fell outside the sampled range, collected through global atomics, is one more partial --
// folded here instead of by a launch of its own (round 6: a launch is ~5 us of a 1.6 ms C2 step whatever it does).
__global__ __launch_bounds__(64) void scan_reduce_kernel(const ScanLaunch L,
                                                          const ScanPartial *__restrict__ partials,
                                                          int blocks_per_col,
                                                          ScanAcc *__restrict__ accs,
                                                          const OutlierStats *__restrict__ outliers) {
  const int col = blockIdx.x;
  const int lane = threadIdx.x;
  if (L.acc_index[col] < 0) return;  // a column scanned only for what rode on it (KLL sampler, co-moments)
  const ScanColDesc c = L.cols[col];
  const ScanPartial *p = partials + (size_t)col * blocks_per_col;
  LaneAcc a;
  acc_init(a);
  if (outliers && lane == 0 && outliers->count) {
    LaneAcc b;
    acc_init(b);
    b.mn = outliers->mn;
    b.mx = outliers->mx;
    const __int128 sum = (__int128)(unsigned __int128)outliers->lo32_sum + (((__int128)outliers->hi32_sum) << 32);
    b.lo = (uint64_t)sum;
    b.hi = (int64_t)(sum >> 64);
    b.cnt = (int64_t)outliers->count;
    acc_merge(a, b, c.is_float);
  }
  // (a launch of one wave per column is all latency: four partials per lane are requested before the first is merged
  //  -- merged in the order they always were, lane by lane: i, i + 64, i + 128 ...)
  for (int i0 = lane; i0 < blocks_per_col; i0 += 256) {
    LaneAcc b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + 64 * u;
      if (i < blocks_per_col) {
        b[u].mn = p[i].min_k;
        b[u].mx = p[i].max_k;
        b[u].lo = p[i].sum_lo;
        b[u].hi = p[i].sum_hi;
        b[u].s = p[i].sum;
        b[u].c = p[i].comp;
        b[u].s1 = p[i].s1;
        b[u].s2 = p[i].s2;
        b[u].cnt = p[i].non_null;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (i0 + 64 * u < blocks_per_col) acc_merge(a, b[u], c.is_float);
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
// The narrow types of round 5 (Int8, Int16, UInt8, UInt16, UInt32 -> Int64; Boolean bits -> 0 / 1) take the plain
// loop: one value per lane and trip, an 8-byte store each (these are the rare columns of a table, not its bulk).
__global__ __launch_bounds__(256) void widen_narrow_kernel(const void *__restrict__ src, long long *__restrict__ dst,
                                                           int64_t n, int mode) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    long long v;
    switch (mode) {
      case 2: v = (long long)((const int8_t *)src)[i]; break;
      case 3: v = (long long)((const int16_t *)src)[i]; break;
      case 4: v = (long long)((const uint8_t *)src)[i]; break;
      case 5: v = (long long)((const uint16_t *)src)[i]; break;
      case 6: v = (long long)((const uint32_t *)src)[i]; break;
      default: v = (long long)((((const uint8_t *)src)[i >> 3] >> (i & 7)) & 1); break;  // 7: Boolean
    }
    dst[i] = v;
  }
}

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

void launch_widen32(const void *src, void *dst, int64_t n, int mode, int n_cu, hipStream_t stream) {
  if (n <= 0) return;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > (int64_t)n_cu * 16) blocks = (int64_t)n_cu * 16;
  if (blocks < 1) blocks = 1;
  if (mode >= 2) {
    hipLaunchKernelGGL(widen_narrow_kernel, dim3((int)blocks), dim3(256), 0, stream, src, (long long *)dst, n, mode);
    return;
  }
  hipLaunchKernelGGL(widen32_kernel, dim3((int)blocks), dim3(256), 0, stream, src, dst, n, mode);
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

// columns whose KLL samplers ride on the scan (d.kll.picks set): `lds_bytes` = the workgroup's rings
void launch_scan_kll(const ScanLaunch &L, int n_cols, int blocks_per_col, size_t lds_bytes, ScanPartial *d_partials,
                     hipStream_t stream) {
  hipLaunchKernelGGL(scan_kll_kernel<1>, dim3(blocks_per_col, n_cols), dim3(kScanBlock), lds_bytes, stream, L,
                     d_partials);
}

// columns with a HyperLogLog lane: d.hll = the launch's [workgroup][register] rows of each, d.hll_regs = the running
// registers of its task
void launch_scan_hll(const ScanLaunch &L, int n_cols, int blocks_per_col, ScanPartial *d_partials, hipStream_t stream) {
  hipLaunchKernelGGL(scan_hll_kernel<3>, dim3(blocks_per_col, n_cols), dim3(kScanBlock), 0, stream, L, d_partials);
  const int slices = blocks_per_col >= 64 ? 32 : (blocks_per_col >= 8 ? 4 : 1);
  const int rows_per_slice = (blocks_per_col + slices - 1) / slices;
  hipLaunchKernelGGL(hll_reduce_kernel, dim3(kHllRegisters / 4 / 256 * slices, n_cols), dim3(256), 0, stream, L,
                     blocks_per_col, rows_per_slice);
}

// two Float64 columns: the instance needs 129 registers as the compiler allocates it freely -- one too many for four
// waves per SIMD; held to 128 it runs at 4 instead of 3
template <bool KLL>
__global__ __launch_bounds__(kScanBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void scan_pair_ff_kernel(
    const ScanPairLaunch L, int first, ScanPartial *__restrict__ partials, ComomentPartial *__restrict__ como,
    const ComomentAcc *__restrict__ como_accs) {
  const int pair = first + (int)blockIdx.y;
  const ScanPairDesc &P = L.pairs[pair];
  const double pvx = como_accs[P.como_acc].px, pvy = como_accs[P.como_acc].py;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  ScanPartial *ox = partials + ((size_t)2 * pair) * gridDim.x + blockIdx.x;
  ScanPartial *oy = partials + ((size_t)2 * pair + 1) * gridDim.x + blockIdx.x;
  ComomentPartial *oc = como + (size_t)pair * gridDim.x + blockIdx.x;
  pair_body<true, true, KLL>(P, ox, oy, oc, wave, lane, pvx, pvy);
}

template <bool XF, bool YF>
static void launch_scan_pairs_of(const ScanPairLaunch &L, int first, int count, int blocks_per_pair, size_t lds_bytes,
                                 ScanPartial *d_partials, void *d_como_partials, const ComomentAcc *d_como_accs,
                                 hipStream_t stream) {
  if (XF && YF) {
    if (lds_bytes)
      hipLaunchKernelGGL(scan_pair_ff_kernel<true>, dim3(blocks_per_pair, count), dim3(kScanBlock), lds_bytes, stream, L,
                         first, d_partials, (ComomentPartial *)d_como_partials, d_como_accs);
    else
      hipLaunchKernelGGL(scan_pair_ff_kernel<false>, dim3(blocks_per_pair, count), dim3(kScanBlock), 0, stream, L, first,
                         d_partials, (ComomentPartial *)d_como_partials, d_como_accs);
    return;
  }
  if (lds_bytes)
    hipLaunchKernelGGL((scan_pair_kernel<XF, YF, true>), dim3(blocks_per_pair, count), dim3(kScanBlock), lds_bytes, stream,
                       L, first, d_partials, (ComomentPartial *)d_como_partials, d_como_accs);
  else
    hipLaunchKernelGGL((scan_pair_kernel<XF, YF, false>), dim3(blocks_per_pair, count), dim3(kScanBlock), 0, stream, L,
                       first, d_partials, (ComomentPartial *)d_como_partials, d_como_accs);
}

// the pairs arrive grouped by type combination (tgx_update sorts them): one launch per run of equal combinations
void launch_scan_pairs(const ScanPairLaunch &L, int n_pairs, int blocks_per_pair, size_t lds_bytes,
                       ScanPartial *d_partials, void *d_como_partials, const ComomentAcc *d_como_accs,
                       hipStream_t stream) {
  int first = 0;
  while (first < n_pairs) {
    const bool xf = L.pairs[first].x.is_float != 0, yf = L.pairs[first].y.is_float != 0;
    int count = 1;
    while (first + count < n_pairs && (L.pairs[first + count].x.is_float != 0) == xf &&
           (L.pairs[first + count].y.is_float != 0) == yf)
      count++;
    if (xf && yf)
      launch_scan_pairs_of<true, true>(L, first, count, blocks_per_pair, lds_bytes, d_partials, d_como_partials, d_como_accs, stream);
    else if (xf)
      launch_scan_pairs_of<true, false>(L, first, count, blocks_per_pair, lds_bytes, d_partials, d_como_partials, d_como_accs, stream);
    else if (yf)
      launch_scan_pairs_of<false, true>(L, first, count, blocks_per_pair, lds_bytes, d_partials, d_como_partials, d_como_accs, stream);
    else
      launch_scan_pairs_of<false, false>(L, first, count, blocks_per_pair, lds_bytes, d_partials, d_como_partials, d_como_accs, stream);
    first += count;
  }
}

void launch_scan_reduce_only(const ScanLaunch &L, int n_cols, int blocks_per_col, ScanPartial *d_partials,
                             ScanAcc *d_accs, hipStream_t stream, const OutlierStats *outliers) {
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(n_cols), dim3(64), 0, stream, L, d_partials, blocks_per_col,
                     d_accs, outliers);
}

void launch_count(const CountLaunch &L, int n_cols, int blocks_per_col, unsigned long long *d_block_counts,
                  CountAcc *d_accs, hipStream_t stream) {
  hipLaunchKernelGGL(count_kernel, dim3(blocks_per_col, n_cols), dim3(256), 0, stream, L, d_block_counts,
                     blocks_per_col == 1 ? d_accs : nullptr);
  if (blocks_per_col > 1)
    hipLaunchKernelGGL(count_reduce_kernel, dim3(n_cols), dim3(64), 0, stream, L, d_block_counts,
                       blocks_per_col, d_accs);
}

// ---- tgx_state_reset's device part (device_types.h, StateResetArgs) ----
__global__ __launch_bounds__(256) void state_reset_kernel(StateResetArgs a) {
  for (int r = 0; r < a.n_zero; r++) {
    uint32_t *p = (uint32_t *)a.zero[r];
    const uint32_t words = a.zero_bytes[r] >> 2;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < words; i += gridDim.x * 256u) p[i] = 0;
  }
  // the accumulators' identities, two 8-byte words at a time (ScanAcc is 12 of them: words 2 and 3 are MIN and MAX)
  constexpr uint32_t W = sizeof(ScanAcc) / 8;
  static_assert(sizeof(ScanAcc) % 8 == 0 && offsetof(ScanAcc, min_k) == 16 && offsetof(ScanAcc, max_k) == 24, "layout");
  long long *dst = (long long *)a.ident;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < a.n_ident * W; i += gridDim.x * 256u) {
    const uint32_t w = i % W;
    dst[i] = w == 2 ? INT64_MAX : w == 3 ? INT64_MIN : 0;
  }
}
void launch_state_reset(const StateResetArgs &a, hipStream_t stream) {
  uint32_t most = a.n_ident * (uint32_t)sizeof(ScanAcc);
  for (int r = 0; r < a.n_zero; r++) most = a.zero_bytes[r] > most ? a.zero_bytes[r] : most;
  const unsigned blocks = most <= 4096u ? 1u : (most / 4096u > 64u ? 64u : most / 4096u);
  hipLaunchKernelGGL(state_reset_kernel, dim3(blocks), dim3(256), 0, stream, a);
}

}  // namespace tgx
