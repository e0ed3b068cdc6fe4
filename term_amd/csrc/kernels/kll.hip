// kll.hip -- KLL quantile sketch construction on gfx950.
//
// The reference drives KllSketch::update one value at a time on the host
// (TG/analyzers/advanced/kll_sketch.rs:195-229: push to the level-0 buffer, sort + halve when it is
// full, cascade).  Here every workgroup sketches its own row range with the same primitive --
// "sort a buffer, keep every other item, promote them one level (weight x2)" -- at a fixed run length:
//   level 0      : up to 1023 raw items (weight 1)
//   level l >= 1 : zero or one sorted run of exactly 512 items (weight 2^l)
// Inserting a run into an occupied level merges the two runs (bitonic merge in LDS), keeps every
// other item and carries the result one level up, like a binary counter.  Workgroup sketches are then
// combined pairwise by the same insertion (a log2(G)-round tree) and folded into the state's running
// sketch, so the whole column costs one HBM pass; nothing but the final ~100 KiB sketch ever reaches
// the host.  Total weight is preserved exactly (sum over levels of items x 2^level == n).
#include <hip/hip_runtime.h>
#include <string.h>

#include "kll_types.h"

namespace tgx {

typedef const int64_t __attribute__((address_space(1))) *global_i64_ptr;
typedef const uint8_t __attribute__((address_space(1))) *global_u8_ptr;

constexpr int kKllThreads = 256;

__device__ __forceinline__ uint64_t kll_mix(uint64_t x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}

// full ascending bitonic sort of buf[0..1023] by 256 threads
__device__ void block_sort_1024(double *buf) {
  const uint32_t t0 = threadIdx.x;
  for (uint32_t k = 2; k <= 1024; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (uint32_t u = 0; u < 2; u++) {
        const uint32_t t = t0 + u * kKllThreads;  // comparator index 0..511
        const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const uint32_t p = i | j;
        const double a = buf[i], b = buf[p];
        const bool up = (i & k) == 0;
        if ((a > b) == up) {
          buf[i] = b;
          buf[p] = a;
        }
      }
      __syncthreads();
    }
  }
}

// buf[0..511] ascending and buf[512..1023] DESCENDING (a bitonic sequence) -> fully ascending
__device__ void block_bitonic_merge_1024(double *buf) {
  const uint32_t t0 = threadIdx.x;
  for (uint32_t j = 512; j > 0; j >>= 1) {
#pragma unroll
    for (uint32_t u = 0; u < 2; u++) {
      const uint32_t t = t0 + u * kKllThreads;
      const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
      const uint32_t p = i | j;
      const double a = buf[i], b = buf[p];
      if (a > b) {
        buf[i] = b;
        buf[p] = a;
      }
    }
    __syncthreads();
  }
}

// Inserts the sorted 512-item run held in buf[0..511] at `level` of sketch s (global memory),
// carrying upwards while the level is occupied.  buf (1024 doubles of LDS) is scratch.
__device__ void insert_run(KllDeviceSketch *s, uint32_t level, double *buf, uint64_t salt) {
  const uint32_t t = threadIdx.x;
  for (;;) {
    const uint64_t mask = s->level_mask;  // uniform: every thread reads the same word
    __syncthreads();
    if (level >= kKllMaxLevels) return;  // unreachable for n < 2^56
    if (!((mask >> level) & 1)) {
      double *dst = s->runs[level];
      dst[t] = buf[t];
      dst[t + 256] = buf[t + 256];
      __syncthreads();
      if (t == 0) s->level_mask = mask | (1ull << level);
      __threadfence_block();
      __syncthreads();
      return;
    }
    // occupied: merge the stored run (loaded reversed into the upper half) with the incoming one
    const double *src = s->runs[level];
    buf[1023 - t] = src[t];
    buf[1023 - (t + 256)] = src[t + 256];
    __syncthreads();
    block_bitonic_merge_1024(buf);
    const uint32_t parity = (uint32_t)(kll_mix(salt ^ ((uint64_t)level << 40) ^ mask) >> 33) & 1u;
    const double a = buf[2 * t + parity], b = buf[2 * (t + 256) + parity];
    __syncthreads();
    buf[t] = a;
    buf[t + 256] = b;
    __syncthreads();
    if (t == 0) s->level_mask = mask & ~(1ull << level);
    __syncthreads();
    level += 1;
  }
}

// sorts buf[0..1023] (raw items), halves them into buf[0..511] and inserts the run at level 1
__device__ void compact_level0(KllDeviceSketch *s, double *buf, uint64_t salt, uint32_t shift) {
  const uint32_t t = threadIdx.x;
  block_sort_1024(buf);
  const uint32_t parity = (uint32_t)(kll_mix(salt ^ 0x5bd1e995ULL) >> 35) & 1u;
  const double a = buf[2 * t + parity], b = buf[2 * (t + 256) + parity];
  __syncthreads();
  buf[t] = a;
  buf[t + 256] = b;
  __syncthreads();
  insert_run(s, 1 + shift, buf, salt);
}

// the same with ONE barrier: the caller alternates between two `wave_tot` buffers from call to call (a wave that
// runs ahead writes the other buffer; it cannot come round to this one again before everybody has passed the next
// call's barrier, i.e. has finished reading here).  The barriers were the cost of phase A: 3 per 2048-row step ran
// 2.13 ms per 1 G-row column, a fourth one 3.07 ms.
__device__ __forceinline__ uint32_t block_exclusive_scan1(uint32_t v, uint32_t *wave_tot, uint32_t *total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t up = __shfl_up(incl, d, 64);
    if (lane >= (uint32_t)d) incl += up;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (uint32_t w = 0; w < kKllThreads / 64; w++) {
    const uint32_t x = wave_tot[w];
    if (w < wave) base += x;
    tot += x;
  }
  *total = tot;
  return base + incl - v;
}

// block-wide exclusive prefix sum of a small per-thread count; returns the total through *total
__device__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *wave_tot, uint32_t *total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t up = __shfl_up(incl, d, 64);
    if (lane >= (uint32_t)d) incl += up;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (uint32_t w = 0; w < kKllThreads / 64; w++) {
    if (w < wave) base += wave_tot[w];
    tot += wave_tot[w];
  }
  *total = tot;
  __syncthreads();
  return base + incl - v;
}

constexpr int kKllRowsPerThread = 16;
constexpr int kKllStepRows = kKllThreads * kKllRowsPerThread;  // 2048 rows per workgroup step

// The 16 rows a thread takes of the 2048-row step at `base` (row pairs base + 2 (t + 256 u) + {0, 1}: one
// global_load_dwordx4 per pair when the buffer allows).  Split in two so that phase A can request the NEXT step's
// rows before it works on the current ones: with two workgroups per CU (64 KiB ring) nothing else hides the load
// latency -- unprefetched the kernel streamed at 3.1 TB/s.
struct KllRaw {
  int64_t bits[kKllRowsPerThread];
  uint8_t vb[kKllRowsPerThread / 2];
  bool fast;  // whole step, 16-byte aligned: `bits` (and `vb` when pair_bytes) are in flight
};

__device__ __forceinline__ void kll_issue(const KllColDesc &d, global_i64_ptr vals, global_u8_ptr vbits, int64_t base,
                                          int64_t r1, bool wide, KllRaw &raw) {
  typedef long long i64x2 __attribute__((ext_vector_type(2)));
  typedef const i64x2 __attribute__((address_space(1))) *global_i64x2_ptr;
  raw.fast = wide && base + kKllStepRows <= r1;
  if (!raw.fast) return;
  const bool pair_bytes = ((d.offset + base) & 1) == 0;  // both rows of a pair share a validity byte
  if (vbits && pair_bytes) {
#pragma unroll
    for (int u = 0; u < kKllRowsPerThread / 2; u++)
      raw.vb[u] = vbits[(d.offset + base + 2 * (threadIdx.x + u * kKllThreads)) >> 3];
  }
  global_i64x2_ptr pv = (global_i64x2_ptr)(vals + base) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < kKllRowsPerThread / 2; u++) {
    const i64x2 x = __builtin_nontemporal_load(pv + u * kKllThreads);
    raw.bits[2 * u] = x.x;
    raw.bits[2 * u + 1] = x.y;
  }
}

// values + the mask of rows that enter the sketch (in range, non-NULL, not NaN: KllSketch::update drops NaN,
// kll_sketch.rs:197-199)
__device__ __forceinline__ uint32_t kll_finish(const KllColDesc &d, global_i64_ptr vals, global_u8_ptr vbits,
                                               int64_t base, int64_t r1, KllRaw &raw,
                                               double (&v)[kKllRowsPerThread]) {
  uint32_t okm = 0;
  if (raw.fast) {
    const bool pair_bytes = ((d.offset + base) & 1) == 0;
    okm = (1u << kKllRowsPerThread) - 1u;
    if (vbits && pair_bytes) {
      okm = 0;
#pragma unroll
      for (int u = 0; u < kKllRowsPerThread / 2; u++) {
        const int64_t b = d.offset + base + 2 * (threadIdx.x + u * kKllThreads);
        okm |= (uint32_t)((raw.vb[u] >> (b & 7)) & 3) << (2 * u);
      }
    } else if (vbits) {
      okm = 0;
#pragma unroll
      for (int q = 0; q < kKllRowsPerThread; q++) {
        const int64_t b = d.offset + base + 2 * (threadIdx.x + (q / 2) * kKllThreads) + (q & 1);
        okm |= (uint32_t)((vbits[b >> 3] >> (b & 7)) & 1) << q;
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < kKllRowsPerThread; q++) {
      const int64_t i = base + 2 * (threadIdx.x + (q / 2) * kKllThreads) + (q & 1);
      bool ok = i < r1;
      raw.bits[q] = ok ? vals[i] : 0;
      if (ok && vbits) {
        const int64_t b = d.offset + i;
        ok = (vbits[b >> 3] >> (b & 7)) & 1;
      }
      okm |= (uint32_t)ok << q;
    }
  }
  // rows that do not enter the sketch leave as NaN: min / max then need no test per value (v_min_f64 / v_max_f64
  // return the other operand), and "enters the sketch" is x == x
  const double nan = __longlong_as_double(0x7FF8000000000000LL);
#pragma unroll
  for (int q = 0; q < kKllRowsPerThread; q++) {
    double x = d.is_float ? __longlong_as_double(raw.bits[q]) : (double)raw.bits[q];
    x = ((okm >> q) & 1) ? x : nan;
    if (!(x == x)) okm &= ~(1u << q);
    v[q] = x;
  }
  return okm;
}

__device__ __forceinline__ uint32_t kll_load_step(const KllColDesc &d, global_i64_ptr vals, global_u8_ptr vbits,
                                                  int64_t base, int64_t r1, bool wide,
                                                  double (&v)[kKllRowsPerThread]) {
  KllRaw raw;
  kll_issue(d, vals, vbits, base, r1, wide, raw);
  return kll_finish(d, vals, vbits, base, r1, raw, v);
}

// One workgroup sketches rows [wg * chunk, (wg+1) * chunk) of the column into sketches[wg].
//
// Sorting every 1024 values (the level-0 compaction) costs ~40 LDS compare-exchanges per value: 36.6 ms per
// 1 G-row column.  The KLL sampler (Karnin-Lang-Liberty, sec. 3.2 "sampling": below the lowest kept level an
// item of weight 2^l is ONE uniformly chosen member of 2^l consecutive stream items) removes that work
// without touching the error budget, arranged so that the total weight stays exact.  The V values of the range
// (NULL / NaN excluded) are cut into
//     a_top segments of 1024 * 2^top values, then one optional segment of 1024 * 2^l values for every
//     l < top (the binary digits of the rest), then < 1024 raw values:   V = sum_l a_l * 1024 * 2^l + r.
// A level-l segment contributes one uniformly chosen value per group of 2^l consecutive values -- 1024 values
// of weight 2^l, sorted, halved to a run of 512 at level l+1 -- and the r raw values are the sketch's level 0.
// V is not known in advance, so the range is streamed ONCE at level `top` (phase A: whole level-top segments
// are cut as they complete; n / min / max come out of the same pass) and only the rows after the last complete
// segment -- fewer than 1024 * 2^top values -- are read again (phase B) to place the lower digits and the raw
// tail.  `top` grows with the batch (2^top ~ rows / 2^22): rank variance added by sampling is <= rows * 2^top / 4,
// i.e. a relative standard error <= 2.5e-4 next to the 2e-3 of the level structure itself; batches under 8 M rows
// are not sampled at all (top = 0: phase A is then the plain "sort every 1024 values").
// `shift`: the input values are pre-sampled items of weight 2^shift (the picks of the fused scan): runs land `shift`
// levels higher, the loose items keep weight 2^shift (KllDeviceSketch::shift), n counts stream items.
__global__ __launch_bounds__(kKllThreads) void kll_build_kernel(const KllJobs J) {
  const KllJob &job = J.job[blockIdx.y];
  if ((int)blockIdx.x >= job.groups) return;  // (uniform: jobs of one launch differ in size)
  const KllColDesc d = job.d;
  const int64_t chunk = job.chunk;
  KllDeviceSketch *sketches = job.sketches;
  const uint64_t salt0 = job.salt;
  const uint32_t top = job.top, shift = job.shift;
  __shared__ double ring[8192];  // eight batches of sampled values: slot = sampled index & 8191 (a step brings <= 4)
  __shared__ double buf[1024];
  __shared__ uint32_t wave_tot[kKllThreads / 64];
  __shared__ uint32_t wave_tot2[2][kKllThreads / 64];  // phase A's one-barrier scan alternates between the two
  uint32_t step_parity = 0;
  __shared__ double red_min[kKllThreads / 64], red_max[kKllThreads / 64];
  KllDeviceSketch *s = sketches + blockIdx.x;
  const uint32_t t = threadIdx.x;
  if (t == 0) {
    s->n = 0;
    s->level_mask = 0;
    s->lv0_count = 0;
    s->shift = shift;
    s->min_v = __longlong_as_double(0x7FF0000000000000LL);
    s->max_v = __longlong_as_double((long long)0xFFF0000000000000ULL);
  }
  __syncthreads();
  global_i64_ptr vals = (global_i64_ptr)(uintptr_t)((const int64_t *)d.values + d.offset);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  const int64_t r0 = (int64_t)blockIdx.x * chunk;
  int64_t r1 = r0 + chunk;
  if (r1 > d.length) r1 = d.length;
  const uint64_t salt = salt0 ^ ((uint64_t)blockIdx.x * 0x9e3779b97f4a7c15ULL);
  const bool wide = (((uintptr_t)(vals + r0)) & 15) == 0;  // chunk is a multiple of the step: parity holds
  const uint64_t top_items = 1024ull << top;

  // sorts the 1024 sampled values of batch kb (weight 2^level), keeps every other one, inserts the run above
  auto flush_batch = [&](uint64_t kb, uint32_t level) {
#pragma unroll
    for (int u = 0; u < 4; u++) buf[t + u * kKllThreads] = ring[((kb & 7) << 10) + t + u * kKllThreads];
    __syncthreads();
    block_sort_1024(buf);
    const uint32_t parity = (uint32_t)(kll_mix(salt ^ 0x5bd1e995ULL ^ (kb << 1)) >> 35) & 1u;
    const double a = buf[2 * t + parity], b = buf[2 * (t + 256) + parity];
    __syncthreads();
    buf[t] = a;
    buf[t + 256] = b;
    __syncthreads();
    insert_run(s, level + 1 + shift, buf, salt ^ (kb << 24));
  };
  auto pick_of = [&](uint32_t level, uint64_t seg0, uint64_t g) -> uint64_t {
    return kll_mix(salt ^ ((uint64_t)level << 56) ^ (g + (seg0 << 8))) & ((1ull << level) - 1);
  };

  // ---- phase A: stream the range once; every value is a candidate of a level-top group ----
  double mn = __longlong_as_double(0x7FF0000000000000LL), mx = -mn;
  uint64_t consumed = 0, kb = 0;               // values seen so far / level-top batches flushed (uniform)
  int64_t tail_base = r0;                      // first step holding values past the last complete batch
  uint64_t tail_consumed = 0;                  // values before that step
  KllRaw raw;
  kll_issue(d, vals, vbits, r0, r1, wide, raw);
  for (int64_t base = r0; base < r1; base += kKllStepRows) {
    double v[kKllRowsPerThread];
    const uint32_t okm = kll_finish(d, vals, vbits, base, r1, raw, v);
    if (base + kKllStepRows < r1) kll_issue(d, vals, vbits, base + kKllStepRows, r1, wide, raw);  // in flight below
    uint32_t total;
    uint64_t j = consumed + block_exclusive_scan1(__builtin_popcount(okm), wave_tot2[step_parity], &total);
    step_parity ^= 1;
    if (top >= 4) {
      // a thread's <= 16 values are consecutive in j, so they lie in at most two level-top groups: two picks
      // (two 64-bit hashes) per step instead of one per value; positions are compared as 32-bit offsets from the
      // thread's first one.  The loop was the kernel's bound (35 VALU instructions per value at two waves per
      // SIMD): no branch per value, NaN-ignoring min / max (kll_finish).
      const uint64_t g0 = j >> top;
      const uint64_t d0 = (g0 << top) + pick_of(top, 0, g0) - j;            // < 16: the pick of g0 is one of ours
      const uint64_t d1 = ((g0 + 1) << top) + pick_of(top, 0, g0 + 1) - j;  // < 16: the pick of g0 + 1 is
      const uint32_t w0 = d0 < 64 ? (uint32_t)d0 : 64u, w1 = d1 < 64 ? (uint32_t)d1 : 64u;
      uint32_t jr = 0;
#pragma unroll
      for (int u = 0; u < kKllRowsPerThread; u++) {
        const uint32_t ok = (okm >> u) & 1;
        mn = __builtin_fmin(mn, v[u]);
        mx = __builtin_fmax(mx, v[u]);
        if (ok && jr == w0) ring[g0 & 8191] = v[u];
        if (ok && jr == w1) ring[(g0 + 1) & 8191] = v[u];
        jr += ok;
      }
    } else {
#pragma unroll
      for (int u = 0; u < kKllRowsPerThread; u++) {
        if (!((okm >> u) & 1)) continue;
        mn = v[u] < mn ? v[u] : mn;
        mx = v[u] > mx ? v[u] : mx;
        const uint64_t g = j >> top;
        if ((j & ((1ull << top) - 1)) == pick_of(top, 0, g)) ring[g & 8191] = v[u];
        j++;
      }
    }
    const uint64_t before = consumed;
    consumed += total;
    // (the picks of batch kb must be in the ring before it is flushed; no other step-to-step hazard: the scan
    // alternates its buffers, and ring slots are reused eight batches later, with this barrier in between)
    if (consumed >= (kb + 1) * top_items) __syncthreads();
    while (consumed >= (kb + 1) * top_items) {  // uniform: every pick of batch kb has been written
      flush_batch(kb, top);
      kb++;
      const uint64_t end = kb * top_items;
      tail_base = consumed > end ? base : base + kKllStepRows;
      tail_consumed = consumed > end ? before : consumed;
    }
  }
  const uint64_t V = consumed;
#pragma unroll
  for (int dlt = 32; dlt >= 1; dlt >>= 1) {
    const double omn = __shfl_down(mn, dlt, 64), omx = __shfl_down(mx, dlt, 64);
    mn = omn < mn ? omn : mn;
    mx = omx > mx ? omx : mx;
  }
  if ((t & 63) == 0) {
    red_min[t >> 6] = mn;
    red_max[t >> 6] = mx;
  }
  __syncthreads();
  if (t == 0) {
    double a = red_min[0], b = red_max[0];
    for (int w = 0; w < kKllThreads / 64; w++) {
      a = red_min[w] < a ? red_min[w] : a;
      b = red_max[w] > b ? red_max[w] : b;
    }
    s->n = V << shift;
    s->min_v = a;
    s->max_v = b;
  }

  // ---- phase B: the values after the last level-top segment: the binary digits X of their count, then raw ----
  const uint64_t end_top = kb * top_items;
  const uint32_t X = (uint32_t)((V - end_top) >> 10);        // < 2^top: digit l = one segment of level l
  const uint64_t raw_start = end_top + ((uint64_t)X << 10);  // first raw value
  const uint64_t batches_top = kb;
  uint32_t xrest = X, lb = 0;
  uint64_t jb1 = end_top;  // end of the open lower-level batch
  bool have_batch = false;
  auto open_batch = [&]() {
    have_batch = xrest != 0;
    if (have_batch) {
      lb = 31 - __builtin_clz(xrest);
      xrest &= ~(1u << lb);
      jb1 += 1024ull << lb;
    }
  };
  open_batch();
  if (V > end_top) {
    consumed = tail_consumed;
    for (int64_t base = tail_base; base < r1; base += kKllStepRows) {
      double v[kKllRowsPerThread];
      const uint32_t okm = kll_load_step(d, vals, vbits, base, r1, wide, v);
      uint32_t total;
      uint64_t j = consumed + block_exclusive_scan(__builtin_popcount(okm), wave_tot, &total);
#pragma unroll
      for (int u = 0; u < kKllRowsPerThread; u++) {
        if (!((okm >> u) & 1)) continue;
        if (j >= raw_start) {
          s->lv0[j - raw_start] = v[u];
        } else if (j >= end_top) {
          const uint32_t q = (uint32_t)((j - end_top) >> 10);   // < X
          const uint32_t level = 31 - __builtin_clz(X ^ q);      // the digit of X that q falls under
          const uint32_t before = X & ~((2u << level) - 1u);     // 1024-value blocks of the higher digits
          const uint64_t seg0 = end_top + ((uint64_t)before << 10);
          const uint64_t sampled0 = (batches_top + __builtin_popcount(before)) << 10;
          const uint64_t o = j - seg0, g = o >> level;
          if ((o & ((1ull << level) - 1)) == pick_of(level, seg0, g)) ring[(sampled0 + g) & 8191] = v[u];
        }
        j++;
      }
      consumed += total;
      __syncthreads();
      while (have_batch && consumed >= jb1) {
        flush_batch(kb, lb);
        kb++;
        open_batch();
      }
    }
  }
  if (t == 0) s->lv0_count = (uint32_t)(V - raw_start);
}

// dst += src (both in global memory); one workgroup.
__device__ void sketch_add(KllDeviceSketch *dst, const KllDeviceSketch *src, double *buf,
                           double *staging, uint64_t salt) {
  const uint32_t t = threadIdx.x;
  const unsigned long long src_n = src->n;
  if (src_n == 0) return;  // uniform
  // level 0: concatenate the raw items; 1024 or more -> compact 1024 of them
  const uint32_t a = dst->lv0_count, b = src->lv0_count;
  for (uint32_t i = t; i < a; i += kKllThreads) staging[i] = dst->lv0[i];
  for (uint32_t i = t; i < b; i += kKllThreads) staging[a + i] = src->lv0[i];
  __syncthreads();
  uint32_t staged = a + b;
  if (staged >= 1024) {
    const uint32_t from = staged - 1024;
#pragma unroll
    for (int u = 0; u < 4; u++) buf[t + u * kKllThreads] = staging[from + t + u * kKllThreads];
    __syncthreads();
    compact_level0(dst, buf, salt ^ staged, dst->shift);  // (dst and src carry the same shift: one tree per input)
    staged = from;
  }
  for (uint32_t i = t; i < staged; i += kKllThreads) dst->lv0[i] = staging[i];
  __syncthreads();
  // runs: insert each of src's runs
  const uint64_t src_mask = src->level_mask;
  for (uint32_t l = 1; l < kKllMaxLevels; l++) {
    if (!((src_mask >> l) & 1)) continue;
    buf[t] = src->runs[l][t];
    buf[t + 256] = src->runs[l][t + 256];
    __syncthreads();
    insert_run(dst, l, buf, salt ^ ((uint64_t)l << 20));
  }
  if (t == 0) {
    dst->lv0_count = staged;
    dst->n += src_n;
    dst->min_v = src->min_v < dst->min_v ? src->min_v : dst->min_v;
    dst->max_v = src->max_v > dst->max_v ? src->max_v : dst->max_v;
  }
  __syncthreads();
}

// one round of the pairwise tree: sketches[2*i*stride] += sketches[(2*i+1)*stride]
__global__ __launch_bounds__(kKllThreads) void kll_tree_kernel(const KllJobs J, int stride) {
  __shared__ double staging[2048];
  __shared__ double buf[1024];
  const KllJob &job = J.job[blockIdx.y];
  const int a = 2 * blockIdx.x * stride, b = a + stride;
  if (b >= job.groups) return;
  sketch_add(job.sketches + a, job.sketches + b, buf, staging,
             (job.salt + (uint64_t)stride) ^ ((uint64_t)a << 8) ^ (uint64_t)stride);
}

// state += batch
__global__ __launch_bounds__(kKllThreads) void kll_fold_kernel(const KllJobs J) {
  __shared__ double staging[2048];
  __shared__ double buf[1024];
  const KllJob &job = J.job[blockIdx.x];
  sketch_add(job.state, job.sketches, buf, staging, job.salt ^ 0xabcdefULL);
}

__global__ void kll_init_kernel(KllDeviceSketch *s, uint32_t shift) {
  if (threadIdx.x == 0) {
    s->n = 0;
    s->level_mask = 0;
    s->lv0_count = 0;
    s->shift = shift;
    s->min_v = __longlong_as_double(0x7FF0000000000000LL);
    s->max_v = __longlong_as_double((long long)0xFFF0000000000000ULL);
  }
}

// The fused scan's per-wave facts: the column's true NaN-ignoring MIN / MAX (the sampled picks need not contain
// them; KllSketch::get_quantile answers phi = 0 / 1 with them, kll_sketch.rs:250-256) go into the running sketch.
__global__ __launch_bounds__(256) void kll_meta_kernel(const KllWaveMeta *meta, int n_waves, KllDeviceSketch *state) {
  double mn = __longlong_as_double(0x7FF0000000000000LL), mx = -mn;
  for (int i = threadIdx.x; i < n_waves; i += 256) {
    if (meta[i].count == 0) continue;
    mn = meta[i].min_v < mn ? meta[i].min_v : mn;
    mx = meta[i].max_v > mx ? meta[i].max_v : mx;
  }
#pragma unroll
  for (int dlt = 32; dlt >= 1; dlt >>= 1) {
    const double omn = __shfl_down(mn, dlt, 64), omx = __shfl_down(mx, dlt, 64);
    mn = omn < mn ? omn : mn;
    mx = omx > mx ? omx : mx;
  }
  __shared__ double smn[4], smx[4];
  if ((threadIdx.x & 63) == 0) {
    smn[threadIdx.x >> 6] = mn;
    smx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) {
      mn = smn[w] < mn ? smn[w] : mn;
      mx = smx[w] > mx ? smx[w] : mx;
    }
    state->min_v = mn < state->min_v ? mn : state->min_v;
    state->max_v = mx > state->max_v ? mx : state->max_v;
  }
}

void launch_kll_meta(const KllWaveMeta *meta, int n_waves, KllDeviceSketch *state, hipStream_t stream) {
  hipLaunchKernelGGL(kll_meta_kernel, dim3(1), dim3(256), 0, stream, meta, n_waves, state);
}

void launch_kll_init(KllDeviceSketch *s, hipStream_t stream, uint32_t shift) {
  hipLaunchKernelGGL(kll_init_kernel, dim3(1), dim3(64), 0, stream, s, shift);
}

// sketches: scratch for `groups` per-workgroup sketches; result folded into `state`
// the sampling level of a batch: 2^top ~ rows / 2^22 (none below 8 M rows), see kll_build_kernel
uint32_t kll_top_for(int64_t rows) {
  uint32_t top = 0;
  while (top < 16 && (rows >> (23 + top)) > 0) top++;
  return top;
}

// jobs[k].sketches: scratch for jobs[k].groups per-workgroup sketches; each job's result is folded into its `state`.
// Jobs that share a state must not share a launch (the fold is one workgroup per job).
void launch_kll_jobs(const KllJob *jobs, int n_jobs, hipStream_t stream) {
  for (int j0 = 0; j0 < n_jobs; j0 += kKllMaxJobs) {
    const int n = n_jobs - j0 < kKllMaxJobs ? n_jobs - j0 : kKllMaxJobs;
    KllJobs J;
    memset(&J, 0, sizeof(J));
    int max_groups = 1;
    for (int k = 0; k < n; k++) {
      J.job[k] = jobs[j0 + k];
      max_groups = J.job[k].groups > max_groups ? J.job[k].groups : max_groups;
    }
    // (measured on 1 G rows, 1024 workgroups: top - 1 / - 2 / - 3 = 2.39 / 2.92 / 3.87 ms instead of 2.10 -- a flush
    // (sort of 1024 + insert) costs ~35 us per workgroup; top + 2 / + 4 = 3.36 / 4.97 ms -- the tail re-read of phase B)
    hipLaunchKernelGGL(kll_build_kernel, dim3(max_groups, n), dim3(kKllThreads), 0, stream, J);
    for (int stride = 1; stride < max_groups; stride <<= 1) {
      const int pairs = (max_groups + 2 * stride - 1) / (2 * stride);
      hipLaunchKernelGGL(kll_tree_kernel, dim3(pairs, n), dim3(kKllThreads), 0, stream, J, stride);
    }
    hipLaunchKernelGGL(kll_fold_kernel, dim3(n), dim3(kKllThreads), 0, stream, J);
  }
}

}  // namespace tgx
