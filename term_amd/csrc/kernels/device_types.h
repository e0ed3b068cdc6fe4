// Device-visible descriptors and partial-state layouts shared by the HIP kernels and the C-ABI
// host code (term_amd/csrc/tgx_api.cpp).  gfx950 only.
#pragma once
#include <stdint.h>

#include "kll_types.h"

namespace tgx {

constexpr int kScanBlock = 256;      // 4 waves of 64
constexpr int kWavesPerBlock = 4;
constexpr int kTileRows = 512;       // rows one wave consumes per iteration: 4 x (64 lanes x 16 B)
constexpr int kMaxScanBlocksPerCol = 2048;  // 256 CUs x 8 resident 256-thread blocks

// One numeric column of one batch, as the scan kernel sees it.
struct ScanColDesc {
  const void *values;       // element 0 of the Arrow values buffer (int64 / float64)
  const uint8_t *validity;  // LSB-first bitmap or nullptr
  int64_t offset;           // Arrow offset (slots)
  int64_t length;           // rows
  int64_t head;             // rows [0, head): ragged edge, per-lane path
  int64_t n_tiles;          // full kTileRows tiles starting at row `head` (0 => whole column ragged)
  int32_t is_float;
  int32_t want_variance;
  const double *pivot;      // device scalar: shift for the variance lanes (may be nullptr)
  int32_t elem32;           // 1: `values` holds 4-byte elements (Int32 / Float32), widened as they are loaded
  int32_t skip_stats;       // scan_hll_kernel: nobody asked for MIN / MAX / SUM of this column -- COUNT and registers only
  ScanKll kll;              // kll.picks != nullptr: the column's KLL sampler rides on this scan (kll_types.h)
  uint8_t *hll;             // scan_hll_kernel: [workgroup][kHllRegisters] register bytes of this launch (else nullptr)
  uint8_t *hll_regs;        // ... and the task's running registers the launch is folded into (hll_reduce_kernel)
};

// HyperLogLog lane of the scan (APPROX_DISTINCT, TG/constraints/approx_count_distinct.rs:56-66): 2^14 registers like
// DataFusion's sketch (datafusion-functions-aggregate 50.3.0, hyperloglog.rs: HLL_P = 14).  A value's 64 bits are
// mixed by a bijection into (a, b): register = low 14 bits of a, rank = leading zeros of b + 1 (1 .. 33).
constexpr int kHllBits = 14;
constexpr int kHllRegisters = 1 << kHllBits;
constexpr int kHllMaxRank = 33;
__host__ __device__ inline uint32_t hll_rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__host__ __device__ inline void hll_hash(uint64_t bits, uint32_t *a_out, uint32_t *b_out) {
  const uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
  uint32_t a = lo ^ hll_rotl32(hi * 0x9E3779B1u, 15);
  a ^= a >> 16;  // Murmur3's 32-bit finaliser
  a *= 0x85EBCA6Bu;
  a ^= a >> 13;
  a *= 0xC2B2AE35u;
  a ^= a >> 16;
  const uint32_t b = (hi ^ hll_rotl32(a, 16)) * 0x27D4EB2Fu;  // (only its leading zeros are used)
  *a_out = a;
  *b_out = b;
}

// Per (column, block) partial written by scan_kernel; reduced in fixed order by scan_reduce_kernel.
struct ScanPartial {
  int64_t non_null;
  int64_t min_k, max_k;     // int64 values, or IEEE totalOrder keys of the doubles
  uint64_t sum_lo;          // int64 columns: 128-bit two's complement sum
  int64_t sum_hi;
  double sum, comp;         // float64 columns: two-sum compensated sum (sum + comp)
  double s1, s2;            // sum(x - pivot), sum((x - pivot)^2) over doubles
};

// Running per-column state (device resident between batches, merged on the host at finalize).
struct ScanAcc {
  int64_t total;            // rows seen
  int64_t non_null;
  int64_t min_k, max_k;
  uint64_t sum_lo;
  int64_t sum_hi;
  double sum, comp;
  // variance lanes as (n, mean, M2) so batches / ranks merge with Chan's formula
  int64_t var_n;
  double var_mean, var_m2;
  int32_t is_float;
  int32_t pad;
};

// Launch descriptors travel BY VALUE in the kernel-argument segment (no host->device copy, no
// synchronisation per batch, nothing to keep alive): at most kMaxColsPerLaunch columns per launch.
constexpr int kMaxColsPerLaunch = 24;
struct ScanLaunch {
  ScanColDesc cols[kMaxColsPerLaunch];
  int32_t acc_index[kMaxColsPerLaunch];  // running-state slot of each column
};

// Two columns scanned TOGETHER by one workgroup (scan_pair_kernel): their own aggregates as above plus the raw
// co-moments over the rows where both are non-NULL -- a COMOMENTS check whose columns also carry NUMERIC_STATS /
// COUNT / KLL checks costs no second pass over them.  Both columns share head / n_tiles (same length, offsets
// congruent modulo 64).
struct ScanPairDesc {
  ScanColDesc x, y;
  int32_t x_acc, y_acc;   // running-state slots of the two columns (-1: the column has no scan task of its own)
  int32_t como_acc;       // running-state slot of the pair's co-moments
  int32_t pad;
};
constexpr int kMaxPairsPerLaunch = 8;
struct ScanPairLaunch {
  ScanPairDesc pairs[kMaxPairsPerLaunch];
};

// Validity-only columns (COUNT(*), COUNT(col)).
struct CountColDesc {
  const uint8_t *validity;  // never nullptr here (no-validity columns are answered on the host)
  int64_t offset;
  int64_t length;
};

struct CountLaunch {
  CountColDesc cols[kMaxColsPerLaunch];
  int32_t acc_index[kMaxColsPerLaunch];
};

struct CountAcc {
  int64_t total;
  int64_t non_null;
};

// Two-column co-moments, accumulated about a per-pair PIVOT (px, py) that the first batch picks near the data
// (como_pivot_kernel): sums of (x - px), (y - py) and their products.  DataFusion's CORR / COVAR_SAMP are online
// (Welford) accumulators (TG/constraints/correlation.rs:260-275): on offset data -- timestamps, ids around 1e9 -- the
// raw sums SUM(x*x), SUM(x*y) cancel catastrophically in n*Sxy - Sx*Sy, the shifted ones do not.  The raw sums the
// CorrelationAnalyzer reports (TG/analyzers/advanced/correlation.rs:239-249) are rebuilt from the shifted ones on the
// host (tgx_finalize).
struct ComomentColDesc {
  const void *x, *y;
  const uint8_t *xv, *yv;
  int64_t xoff, yoff;
  int64_t length;
  int32_t x_is_float, y_is_float;
};

struct ComomentLaunch {
  ComomentColDesc pairs[kMaxColsPerLaunch];
  int32_t acc_index[kMaxColsPerLaunch];
};

// per (pair, block) partial of the co-moment kernels (comoments.hip, scan_pair_kernel): sums about the pair's pivot
struct ComomentPartial {
  int64_t n;
  double s[5], c[5];
};

struct ComomentAcc {
  int64_t total;
  int64_t n;
  // sums of x', y', x'x', y'y', x'y' with x' = x - px, y' = y - py, as (sum, compensation)
  double s[5], c[5];
  double px, py;       // the pivots: fixed once rows have been folded in (n > 0); (0, 0) = the raw sums
  int32_t pivot_set;
  int32_t pad;
};

// One (column, batch) window of the library-side batch coalescing (kernels/gather.hip): where the window lives and
// where it lands in the coalesced column.  A table of these is uploaded per flush; one workgroup per entry.
struct GatherSeg {
  const void *src_values;       // kind 0: first value of the window; kinds 1 / 2: its first offset (length + 1 offsets)
  const uint8_t *src_validity;  // byte holding the window's first validity bit, or nullptr (no nulls)
  const uint8_t *src_data;      // strings: the window's first value byte (offset value data_first)
  void *dst_values;             // the coalesced column's values / offsets buffer (element 0)
  uint8_t *dst_validity;        // the coalesced bitmap (zeroed before the gather), or nullptr (no segment has nulls)
  uint8_t *dst_data;            // strings: the coalesced value bytes
  int64_t src_bit0;             // bit of row 0 within *src_validity (0..7)
  int64_t length;               // rows
  int64_t dst_row;              // first row in the coalesced column
  int64_t data_first, data_base, data_len;  // strings: first source offset value, destination byte position, bytes
  int32_t elem_bytes;           // kind 0: 8 or 4
  int32_t kind;                 // 0 fixed width, 1 Utf8 (int32 offsets), 2 LargeUtf8 (int64 offsets),
                                // 3 Utf8View (16-byte views), 4 dictionary indices (int32, shifted)
  // kind 3: the stretches of the window's data buffers its long views point into (at most kGatherViewBufs buffers per
  // window): stretch k is bytes [vb_min, vb_min + vb_len) of source buffer vb_index, copied from vb_src to
  // dst_data + vb_base; a long view (buffer, offset) becomes (0, vb_base + offset - vb_min)
  int32_t vb_count;
  int32_t index_shift;          // kind 4: added to every index (where the window's dictionary starts in the coalesced one)
  int32_t vb_index[4];
  int64_t vb_min[4], vb_len[4], vb_base[4];
  const uint8_t *vb_src[4];
};
constexpr int kGatherViewBufs = 4;

// tgx_state_reset: the small per-state accumulators back to their identities with ONE launch (a copy and five fills
// were five trips through the runtime: 35 us of a 1.8 ms step)
constexpr int kResetRegions = 8;
struct StateResetArgs {
  void *zero[kResetRegions];       // regions to clear (16-byte aligned allocations)
  uint32_t zero_bytes[kResetRegions];
  int32_t n_zero;
  ScanAcc *ident;                  // the scan accumulators: set to their identity (MIN = INT64_MAX, MAX = INT64_MIN, else 0)
  uint32_t n_ident;
};

__host__ __device__ inline int64_t f64_total_key(int64_t bits) {
  return bits ^ (int64_t)(((uint64_t)(bits >> 63)) >> 1);
}

}  // namespace tgx
