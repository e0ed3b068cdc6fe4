// Device-side KLL sketch layout (see kll.hip).
#pragma once
#include <stdint.h>

namespace tgx {

constexpr int kKllMaxLevels = 56;   // level l holds weight-2^l items; 512 * 2^55 rows is out of reach
constexpr int kKllRunItems = 512;   // items of a level >= 1 run
constexpr int kKllLv0Cap = 1024;    // level 0 holds < 1024 raw items

struct KllDeviceSketch {
  unsigned long long n;  // values sketched (NULL / NaN excluded)
  double min_v, max_v;
  uint64_t level_mask;   // bit l: runs[l] is occupied
  uint32_t lv0_count;
  // a sketch of pre-sampled values (the picks of the fused scan, each standing for 2^shift stream items): its loose
  // items weigh 2^shift, its runs sit `shift` levels higher, n counts stream items.  0 for sketches of raw values.
  uint32_t shift;
  double lv0[kKllLv0Cap];
  double runs[kKllMaxLevels][kKllRunItems];  // runs[0] unused
};

struct KllColDesc {
  const void *values;
  const uint8_t *validity;
  int64_t offset;
  int64_t length;
  int32_t is_float;
  int32_t pad;
};

// Several sketching jobs in one launch sequence (grid.y = job): the columns of a batch -- and, on the fused path, the
// picks and the leftovers of each -- go through build / tree rounds / fold side by side instead of one after the other
// (a tree round is a handful of workgroups: latency, not work).
struct KllJob {
  KllColDesc d;
  int64_t chunk;                 // rows per workgroup of the build
  struct KllDeviceSketch *sketches;  // scratch: one sketch per workgroup
  struct KllDeviceSketch *state;     // the running sketch the batch is folded into
  uint64_t salt;
  uint32_t top;                  // sampling level of the build (0: every value is kept)
  uint32_t shift;                // the input values are pre-sampled items of weight 2^shift
  int32_t groups;                // workgroups of the build
  int32_t pad;
};
constexpr int kKllMaxJobs = 16;
struct KllJobs {
  KllJob job[kKllMaxJobs];
};

// ---- the KLL sampler riding on the numeric scan (scan.hip): what one wave of the scan leaves behind for a column.
// A wave's rows are a stream of their own; it cuts its non-NULL, non-NaN values into groups of 2^top consecutive
// values and keeps ONE member of each, chosen by a counter hash (Karnin-Lang-Liberty sec. 3.2: an item of weight 2^l
// below the lowest kept level is one uniformly chosen member of 2^l consecutive stream items).
//   picks[wave * cap + g]        the chosen member of the wave's group g (weight 2^top), NaN past its last group
//   left[wave * 2^top + i]       the < 2^top values after its last complete group (weight 1), NaN-padded
//   meta[wave]                   values taken, NaN-ignoring min / max
// so total weight = sum over waves of (groups * 2^top + leftovers) = the number of values, exactly.  Both arrays are
// then sketched by kll_build_kernel (it drops NaN): the picks with shift = top, the leftovers with shift = 0.
struct KllWaveMeta {
  unsigned long long count;
  double min_v, max_v;
};
struct ScanKll {
  double *picks;      // nullptr: no KLL on this column
  double *left;
  KllWaveMeta *meta;
  uint64_t salt;
  int32_t top;        // 1 .. kScanKllMaxTop
  int32_t cap;        // picks per wave
};
constexpr int kScanKllMaxTop = 8;

}  // namespace tgx
