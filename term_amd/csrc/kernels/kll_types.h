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
  uint32_t pad;
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

}  // namespace tgx
