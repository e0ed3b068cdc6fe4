// kll_host.h -- host-side KLL sketch state (product code, not the oracle).
//
// Mirrors the interface of the reference's KllSketch (TG/analyzers/advanced/kll_sketch.rs:
// new :166, merge :327-366, get_quantile :246-322, relative_error_bound :397-399): a stack of
// levels whose items weigh 2^level.  Unlike the reference's Compactor::compact (:57-76), which
// keeps the selected half in place *and* hands the other half up (so no item is ever dropped and
// the retained size grows with n), compaction here is the textbook one: the promoted half doubles
// its weight and the other half is discarded, which preserves total weight == n and keeps the
// state O(levels x capacity).  DESIGN.md "KLL" states the deviation and the error bound.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace tgx {

constexpr uint32_t kKllLevelCap = 1024;  // items a level may hold before it is compacted
constexpr uint32_t kKllRun = 512;        // device-side run length (kll.hip)

struct KllHost {
  uint32_t k = 200;  // the caller's k: only used for relative_error_bound() and merge checks
  uint64_t n = 0;    // values seen (NaN and NULL excluded)
  double min_v = std::numeric_limits<double>::infinity();
  double max_v = -std::numeric_limits<double>::infinity();
  std::vector<std::vector<double>> levels;
  uint64_t compactions = 0;  // feeds the parity choice so repeated merges do not correlate

  static uint64_t mix(uint64_t x) {
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
  }

  void ensure_level(size_t l) {
    if (levels.size() <= l) levels.resize(l + 1);
  }

  // sort level l, promote every other item of an even-sized prefix, keep the odd leftover
  void compact_level(size_t l) {
    ensure_level(l + 1);
    std::vector<double> &cur = levels[l];
    std::sort(cur.begin(), cur.end());
    size_t m = cur.size() & ~(size_t)1;
    uint64_t parity = mix(0x6b6c6cULL ^ (n * 0x9e3779b97f4a7c15ULL) ^ ((uint64_t)l << 48) ^
                          (compactions++ << 8)) & 1;
    // when the size is odd the leftover is the largest or the smallest item, alternating
    size_t first = 0;
    double leftover = 0;
    bool has_left = (cur.size() & 1) != 0;
    if (has_left) {
      if (parity) {
        leftover = cur.front();
        first = 1;
      } else {
        leftover = cur.back();
      }
    }
    uint64_t pick = (mix(parity + 0x1234567ULL + compactions) >> 17) & 1;
    std::vector<double> &up = levels[l + 1];
    for (size_t i = pick; i < m; i += 2) up.push_back(cur[first + i]);
    cur.clear();
    if (has_left) cur.push_back(leftover);
  }

  void normalize() {
    for (size_t l = 0; l < levels.size(); l++)
      while (levels[l].size() > kKllLevelCap) compact_level(l);
  }

  void add_level_items(size_t l, const double *items, size_t count) {
    ensure_level(l);
    levels[l].insert(levels[l].end(), items, items + count);
  }

  // KllSketch::merge (kll_sketch.rs:327-366); differing k is an error there, here too
  bool merge(const KllHost &o) {
    if (o.n == 0) return true;
    if (n != 0 && k != o.k) return false;
    if (n == 0) k = o.k;
    n += o.n;
    min_v = std::fmin(min_v, o.min_v);
    max_v = std::fmax(max_v, o.max_v);
    for (size_t l = 0; l < o.levels.size(); l++)
      add_level_items(l, o.levels[l].data(), o.levels[l].size());
    normalize();
    return true;
  }

  uint64_t retained() const {
    uint64_t t = 0;
    for (auto &v : levels) t += v.size();
    return t;
  }

  // KllSketch::get_quantile (kll_sketch.rs:246-322): phi=0 -> min, phi=1 -> max, otherwise the
  // first item whose cumulative weight reaches ceil(phi * total_weight).
  // returns 0 ok, 1 empty sketch, 2 phi out of range
  int quantile(double phi, double *out) const {
    if (n == 0) return 1;
    if (!(phi >= 0.0 && phi <= 1.0)) return 2;
    if (phi == 0.0) { *out = min_v; return 0; }
    if (phi == 1.0) { *out = max_v; return 0; }
    std::vector<std::pair<double, uint64_t>> items;
    items.reserve(retained());
    for (size_t l = 0; l < levels.size(); l++) {
      uint64_t w = l >= 63 ? (UINT64_MAX / 2) : (1ull << l);
      for (double v : levels[l]) items.emplace_back(v, w);
    }
    if (items.empty()) return 1;
    std::stable_sort(items.begin(), items.end(),
                     [](const std::pair<double, uint64_t> &a, const std::pair<double, uint64_t> &b) {
                       return a.first < b.first;
                     });
    uint64_t total = 0;
    for (auto &it : items) {
      uint64_t t = total + it.second;
      total = t < total ? UINT64_MAX : t;
    }
    double target = std::ceil(phi * (double)total);
    uint64_t cum = 0;
    for (auto &it : items) {
      uint64_t t = cum + it.second;
      cum = t < cum ? UINT64_MAX : t;
      if ((double)cum >= target) {
        *out = it.first;
        return 0;
      }
    }
    *out = max_v;
    return 0;
  }
};

}  // namespace tgx
