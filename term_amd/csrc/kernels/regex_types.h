// Descriptors of the pattern-match kernel (regex.hip).
#pragma once
#include <stdint.h>

namespace tgx {

constexpr uint32_t kRegexLdsEntries = 16384;  // 32 KiB of LDS for the transition table

struct RegexColDesc {
  const void *offsets;      // int32 or int64 value offsets
  const uint8_t *data;      // UTF-8 bytes
  const uint8_t *validity;  // or nullptr
  int64_t offset;           // Arrow offset
  int64_t length;
  int32_t large_offsets;    // 1 = int64 offsets (LargeUtf8)
  int32_t trim;             // TRIM(col): strip U+0020 on both ends before matching
  int32_t null_is_valid;    // NULL rows count as matches
  int32_t pad;
  uint8_t *hits;            // optional: one byte per row (1 match, 0 no match, 2 NULL row) -- dictionary columns
  const void *views;        // Utf8View: 16-byte views (then offsets / data are unused)
  const uint8_t *const *buffers;  // Utf8View: device array of the data buffers' device pointers
  // several patterns of the column in ONE walk (product automaton, DfaView::n_final > 2): pattern k counts into
  // counters[k]; bit k of null_mask = its NULL rows count as matches; hits_k as `hits`, per pattern
  int32_t n_pat;            // 0 / 1: a single pattern (null_is_valid, hits, one counter)
  uint32_t null_mask;
  uint8_t *hits_k[4];
  unsigned long long *counters_k[4];
};
constexpr int kMaxRegexGroup = 4;

// LENGTH check (kernels/regex.hip: length_kernel): inclusive character-count bounds
struct LengthBounds {
  uint64_t min_chars, max_chars;
};

struct DfaView {
  const uint16_t *table;       // n_states x n_classes
  const uint8_t *byte_class;   // 256
  const uint8_t *accept_end;   // n_states
  uint32_t n_states, n_classes, start;
  uint32_t direct;  // 1: `table` has 256 columns, one per byte value (byte_class is the identity)
  // states 0 .. n_final - 1 are decided (absorbing).  2 for one pattern (DEAD, MATCHED); 2^k for the product of k
  // patterns, where accept_end[s] is the bit mask of the patterns that match a haystack ending in s
  uint32_t n_final;
};

}  // namespace tgx
