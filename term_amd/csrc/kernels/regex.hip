// regex.hip -- pattern checks over Utf8 columns on gfx950.
//
//   COUNT(CASE WHEN [TRIM(]c[)] ~ 'pat' [OR c IS NULL] THEN 1 END)      TG/constraints/format.rs:762-776
//
// The host compiles the pattern (Rust `regex` syntax, unanchored search, Unicode classes expanded to
// UTF-8) into a byte DFA (regex/regex_compile.cpp).  One row per lane: the lane walks its value's bytes
// through `state = table[state * n_classes + class[byte]]` with the table held in LDS (<= 48 KiB, else it
// stays in global memory / L2), leaving early once the automaton has matched or died.  Value bytes are
// fetched as aligned 8-byte words (neighbouring lanes read neighbouring strings, so the words of a wave
// fall in a few cache lines).  Match counts are block-reduced and added with one atomic per block.
#include <hip/hip_runtime.h>

#include "regex_types.h"

namespace tgx {

typedef const uint8_t __attribute__((address_space(1))) *global_u8_ptr;
typedef const uint64_t __attribute__((address_space(1))) *global_u64_ptr;
typedef const int32_t __attribute__((address_space(1))) *global_i32_ptr;
typedef const int64_t __attribute__((address_space(1))) *global_i64_ptr;
typedef const uint16_t __attribute__((address_space(1))) *global_u16_ptr;

template <bool LDS_TABLE>
__global__ __launch_bounds__(256) void regex_match_kernel(RegexColDesc d, DfaView dfa,
                                                           unsigned long long *counters) {
  __shared__ uint16_t s_table[LDS_TABLE ? kRegexLdsEntries : 1];
  __shared__ uint8_t s_class[256];
  __shared__ unsigned long long s_part[4];
  const uint32_t n_entries = dfa.n_states * dfa.n_classes;
  if (LDS_TABLE)
    for (uint32_t i = threadIdx.x; i < n_entries; i += 256) s_table[i] = dfa.table[i];
  s_class[threadIdx.x] = dfa.byte_class[threadIdx.x];
  __syncthreads();
  global_u16_ptr g_table = (global_u16_ptr)(uintptr_t)dfa.table;
  global_u8_ptr g_acc = (global_u8_ptr)(uintptr_t)dfa.accept_end;
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  const uintptr_t data = (uintptr_t)d.data;
  const uint32_t ncls = dfa.n_classes;
  unsigned long long matches = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.length; i += stride) {
    const int64_t slot = d.offset + i;
    bool valid = true;
    if (vbits) valid = (vbits[slot >> 3] >> (slot & 7)) & 1;
    if (!valid) {
      matches += d.null_is_valid ? 1 : 0;
      continue;
    }
    int64_t b, e;
    if (d.large_offsets) {
      global_i64_ptr off = (global_i64_ptr)(uintptr_t)d.offsets;
      b = off[slot];
      e = off[slot + 1];
    } else {
      global_i32_ptr off = (global_i32_ptr)(uintptr_t)d.offsets;
      b = off[slot];
      e = off[slot + 1];
    }
    if (d.trim) {
      // SQL TRIM(col) = btrim(col, ' '): U+0020 only
      global_u8_ptr bytes = (global_u8_ptr)data;
      while (b < e && bytes[b] == 0x20) b++;
      while (e > b && bytes[e - 1] == 0x20) e--;
    }
    uint32_t st = dfa.start;
    uintptr_t p = data + (uintptr_t)b;
    const uintptr_t pe = data + (uintptr_t)e;
    while (p < pe && st > 1) {
      const uint64_t word = *(global_u64_ptr)(p & ~(uintptr_t)7);
      const uint32_t skip = (uint32_t)(p & 7);
      uint64_t w = word >> (8 * skip);
      uint32_t nb = 8 - skip;
      if (pe - p < nb) nb = (uint32_t)(pe - p);
      p += nb;
      for (uint32_t k = 0; k < nb && st > 1; k++) {
        const uint32_t c = s_class[w & 0xFF];
        w >>= 8;
        st = LDS_TABLE ? s_table[st * ncls + c] : g_table[st * ncls + c];
      }
    }
    matches += (st == 1 || g_acc[st]) ? 1 : 0;
  }
#pragma unroll
  for (int dlt = 32; dlt >= 1; dlt >>= 1) matches += __shfl_down(matches, dlt, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = matches;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    if (t) atomicAdd(&counters[0], t);
  }
}

void launch_regex(const RegexColDesc &d, const DfaView &dfa, unsigned long long *d_counters, int n_cu,
                  hipStream_t stream) {
  int64_t blocks = (d.length + 255) / 256;
  if (blocks > (int64_t)n_cu * 8) blocks = (int64_t)n_cu * 8;
  if (blocks < 1) blocks = 1;
  if ((uint64_t)dfa.n_states * dfa.n_classes <= kRegexLdsEntries)
    hipLaunchKernelGGL(regex_match_kernel<true>, dim3((int)blocks), dim3(256), 0, stream, d, dfa, d_counters);
  else
    hipLaunchKernelGGL(regex_match_kernel<false>, dim3((int)blocks), dim3(256), 0, stream, d, dfa, d_counters);
}

}  // namespace tgx
