// regex.hip -- pattern checks over Utf8 columns on gfx950.
//
//   COUNT(CASE WHEN [TRIM(]c[)] ~ 'pat' [OR c IS NULL] THEN 1 END)      TG/constraints/format.rs:762-776
//
// The host compiles the pattern (Rust `regex` syntax, unanchored search, Unicode classes expanded to
// UTF-8) into a byte DFA (regex/regex_compile.cpp).  One row per lane: the lane walks its value's bytes
// through `state = table[state * n_classes + class[byte]]` with the table held in LDS (<= 48 KiB, else it
// stays in global memory / L2), leaving early once the automaton has matched or died.  A wave takes 64
// consecutive rows, whose value bytes are one contiguous span: the span is copied into LDS with coalesced
// 16-byte loads and each lane then reads its own value from LDS as aligned 8-byte words (per-lane global loads
// at a ~28-byte stride ran the kernel at 1.2 TB/s, the staged form at 2.3 TB/s); spans longer than 4 KiB fall
// back to per-lane global reads.  Evaluating several patterns of one column in the same pass was tried and was
// slower than one pass each (6.6 ms vs 4.6 ms for three patterns on 100 M rows): the per-byte dependent LDS
// lookups, not the reads, bound the kernel.  Match counts are block-reduced: one atomic per block.
#include <hip/hip_runtime.h>

#include "regex_types.h"

namespace tgx {

typedef const uint8_t __attribute__((address_space(1))) *global_u8_ptr;
typedef const uint64_t __attribute__((address_space(1))) *global_u64_ptr;
typedef const int32_t __attribute__((address_space(1))) *global_i32_ptr;
typedef const int64_t __attribute__((address_space(1))) *global_i64_ptr;
typedef const uint16_t __attribute__((address_space(1))) *global_u16_ptr;

constexpr uint32_t kStageBytes = 4096;  // LDS staging per wave: 64 consecutive values up to 64 B on average

// walks value bytes [b, e) (absolute offsets into `data`) through the automaton; STAGED means the bytes
// [stage_base, ...) are already in LDS (16-byte aligned image of the wave's contiguous value span)
template <bool LDS_TABLE, bool STAGED, bool DIRECT>
__device__ __forceinline__ uint32_t walk(const DfaView &dfa, const uint16_t *s_table, const uint8_t *s_class,
                                         uintptr_t data, int64_t b, int64_t e, const uint8_t *stage,
                                         int64_t stage_base) {
  global_u16_ptr g_table = (global_u16_ptr)(uintptr_t)dfa.table;
  const uint32_t ncls = dfa.n_classes;
  uint32_t st = dfa.start;
  int64_t p = b;
  while (p < e && st > 1) {
    uint64_t word;
    if (STAGED) {
      word = *(const uint64_t *)(stage + ((p - stage_base) & ~(int64_t)7));
    } else {
      word = *(global_u64_ptr)((data + (uintptr_t)p) & ~(uintptr_t)7);
    }
    const uint32_t skip = STAGED ? (uint32_t)((p - stage_base) & 7) : (uint32_t)((data + (uintptr_t)p) & 7);
    uint64_t w = word >> (8 * skip);
    uint32_t nb = 8 - skip;
    if (e - p < (int64_t)nb) nb = (uint32_t)(e - p);
    p += nb;
    for (uint32_t k = 0; k < nb && st > 1; k++) {
      // DIRECT: the table has one column per BYTE (small automata), so a step is one LDS lookup instead of two
      const uint32_t c = DIRECT ? (uint32_t)(w & 0xFF) : s_class[w & 0xFF];
      w >>= 8;
      st = DIRECT ? s_table[(st << 8) + c] : LDS_TABLE ? s_table[st * ncls + c] : g_table[st * ncls + c];
    }
  }
  return st;
}

// two values per lane walked in lockstep: the two chains of dependent LDS lookups interleave (the walk is bound by
// lookup latency, not by issue), so the second value comes almost for free
template <bool LDS_TABLE, bool STAGED, bool DIRECT>
__device__ __forceinline__ void walk2(const DfaView &dfa, const uint16_t *s_table, const uint8_t *s_class,
                                      uintptr_t data0, int64_t b0, int64_t e0, uintptr_t data1, int64_t b1,
                                      int64_t e1, const uint8_t *stage, int64_t stage_base, uint32_t *out0,
                                      uint32_t *out1) {
  global_u16_ptr g_table = (global_u16_ptr)(uintptr_t)dfa.table;
  const uint32_t ncls = dfa.n_classes;
  uint32_t st0 = dfa.start, st1 = dfa.start;
  int64_t p0 = b0, p1 = b1;
  auto load = [&](uintptr_t data, int64_t p, uint32_t *skip) -> uint64_t {
    if (STAGED) {
      *skip = (uint32_t)((p - stage_base) & 7);
      return *(const uint64_t *)(stage + ((p - stage_base) & ~(int64_t)7));
    }
    *skip = (uint32_t)((data + (uintptr_t)p) & 7);
    return *(global_u64_ptr)((data + (uintptr_t)p) & ~(uintptr_t)7);
  };
  auto step = [&](uint32_t st, uint64_t &w) -> uint32_t {
    const uint32_t c = DIRECT ? (uint32_t)(w & 0xFF) : s_class[w & 0xFF];
    w >>= 8;
    return DIRECT ? s_table[(st << 8) + c] : LDS_TABLE ? s_table[st * ncls + c] : g_table[st * ncls + c];
  };
  for (;;) {
    const bool a0 = p0 < e0 && st0 > 1, a1 = p1 < e1 && st1 > 1;
    if (!a0 && !a1) break;
    uint32_t skip0 = 0, skip1 = 0, nb0 = 0, nb1 = 0;
    uint64_t w0 = 0, w1 = 0;
    if (a0) {
      w0 = load(data0, p0, &skip0) >> (8 * skip0);
      nb0 = 8 - skip0;
      if (e0 - p0 < (int64_t)nb0) nb0 = (uint32_t)(e0 - p0);
      p0 += nb0;
    }
    if (a1) {
      w1 = load(data1, p1, &skip1) >> (8 * skip1);
      nb1 = 8 - skip1;
      if (e1 - p1 < (int64_t)nb1) nb1 = (uint32_t)(e1 - p1);
      p1 += nb1;
    }
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      if (k < nb0 && st0 > 1) st0 = step(st0, w0);
      if (k < nb1 && st1 > 1) st1 = step(st1, w1);
    }
  }
  *out0 = st0;
  *out1 = st1;
}

// TABLE_ENTRIES: LDS budget of the transition table (0 = the table stays in global memory / L2).  The small
// instance leaves room for 6 workgroups per CU, the large one for 3.
template <int TABLE_ENTRIES, bool DIRECT = false, bool VIEW = false>
__global__ __launch_bounds__(256) void regex_match_kernel(RegexColDesc d, DfaView dfa,
                                                           unsigned long long *counters) {
  constexpr bool LDS_TABLE = TABLE_ENTRIES > 0;
  __shared__ uint16_t s_table[LDS_TABLE ? TABLE_ENTRIES : 1];
  __shared__ uint8_t s_class[256];
  __shared__ __attribute__((aligned(16))) uint8_t s_stage[4][kStageBytes + 32];
  __shared__ unsigned long long s_part[4];
  const uint32_t n_entries = dfa.n_states * dfa.n_classes;
  if (LDS_TABLE)
    for (uint32_t i = threadIdx.x; i < n_entries; i += 256) s_table[i] = dfa.table[i];
  s_class[threadIdx.x] = dfa.byte_class[threadIdx.x];
  __syncthreads();
  global_u8_ptr g_acc = (global_u8_ptr)(uintptr_t)dfa.accept_end;
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  const uintptr_t data0 = (uintptr_t)d.data;
  constexpr bool is_view = VIEW;  // Utf8View columns run their own instance: no data-dependent branches in the
                                  // offset path, so its loads are all issued before the first wait
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t *stage = s_stage[wave];
  unsigned long long matches = 0;
  // A wave step takes 128 consecutive rows, two per lane (rows `lane` and `lane + 64` of the step): twice the bytes
  // in flight for the same LDS, and the two values of a lane are walked in lockstep.  When the 128 values do not
  // fit the stage the two 64-row halves are staged one after the other; a half that still does not fit is read
  // straight from global memory.
  const int64_t n_groups = (d.length + 127) / 128;
  // row descriptor: value bytes [b, e) relative to `data` (raw offsets, also for NULL rows), validity
  struct Row {
    int64_t b, e;
    uintptr_t data;
    bool valid, in;
  };
  // What a step requests up front.  Offsets: ONE load per row -- a row's end is the next row's start, fetched from
  // the neighbouring lane when the step is used (rows past the end of the column read the end offset, so they
  // come out empty), plus the step's end offset (`tail`, one address for the whole wave).
  struct Step {
    int64_t b0, b1, tail;  // !VIEW: raw offsets
    uint32_t vb0, vb1;     // validity bytes
    Row v0, v1;            // VIEW: complete rows
  };
  auto offset_at = [&](int64_t row) -> int64_t {
    const int64_t slot = d.offset + (row < d.length ? row : d.length);
    return d.large_offsets ? ((global_i64_ptr)(uintptr_t)d.offsets)[slot]
                           : (int64_t)((global_i32_ptr)(uintptr_t)d.offsets)[slot];
  };
  auto view_row = [&](int64_t i) -> Row {
    // {length, inline bytes | prefix, buffer, offset}: the value is wherever the view says (no common span).
    // The 16 view bytes are always readable; what they SAY is only trusted for valid rows (the view of a NULL
    // slot is arbitrary)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef const i32x4 __attribute__((address_space(1))) *gi32x4;
    Row r;
    r.in = i < d.length;
    const int64_t slot = d.offset + (r.in ? i : d.length - 1);
    const uint32_t vbyte = vbits ? (uint32_t)vbits[slot >> 3] : 0xFFu;
    const i32x4 vw = *(gi32x4)((uintptr_t)d.views + (uintptr_t)slot * 16);
    r.valid = r.in && ((vbyte >> (slot & 7)) & 1);
    const int32_t len = r.valid ? vw.x : 0;
    r.b = 0;
    r.e = len;
    if (len <= 12)
      r.data = (uintptr_t)d.views + (uintptr_t)slot * 16 + 4;
    else
      r.data = (uintptr_t)d.buffers[vw.z] + (uintptr_t)(uint32_t)vw.w;
    return r;
  };
  auto fetch = [&](int64_t g) -> Step {
    Step s;
    const int64_t i0 = g * 128 + lane, i1 = i0 + 64;
    if (is_view) {
      s.v0 = view_row(i0);
      s.v1 = view_row(i1);
      return s;
    }
    const int64_t s0 = d.offset + (i0 < d.length ? i0 : d.length - 1), s1 = d.offset + (i1 < d.length ? i1 : d.length - 1);
    s.vb0 = vbits ? (uint32_t)vbits[s0 >> 3] : 0xFFu;
    s.vb1 = vbits ? (uint32_t)vbits[s1 >> 3] : 0xFFu;
    s.b0 = offset_at(i0);
    s.b1 = offset_at(i1);
    s.tail = offset_at(g * 128 + 128);
    return s;
  };
  auto rows_of = [&](const Step &s, int64_t g, Row *r0, Row *r1) {
    if (is_view) {
      *r0 = s.v0;
      *r1 = s.v1;
      return;
    }
    const int64_t i0 = g * 128 + lane, i1 = i0 + 64;
    const int64_t next0 = __shfl_down(s.b0, 1, 64), next1 = __shfl_down(s.b1, 1, 64), first1 = __shfl(s.b1, 0, 64);
    r0->b = s.b0;
    r0->e = lane < 63 ? next0 : first1;
    r1->b = s.b1;
    r1->e = lane < 63 ? next1 : s.tail;
    r0->data = r1->data = data0;
    r0->in = i0 < d.length;
    r1->in = i1 < d.length;
    const int64_t s0 = d.offset + (r0->in ? i0 : d.length - 1), s1 = d.offset + (r1->in ? i1 : d.length - 1);
    r0->valid = r0->in && ((s.vb0 >> (s0 & 7)) & 1);
    r1->valid = r1->in && ((s.vb1 >> (s1 & 7)) & 1);
  };
  // copies bytes [base, span_e) of the value buffer into the wave's stage.  16-byte blocks by ABSOLUTE address: a
  // block that holds one byte of the buffer lies in the same page, so the rounded-out copy cannot fault whatever
  // the alignment of `data` (values of NULL rows are copied too; never interpreted)
  auto stage_in = [&](int64_t base, int64_t span_e) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef const u32x4 __attribute__((address_space(1))) *global_u4_ptr;
    global_u4_ptr src = (global_u4_ptr)(data0 + (uintptr_t)base);
    const int64_t n16 = (span_e - base + 15) >> 4;
    for (int64_t k = lane; k < n16; k += 64) *(u32x4 *)(stage + 16 * k) = src[k];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };
  auto stage_done = [&]() {  // every lane is done with the stage: the next copy may overwrite it
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto align16 = [&](int64_t span_b) -> int64_t { return span_b - (int64_t)((data0 + (uintptr_t)span_b) & 15); };
  // walk bounds of a row: NULL rows walk nothing; SQL TRIM(col) = btrim(col, ' '): U+0020 only
  auto bounds = [&](const Row &r, bool staged, int64_t base, int64_t *wb, int64_t *we) {
    int64_t b = r.b, e = r.valid ? r.e : r.b;
    if (d.trim) {
      if (staged) {
        while (b < e && stage[b - base] == 0x20) b++;
        while (e > b && stage[e - 1 - base] == 0x20) e--;
      } else {
        global_u8_ptr bytes = (global_u8_ptr)r.data;
        while (b < e && bytes[b] == 0x20) b++;
        while (e > b && bytes[e - 1] == 0x20) e--;
      }
    }
    *wb = b;
    *we = e;
  };
  auto account = [&](const Row &r, int64_t i, uint32_t st) {
    if (r.valid) {
      const bool hit = st == 1 || g_acc[st];
      matches += hit ? 1 : 0;
      if (d.hits) d.hits[i] = hit ? 1 : 0;
    } else if (r.in) {
      matches += d.null_is_valid ? 1 : 0;
      if (d.hits) d.hits[i] = 2;
    }
  };
  const int64_t g_stride = (int64_t)gridDim.x * 4;
  int64_t g = (int64_t)blockIdx.x * 4 + wave;
  Step nxt = fetch(g < n_groups ? g : 0);
  for (; g < n_groups; g += g_stride) {
    // the offsets / validity of the NEXT step are requested before this step's bytes are staged and walked
    const Step cur = nxt;
    if (g + g_stride < n_groups) nxt = fetch(g + g_stride);
    Row r0, r1;
    rows_of(cur, g, &r0, &r1);
    const int64_t i0 = g * 128 + lane, i1 = i0 + 64;
    uint32_t st0 = 0, st1 = 0;
    int64_t wb0, we0, wb1, we1;
    if (is_view) {
      bounds(r0, false, 0, &wb0, &we0);
      bounds(r1, false, 0, &wb1, &we1);
      walk2<LDS_TABLE, false, DIRECT>(dfa, s_table, s_class, r0.data, wb0, we0, r1.data, wb1, we1, nullptr, 0, &st0, &st1);
    } else {
      // the step's values are contiguous: [b of its first row, e of its last)
      const int64_t b_first = __shfl(r0.b, 0, 64), e_half = __shfl(r0.e, 63, 64), e_last = __shfl(r1.e, 63, 64);
      const int64_t base = align16(b_first);
      if (e_last - base <= (int64_t)kStageBytes) {  // wave-uniform
        stage_in(base, e_last);
        bounds(r0, true, base, &wb0, &we0);
        bounds(r1, true, base, &wb1, &we1);
        walk2<LDS_TABLE, true, DIRECT>(dfa, s_table, s_class, data0, wb0, we0, data0, wb1, we1, stage, base, &st0, &st1);
        stage_done();
      } else {
        const bool fit0 = e_half - base <= (int64_t)kStageBytes;
        if (fit0) stage_in(base, e_half);
        bounds(r0, fit0, base, &wb0, &we0);
        st0 = fit0 ? walk<LDS_TABLE, true, DIRECT>(dfa, s_table, s_class, data0, wb0, we0, stage, base)
                   : walk<LDS_TABLE, false, DIRECT>(dfa, s_table, s_class, data0, wb0, we0, nullptr, 0);
        if (fit0) stage_done();
        const int64_t base1 = align16(__shfl(r1.b, 0, 64));
        const bool fit1 = e_last - base1 <= (int64_t)kStageBytes;
        if (fit1) stage_in(base1, e_last);
        bounds(r1, fit1, base1, &wb1, &we1);
        st1 = fit1 ? walk<LDS_TABLE, true, DIRECT>(dfa, s_table, s_class, data0, wb1, we1, stage, base1)
                   : walk<LDS_TABLE, false, DIRECT>(dfa, s_table, s_class, data0, wb1, we1, nullptr, 0);
        if (fit1) stage_done();
      }
    }
    account(r0, i0, st0);
    account(r1, i1, st1);
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

// ---- LENGTH(col) BETWEEN min AND max, counted per row; NULL rows always count (TG/constraints/length.rs:167-171).
// LENGTH is the number of code points = bytes that are not UTF-8 continuation bytes (10xxxxxx).  Most rows are
// decided by their BYTE length alone (chars <= bytes, chars >= ceil(bytes / 4)); only the undecided ones read
// their bytes, eight at a time.
__global__ __launch_bounds__(256) void length_kernel(RegexColDesc d, LengthBounds lb, unsigned long long *counters) {
  __shared__ unsigned long long s_part[4];
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  const bool is_view = d.views != nullptr;
  unsigned long long matches = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.length; i += (int64_t)gridDim.x * 256) {
    const int64_t slot = d.offset + i;
    bool valid = true;
    if (vbits) valid = (vbits[slot >> 3] >> (slot & 7)) & 1;
    if (!valid) {
      matches += 1;
      if (d.hits) d.hits[i] = 2;
      continue;
    }
    uintptr_t p;
    uint64_t nbytes;
    if (is_view) {
      typedef const int32_t __attribute__((address_space(1))) *gi32;
      gi32 vw = (gi32)((uintptr_t)d.views + (uintptr_t)slot * 16);
      const int32_t len = vw[0];
      nbytes = (uint64_t)len;
      p = len <= 12 ? (uintptr_t)d.views + (uintptr_t)slot * 16 + 4
                    : (uintptr_t)d.buffers[vw[2]] + (uintptr_t)(uint32_t)vw[3];
    } else if (d.large_offsets) {
      global_i64_ptr off = (global_i64_ptr)(uintptr_t)d.offsets;
      const int64_t b = off[slot];
      nbytes = (uint64_t)(off[slot + 1] - b);
      p = (uintptr_t)d.data + (uintptr_t)b;
    } else {
      global_i32_ptr off = (global_i32_ptr)(uintptr_t)d.offsets;
      const int32_t b = off[slot];
      nbytes = (uint64_t)(off[slot + 1] - b);
      p = (uintptr_t)d.data + (uintptr_t)b;
    }
    bool ok;
    if (nbytes < lb.min_chars) {
      ok = false;  // chars <= bytes < min
    } else if ((nbytes + 3) / 4 > lb.max_chars) {
      ok = false;  // chars >= ceil(bytes / 4) > max
    } else if (nbytes <= lb.max_chars && (nbytes + 3) / 4 >= lb.min_chars) {
      ok = true;   // every possible char count of this byte length is inside the bounds
    } else {
      // count the continuation bytes: logical 8-byte words of the value assembled from aligned loads
      uint64_t cont = 0, remaining = nbytes;
      uintptr_t q = p;
      while (remaining > 0) {
        const uint32_t nb = remaining < 8 ? (uint32_t)remaining : 8u;
        const uint32_t skip = (uint32_t)(q & 7);
        const uintptr_t base = q & ~(uintptr_t)7;
        uint64_t w = *(global_u64_ptr)base >> (8 * skip);
        if (skip + nb > 8) w |= *(global_u64_ptr)(base + 8) << (8 * (8 - skip));
        if (nb < 8) w &= (1ull << (8 * nb)) - 1;
        // a byte is 10xxxxxx iff bit 7 set and bit 6 clear
        cont += __builtin_popcountll(w & ~(w << 1) & 0x8080808080808080ULL);
        q += nb;
        remaining -= nb;
      }
      const uint64_t chars = nbytes - cont;
      ok = chars >= lb.min_chars && chars <= lb.max_chars;
    }
    matches += ok ? 1 : 0;
    if (d.hits) d.hits[i] = ok ? 1 : 0;
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

void launch_length(const RegexColDesc &d, const LengthBounds &lb, unsigned long long *d_counters, int n_cu,
                   hipStream_t stream) {
  int64_t blocks = (d.length + 255) / 256;
  if (blocks > (int64_t)n_cu * 8) blocks = (int64_t)n_cu * 8;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(length_kernel, dim3((int)blocks), dim3(256), 0, stream, d, lb, d_counters);
}

void launch_regex(const RegexColDesc &d, const DfaView &dfa, unsigned long long *d_counters, int n_cu,
                  hipStream_t stream) {
  int64_t blocks = (d.length + 511) / 512;  // 128 rows per wave step, four waves
  if (blocks > (int64_t)n_cu * 6) blocks = (int64_t)n_cu * 6;
  if (blocks < 1) blocks = 1;
  const uint64_t entries = (uint64_t)dfa.n_states * dfa.n_classes;
  const dim3 grid((int)blocks), block(256);
#define TGX_RX(ENTRIES, DIRECT)                                                                                      \
  do {                                                                                                               \
    if (d.views)                                                                                                     \
      hipLaunchKernelGGL((regex_match_kernel<ENTRIES, DIRECT, true>), grid, block, 0, stream, d, dfa, d_counters);   \
    else                                                                                                             \
      hipLaunchKernelGGL((regex_match_kernel<ENTRIES, DIRECT, false>), grid, block, 0, stream, d, dfa, d_counters);  \
  } while (0)
  if (dfa.n_classes == 256 && dfa.direct && entries <= 4096)
    TGX_RX(4096, true);
  else if (entries <= 4096)
    TGX_RX(4096, false);
  else if (entries <= kRegexLdsEntries)
    TGX_RX((int)kRegexLdsEntries, false);
  else
    TGX_RX(0, false);
#undef TGX_RX
}

}  // namespace tgx
