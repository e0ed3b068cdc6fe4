// regex.hip -- pattern checks over Utf8 columns on gfx950.
//
//   COUNT(CASE WHEN [TRIM(]c[)] ~ 'pat' [OR c IS NULL] THEN 1 END)      TG/constraints/format.rs:762-776
//
// The host compiles the pattern (Rust `regex` syntax, unanchored search, Unicode classes expanded to
// UTF-8) into a byte DFA (regex/regex_compile.cpp).  Rows are walked one per lane, two rows per lane in lockstep:
// `state = table[state][class[byte]]` with table and class table held in LDS (<= 32 KiB of table, else it stays in
// global memory / L2).  A wave step takes 128 consecutive rows, whose value bytes are one contiguous span: the span
// is copied into LDS with coalesced 16-byte loads and each lane reads its own values from there (per-lane global
// loads at a ~28-byte stride ran the kernel at 1.2 TB/s, the staged form at 2.3 TB/s); spans longer than 4 KiB
// fall back to per-lane global reads.  What the counters said, in the order it was found (100 M rows x 28 B):
//   * ~10 VALU instructions per byte and chain (tests per byte, 64-bit shifts)            1.29 ms
//   * prescaled table + absorbing final states + whole-chunk loop (Tbl, walk2_staged): 3-4 instructions per byte,
//     but every chunk was an UNALIGNED ds_read_b64: 62 LDS cycles per wave instruction
//     (SQ_LDS_UNALIGNED_STALL = 60 % of SQ_LDS_IDX_ACTIVE)                                  1.37 ms
//   * chunks cut from aligned words (ChunkFeed)                                             1.02 ms
//   * LDS sized per launch, grid = exactly the resident workgroups (7 per CU)              0.86-0.93 ms
// VALU (~55 %), LDS (~45 %) and HBM (~65 % of the achievable rate) are now about equally loaded.  Several patterns of
// one column share ONE walk (MULTI, the product automaton); round 5 measured workgroups of 8 / 12 / 14 waves sharing
// one table for big automata: never better than four-wave workgroups at the right count per CU.  Match
// counts are block-reduced: one atomic per block.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

#include "regex_types.h"

namespace tgx {

typedef const uint8_t __attribute__((address_space(1))) *global_u8_ptr;
typedef const uint64_t __attribute__((address_space(1))) *global_u64_ptr;
typedef const int32_t __attribute__((address_space(1))) *global_i32_ptr;
typedef const int64_t __attribute__((address_space(1))) *global_i64_ptr;
typedef const uint16_t __attribute__((address_space(1))) *global_u16_ptr;

constexpr uint32_t kStageBytes = 4096;  // LDS staging per wave: 64 consecutive values up to 64 B on average

// Table formats.  In global memory (DfaView) an entry is the next STATE (0 = dead, 1 = matched, both final).  The
// copy a workgroup keeps in LDS is PRESCALED: an entry is the BYTE OFFSET of the next state's row (state x row
// bytes), the class table holds class x 2, and the rows of the two final states are made absorbing -- so a step is
// `bfe; add; ds_read_u16` with no test per byte (the walk was bound by instruction issue, ~10 VALU per byte, not by
// HBM or LDS).  `Tbl` hides the difference; `term` is the largest final value (1, or one row).
// LDS layout (dynamic allocation, FIXED offsets for everything a byte step touches, so the addresses fold into the
// ds_read offset field): class table, the four waves' stages, then this automaton's table and accept flags.
constexpr uint32_t kLdsClassOff = 0;                                       // u16[256] (class x 2) / u8[256]
constexpr uint32_t kLdsStageOff = 512;                                     // 4 x (kStageBytes + 32)
constexpr uint32_t kLdsTableOff = kLdsStageOff + 4 * (kStageBytes + 32);  // u16[n_states x n_classes]

template <bool LDS_TABLE, bool DIRECT>
struct Tbl {
  const uint8_t *lds;  // base of the dynamic LDS allocation
  global_u16_ptr g_table;
  uint32_t ncls, row_bytes, start, term;
  __device__ __forceinline__ uint32_t step(uint32_t st, uint32_t byte) const {
    if (LDS_TABLE) {
      // ABSOLUTE LDS addresses: the kernel has no static LDS, so its dynamic block starts at LDS address 0 (checked
      // at kernel entry) and class / table offsets become the immediate offset of the ds_read -- through the
      // `extern __shared__` symbol every lookup paid an extra add of the (link-time) base
      typedef const uint16_t __attribute__((address_space(3))) *lds_u16_ptr;
      const uint32_t c2 = DIRECT ? byte << 1 : (uint32_t) * (lds_u16_ptr)(uintptr_t)(kLdsClassOff + (byte << 1));
      return *(lds_u16_ptr)(uintptr_t)(kLdsTableOff + st + c2);
    }
    return g_table[st * ncls + lds[kLdsClassOff + byte]];
  }
  __device__ __forceinline__ uint32_t state_of(uint32_t st) const { return LDS_TABLE ? st / row_bytes : st; }
};

// walks value bytes [b, e) (absolute offsets into `data`) through the automaton, one aligned 8-byte word at a time
// (a word that holds a byte of the buffer lies in the buffer's pages); STAGED means the bytes [stage_base, ...) are
// already in LDS (16-byte aligned image of the wave's contiguous value span)
template <bool LDS_TABLE, bool STAGED, bool DIRECT>
__device__ __forceinline__ uint32_t walk(const Tbl<LDS_TABLE, DIRECT> &t, uintptr_t data, int64_t b, int64_t e,
                                         const uint8_t *stage, int64_t stage_base) {
  uint32_t st = t.start;
  int64_t p = b;
  while (p < e && st > t.term) {
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
    for (uint32_t k = 0; k < nb && st > t.term; k++) {
      st = t.step(st, (uint32_t)(w & 0xFF));
      w >>= 8;
    }
  }
  return st;
}

// two values per lane walked in lockstep from global memory (Utf8View rows, values that do not fit the stage)
template <bool LDS_TABLE, bool DIRECT>
__device__ __forceinline__ void walk2(const Tbl<LDS_TABLE, DIRECT> &t, uintptr_t data0, int64_t b0, int64_t e0,
                                      uintptr_t data1, int64_t b1, int64_t e1, uint32_t *out0, uint32_t *out1) {
  uint32_t st0 = t.start, st1 = t.start;
  int64_t p0 = b0, p1 = b1;
  auto load = [&](uintptr_t data, int64_t p, uint32_t *skip) -> uint64_t {
    *skip = (uint32_t)((data + (uintptr_t)p) & 7);
    return *(global_u64_ptr)((data + (uintptr_t)p) & ~(uintptr_t)7);
  };
  for (;;) {
    const bool a0 = p0 < e0 && st0 > t.term, a1 = p1 < e1 && st1 > t.term;
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
      if (k < nb0 && st0 > t.term) st0 = t.step(st0, (uint32_t)(w0 >> (8 * k)) & 0xFF);
      if (k < nb1 && st1 > t.term) st1 = t.step(st1, (uint32_t)(w1 >> (8 * k)) & 0xFF);
    }
  }
  *out0 = st0;
  *out1 = st1;
}

// The fast path: two STAGED values per lane, prescaled LDS table.  Value bytes are taken eight at a time from the
// value's own start, so only a value's LAST chunk is partial: loop A runs the whole chunks with three or four
// instructions per byte and chain (the final states absorb, a chain that is out of whole chunks is kept by a select
// per chunk), loop B the 0..7 tail bytes with a select per byte.  Both chains always execute, interleaved -- no
// divergent blocks inside the loop.
// A chunk is cut out of ALIGNED 8-byte LDS words (one new word per chunk, funnel-shifted against the previous
// one): an unaligned `ds_read_b64` is served lane by lane -- SQ_LDS_UNALIGNED_STALL showed 62 LDS cycles per such
// wave instruction, 60 % of all LDS cycles of the kernel.  The stage has 32 bytes of slack for the read-ahead.
struct ChunkFeed {
  uint32_t a;       // LDS offset of the next aligned word
  uint32_t s;       // byte shift of the value inside its first word (0..7)
  uint32_t c0, c1;  // the current aligned word
  __device__ __forceinline__ void start(const uint8_t *stage, uint32_t o) {
    a = o & ~7u;
    s = o & 7u;
    const uint64_t w = *(const uint64_t *)(stage + a);
    c0 = (uint32_t)w;
    c1 = (uint32_t)(w >> 32);
    a += 8;
  }
  // the next eight value bytes (lo, hi).  The feed always moves on -- a chain that has run out of whole chunks
  // keeps reading (and discarding) what follows its value; `limit` keeps that inside the stage
  __device__ __forceinline__ void next(const uint8_t *stage, uint32_t limit, uint32_t *lo, uint32_t *hi) {
    const uint64_t w = *(const uint64_t *)(stage + a);
    const uint32_t n0 = (uint32_t)w, n1 = (uint32_t)(w >> 32);
    const bool up = (s & 4) != 0;
    const uint32_t x0 = up ? c1 : c0, x1 = up ? n0 : c1, x2 = up ? n1 : n0;
    *lo = __builtin_amdgcn_alignbyte(x1, x0, s);  // shift by s & 3 bytes
    *hi = __builtin_amdgcn_alignbyte(x2, x1, s);
    c0 = n0;
    c1 = n1;
    a = a + 8 < limit ? a + 8 : limit;
  }
};

template <bool DIRECT>
__device__ __forceinline__ void walk2_staged(const Tbl<true, DIRECT> &t, const uint8_t *stage, uint32_t o0,
                                             uint32_t n0, uint32_t o1, uint32_t n1, uint32_t *out0, uint32_t *out1) {
  uint32_t st0 = t.start, st1 = t.start;
  const uint32_t full0 = n0 >> 3, full1 = n1 >> 3;
  const uint32_t limit = kStageBytes + 16;  // last aligned word of the stage (+ 32 bytes of slack)
  ChunkFeed f0, f1;
  f0.start(stage, o0);
  f1.start(stage, o1);
  for (uint32_t it = 0;; it++) {  // `it` is wave-uniform
    const bool a0 = it < full0, a1 = it < full1;
    if (!((a0 && st0 > t.term) || (a1 && st1 > t.term))) break;
    uint32_t l0, h0, l1, h1;
    f0.next(stage, limit, &l0, &h0);
    f1.next(stage, limit, &l1, &h1);
    uint32_t x0 = st0, x1 = st1;
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      x0 = t.step(x0, ((k < 4 ? l0 : h0) >> (8 * (k & 3))) & 0xFF);
      x1 = t.step(x1, ((k < 4 ? l1 : h1) >> (8 * (k & 3))) & 0xFF);
    }
    st0 = a0 ? x0 : st0;
    st1 = a1 ? x1 : st1;
  }
  // the 0..7 tail bytes of the chains that are still undecided (final states need no more input)
  const uint32_t r0 = st0 > t.term ? n0 & 7 : 0, r1 = st1 > t.term ? n1 & 7 : 0;
  uint32_t rmax = r0 > r1 ? r0 : r1;
#pragma unroll
  for (int dlt = 32; dlt >= 1; dlt >>= 1) {
    const uint32_t o = __shfl_xor(rmax, dlt, 64);
    rmax = o > rmax ? o : rmax;
  }
  rmax = __builtin_amdgcn_readfirstlane(rmax);
  if (rmax) {
    uint32_t l0, h0, l1, h1;
    f0.start(stage, o0 + 8 * full0);
    f1.start(stage, o1 + 8 * full1);
    f0.next(stage, limit, &l0, &h0);
    f1.next(stage, limit, &l1, &h1);
#pragma unroll
    for (uint32_t k = 0; k < 7; k++) {
      if (k < rmax) {  // wave-uniform
        const uint32_t x0 = t.step(st0, ((k < 4 ? l0 : h0) >> (8 * (k & 3))) & 0xFF);
        const uint32_t x1 = t.step(st1, ((k < 4 ? l1 : h1) >> (8 * (k & 3))) & 0xFF);
        st0 = k < r0 ? x0 : st0;
        st1 = k < r1 ? x1 : st1;
      }
    }
  }
  *out0 = st0;
  *out1 = st1;
}

// LDS_TABLE: the transition table lives in LDS (prescaled, see Tbl), else in global memory / L2.  LDS is sized per
// launch (dynamic): stage + exactly this automaton's table, class table and accept flags -- the kernel hides the
// latency of its per-step loads with resident waves only, so every workgroup that fits counts (5 instead of 6 per
// CU cost 35 %): small automata run 8 workgroups per CU.
struct RegexLds {
  uint32_t accept_off, total;
};
static inline RegexLds regex_lds_layout(uint32_t n_states, uint32_t n_classes, bool lds_table) {
  RegexLds l;
  const uint32_t table_bytes = lds_table ? (n_states * n_classes * 2 + 15) / 16 * 16 : 0;
  l.accept_off = kLdsTableOff + table_bytes;
  l.total = (l.accept_off + (lds_table ? n_states : 0) + 15) / 16 * 16 + 32;  // + the block reduction's 4 x 8 B
  return l;
}

// MULTI: the automaton is the PRODUCT of up to four patterns of the column (regex_compile.h, dfa_product): one walk
// per value decides all of them -- k format checks on a column cost one pass over its bytes, not k.
template <bool LDS_TABLE, bool DIRECT = false, bool VIEW = false, bool MULTI = false>
__global__ __launch_bounds__(256) void regex_match_kernel(RegexColDesc d, DfaView dfa, RegexLds lds,
                                                           unsigned long long *counters) {
  // no static LDS: the dynamic block then starts at LDS address 0 and the fixed offsets above are plain immediates
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  unsigned long long *const s_part = (unsigned long long *)(s_dyn + lds.total - 32);
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s_dyn != 0) __builtin_trap();  // see Tbl::step
  uint8_t *const s_stage0 = s_dyn + kLdsStageOff;
  uint16_t *const s_table = (uint16_t *)(s_dyn + kLdsTableOff);
  uint16_t *const s_class2 = (uint16_t *)(s_dyn + kLdsClassOff);
  uint8_t *const s_class = s_dyn + kLdsClassOff;     // global-table instance: plain classes
  uint8_t *const s_accept = s_dyn + lds.accept_off;  // accept-at-end flags: a row's verdict is one LDS read, not a
                                                     // dependent global load at the end of every wave step
  const uint32_t n_entries = dfa.n_states * dfa.n_classes;
  const uint32_t row_bytes = dfa.n_classes * 2;
  if (LDS_TABLE) {
    // prescaled copy (see Tbl): entry = byte offset of the next state's row; rows 0 (dead) and 1 (matched) absorb
    for (uint32_t i = threadIdx.x; i < n_entries; i += 256) {
      const uint32_t row = i / dfa.n_classes;
      const uint32_t n_final = MULTI ? dfa.n_final : 2u;
      s_table[i] = (uint16_t)((row < n_final ? row : (uint32_t)dfa.table[i]) * row_bytes);
    }
    s_class2[threadIdx.x] = (uint16_t)(dfa.byte_class[threadIdx.x] * 2u);
    for (uint32_t i = threadIdx.x; i < dfa.n_states; i += 256)
      s_accept[i] = MULTI ? dfa.accept_end[i] : (i == 1 ? 1 : dfa.accept_end[i]);
  } else {
    s_class[threadIdx.x] = dfa.byte_class[threadIdx.x];
  }
  __syncthreads();
  Tbl<LDS_TABLE, DIRECT> tbl;
  tbl.lds = s_dyn;
  tbl.g_table = (global_u16_ptr)(uintptr_t)dfa.table;
  tbl.ncls = dfa.n_classes;
  tbl.row_bytes = row_bytes;
  tbl.start = LDS_TABLE ? dfa.start * row_bytes : dfa.start;
  tbl.term = MULTI ? (LDS_TABLE ? (dfa.n_final - 1) * row_bytes : dfa.n_final - 1) : (LDS_TABLE ? row_bytes : 1);
  global_u8_ptr g_acc = (global_u8_ptr)(uintptr_t)dfa.accept_end;
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  const uintptr_t data0 = (uintptr_t)d.data;
  constexpr bool is_view = VIEW;  // Utf8View columns run their own instance: no data-dependent branches in the
                                  // offset path, so its loads are all issued before the first wait
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t *stage = s_stage0 + wave * (kStageBytes + 32);
  unsigned long long matches = 0;
  uint32_t multi[kMaxRegexGroup] = {0, 0, 0, 0};  // MULTI: matches per pattern (a lane sees < 2^32 rows)
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
    uint32_t vy, vz, vw;  // VIEW: words 1..3 of the view (inline bytes, or prefix / buffer / offset)
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
    r.vy = (uint32_t)vw.y;
    r.vz = (uint32_t)vw.z;
    r.vw = (uint32_t)vw.w;
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
  auto account = [&](const Row &r, int64_t i, uint32_t st_raw) {
    if (MULTI) {
      if (!r.in) return;
      // the state's accept mask decides every pattern of the group; a NULL row counts for the patterns that say so
      uint32_t mask = d.null_mask, nul = 2;
      if (r.valid) {
        const uint32_t st = tbl.state_of(st_raw);
        mask = LDS_TABLE ? (uint32_t)s_accept[st] : (uint32_t)g_acc[st];
        nul = 0;
      }
#pragma unroll
      for (int k = 0; k < kMaxRegexGroup; k++) {
        if (k < d.n_pat) {  // uniform
          const uint32_t hit = (mask >> k) & 1u;
          multi[k] += hit;
          if (d.hits_k[k]) d.hits_k[k][i] = (uint8_t)(nul ? 2u : hit);
        }
      }
      return;
    }
    if (r.valid) {
      const uint32_t st = tbl.state_of(st_raw);
      const bool hit = LDS_TABLE ? s_accept[st] != 0 : (st == 1 || g_acc[st]);
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
      // A view says where its value is: inline (<= 12 bytes, in the view itself) or at an offset of a data buffer.
      // Arrow's builders append the long values of consecutive rows one after the other, so the wave looks whether
      // the long values of its 128 rows lie in ONE buffer within a span that fits the stage (next to 16-byte slots for
      // the inline values, when there are any): then everything is walked from LDS like a plain column (1.27 -> 1.06 ms
      // per 100 M e-mail rows held as views, 2.37 -> 1.89 ms for three patterns); otherwise every lane walks its own
      // bytes from global memory.
      const uint32_t len0 = (uint32_t)(r0.e - r0.b), len1 = (uint32_t)(r1.e - r1.b);  // 0 for NULL rows
      const bool long0 = len0 > 12, long1 = len1 > 12;
      const bool inl0 = r0.valid && !long0, inl1 = r1.valid && !long1;
      const unsigned long long any_long = __builtin_amdgcn_ballot_w64(long0 || long1);
      const uint32_t area = __builtin_amdgcn_ballot_w64(inl0 || inl1) ? 2048u : 0u;  // the inline slots, 16 B a row
      bool staged = true;
      int64_t sbase = 0;
      uint32_t n16 = 0;
      uintptr_t buf = 0;
      if (any_long) {
        const int first_lane = __builtin_ctzll(any_long);
        const uint32_t bi = (uint32_t)__shfl(long0 ? r0.vz : r1.vz, first_lane, 64);
        const bool same = (!long0 || r0.vz == bi) && (!long1 || r1.vz == bi);
        const uint32_t lo0 = long0 ? r0.vw : 0xFFFFFFFFu, lo1 = long1 ? r1.vw : 0xFFFFFFFFu;
        const uint32_t hi0 = long0 ? r0.vw + len0 : 0u, hi1 = long1 ? r1.vw + len1 : 0u;  // (both < 2^31: no wrap)
        uint32_t lo = lo0 < lo1 ? lo0 : lo1, hi = hi0 > hi1 ? hi0 : hi1;
#pragma unroll
        for (int dlt = 32; dlt >= 1; dlt >>= 1) {
          const uint32_t ol = __shfl_xor(lo, dlt, 64), oh = __shfl_xor(hi, dlt, 64);
          lo = ol < lo ? ol : lo;
          hi = oh > hi ? oh : hi;
        }
        buf = (uintptr_t)d.buffers[bi];
        sbase = (int64_t)lo - (int64_t)((buf + lo) & 15);  // 16-byte blocks by ABSOLUTE address (see stage_in)
        staged = __builtin_amdgcn_ballot_w64(!same) == 0 && (int64_t)hi - sbase <= (int64_t)(kStageBytes - area);
        n16 = (uint32_t)(((int64_t)hi - sbase + 15) >> 4);
      }
      if (staged) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        typedef const u32x4 __attribute__((address_space(1))) *global_u4_ptr;
        if (inl0) *(u32x4 *)(stage + 16 * lane) = u32x4{r0.vy, r0.vz, r0.vw, 0u};
        if (inl1) *(u32x4 *)(stage + 1024 + 16 * lane) = u32x4{r1.vy, r1.vz, r1.vw, 0u};
        if (any_long) {
          global_u4_ptr src = (global_u4_ptr)(buf + (uintptr_t)sbase);
          for (uint32_t k = lane; k < n16; k += 64) *(u32x4 *)(stage + area + 16 * k) = src[k];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t o0 = long0 ? area + (uint32_t)((int64_t)r0.vw - sbase) : 16u * lane;
        uint32_t o1 = long1 ? area + (uint32_t)((int64_t)r1.vw - sbase) : 1024u + 16u * lane;
        uint32_t n0 = len0, n1 = len1;
        if (d.trim) {  // SQL TRIM(col) = btrim(col, ' '): U+0020 only
          while (n0 && stage[o0] == 0x20) o0++, n0--;
          while (n0 && stage[o0 + n0 - 1] == 0x20) n0--;
          while (n1 && stage[o1] == 0x20) o1++, n1--;
          while (n1 && stage[o1 + n1 - 1] == 0x20) n1--;
        }
        if constexpr (LDS_TABLE) {
          walk2_staged<DIRECT>(tbl, stage, o0, n0, o1, n1, &st0, &st1);
        } else {
          st0 = walk<LDS_TABLE, true, DIRECT>(tbl, 0, (int64_t)o0, (int64_t)(o0 + n0), stage, 0);
          st1 = walk<LDS_TABLE, true, DIRECT>(tbl, 0, (int64_t)o1, (int64_t)(o1 + n1), stage, 0);
        }
        stage_done();
      } else {
        bounds(r0, false, 0, &wb0, &we0);
        bounds(r1, false, 0, &wb1, &we1);
        walk2<LDS_TABLE, DIRECT>(tbl, r0.data, wb0, we0, r1.data, wb1, we1, &st0, &st1);
      }
    } else {
      // the step's values are contiguous: [b of its first row, e of its last)
      const int64_t b_first = __shfl(r0.b, 0, 64), e_half = __shfl(r0.e, 63, 64), e_last = __shfl(r1.e, 63, 64);
      const int64_t base = align16(b_first);
      if (e_last - base <= (int64_t)kStageBytes) {  // wave-uniform
        stage_in(base, e_last);
        bounds(r0, true, base, &wb0, &we0);
        bounds(r1, true, base, &wb1, &we1);
        if constexpr (LDS_TABLE) {
          walk2_staged<DIRECT>(tbl, stage, (uint32_t)(wb0 - base), (uint32_t)(we0 - wb0), (uint32_t)(wb1 - base),
                               (uint32_t)(we1 - wb1), &st0, &st1);
        } else {
          st0 = walk<LDS_TABLE, true, DIRECT>(tbl, data0, wb0, we0, stage, base);
          st1 = walk<LDS_TABLE, true, DIRECT>(tbl, data0, wb1, we1, stage, base);
        }
        stage_done();
      } else {
        const bool fit0 = e_half - base <= (int64_t)kStageBytes;
        if (fit0) stage_in(base, e_half);
        bounds(r0, fit0, base, &wb0, &we0);
        st0 = fit0 ? walk<LDS_TABLE, true, DIRECT>(tbl, data0, wb0, we0, stage, base)
                   : walk<LDS_TABLE, false, DIRECT>(tbl, data0, wb0, we0, nullptr, 0);
        if (fit0) stage_done();
        const int64_t base1 = align16(__shfl(r1.b, 0, 64));
        const bool fit1 = e_last - base1 <= (int64_t)kStageBytes;
        if (fit1) stage_in(base1, e_last);
        bounds(r1, fit1, base1, &wb1, &we1);
        st1 = fit1 ? walk<LDS_TABLE, true, DIRECT>(tbl, data0, wb1, we1, stage, base1)
                   : walk<LDS_TABLE, false, DIRECT>(tbl, data0, wb1, we1, nullptr, 0);
        if (fit1) stage_done();
      }
    }
    account(r0, i0, st0);
    account(r1, i1, st1);
  }
  if (MULTI) {
    for (int k = 0; k < kMaxRegexGroup; k++) {
      if (k >= d.n_pat) break;  // uniform
      unsigned long long m = multi[k];
#pragma unroll
      for (int dlt = 32; dlt >= 1; dlt >>= 1) m += __shfl_down(m, dlt, 64);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = m;
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned long long t = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (t) atomicAdd(d.counters_k[k], t);
      }
    }
    return;
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

// ---- a pattern that is an automaton AND a character count (regex_compile.h, Dfa::len_min / len_max: `^C{m,n}$`) ---------
// The walk has left one byte per row in d.hits (1 match, 0 no match, 2 NULL row); this pass takes the rows whose
// [TRIM]med value has fewer than min or more than max characters out of the matches and counts what remains (NULL
// rows by d.null_is_valid, as the walk counts them).  Most rows are decided by their byte length, like length_kernel's.
__global__ __launch_bounds__(256) void length_filter_kernel(RegexColDesc d, LengthBounds lb, unsigned long long *counters) {
  __shared__ unsigned long long s_part[4];
  const bool is_view = d.views != nullptr;
  unsigned long long matches = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.length; i += (int64_t)gridDim.x * 256) {
    const uint8_t h = d.hits[i];
    if (h != 1) {
      matches += (h == 2 && d.null_is_valid) ? 1 : 0;
      continue;
    }
    const int64_t slot = d.offset + i;
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
    if (d.trim) {  // SQL TRIM(col) = btrim(col, ' '): U+0020 only
      global_u8_ptr bytes = (global_u8_ptr)p;
      while (nbytes && bytes[0] == 0x20) bytes++, p++, nbytes--;
      while (nbytes && bytes[nbytes - 1] == 0x20) nbytes--;
    }
    bool ok;
    if (nbytes < lb.min_chars || (nbytes + 3) / 4 > lb.max_chars) {
      ok = false;  // chars <= bytes < min, or chars >= ceil(bytes / 4) > max
    } else if (nbytes <= lb.max_chars && (nbytes + 3) / 4 >= lb.min_chars) {
      ok = true;
    } else {
      uint64_t cont = 0, remaining = nbytes;
      uintptr_t q = p;
      while (remaining > 0) {
        const uint32_t nb = remaining < 8 ? (uint32_t)remaining : 8u;
        const uint32_t skip = (uint32_t)(q & 7);
        const uintptr_t base = q & ~(uintptr_t)7;
        uint64_t w = *(global_u64_ptr)base >> (8 * skip);
        if (skip + nb > 8) w |= *(global_u64_ptr)(base + 8) << (8 * (8 - skip));
        if (nb < 8) w &= (1ull << (8 * nb)) - 1;
        cont += __builtin_popcountll(w & ~(w << 1) & 0x8080808080808080ULL);
        q += nb;
        remaining -= nb;
      }
      const uint64_t chars = nbytes - cont;
      ok = chars >= lb.min_chars && chars <= lb.max_chars;
    }
    if (!ok) d.hits[i] = 0;
    matches += ok ? 1 : 0;
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

void launch_length_filter(const RegexColDesc &d, const LengthBounds &lb, unsigned long long *d_counters, int n_cu,
                          hipStream_t stream) {
  int64_t blocks = (d.length + 255) / 256;
  if (blocks > (int64_t)n_cu * 8) blocks = (int64_t)n_cu * 8;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(length_filter_kernel, dim3((int)blocks), dim3(256), 0, stream, d, lb, d_counters);
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
  const uint64_t entries = (uint64_t)dfa.n_states * dfa.n_classes;
  const bool direct = dfa.n_classes == 256 && dfa.direct && entries <= 4096;
  // An LDS-resident table: whatever stays below 64 KiB (an entry is the byte offset of a row in 16 bits).  Up to round 5
  // the limit was kRegexLdsEntries (32 KiB: five to seven workgroups a CU) and bigger automata were walked from L2 by
  // seven; round 6 measured the 316-state automaton of `^[\w.@+-]*$` (61 KiB: TWO workgroups a CU) at 1.90 ms per
  // 100 M x 28 B from LDS against 2.96 ms from L2 -- the dependent table read is what a walk waits for.
  // TGX_REGEX_LDS_ENTRIES overrides (16384: the old rule).
  static const uint64_t lds_entries = [] {
    const char *e = getenv("TGX_REGEX_LDS_ENTRIES");
    return e ? (uint64_t)strtoull(e, nullptr, 10) : (uint64_t)32767;
  }();
  const bool in_lds = direct || (entries <= lds_entries && entries * 2 <= 65535);
  const RegexLds lds = regex_lds_layout(dfa.n_states, dfa.n_classes, in_lds);
  const bool view = d.views != nullptr;
  // persistent grid: exactly the workgroups that stay resident (LDS, and 7 rather than 8 waves per SIMD at this
  // kernel's 88 SGPRs / 66 VGPRs) -- one workgroup more than fit runs as a second round: 1.09 ms instead of 0.91 ms
  auto resident = [&](auto kernel, int id) -> int64_t {
    static std::mutex mu;
    static std::map<std::pair<int, uint32_t>, int> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({id, lds.total});
    if (it != cache.end()) return it->second;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, lds.total) != hipSuccess || occ < 1) occ = 4;
    // the occupancy API answers one workgroup too many for 256-thread kernels with 81..96 SGPRs (these instances
    // have 88..91; 800 SGPRs per SIMD / (96 + 16) = 7 waves): MI355X_MICROARCH.md, "Correctness boundaries"
    if (occ > 7) occ = 7;
    // ... and one too many when the workgroups' LDS only just fits: six workgroups of 27 136 bytes (three patterns in
    // one walk) are 162 816 of the CU's 163 840 bytes, but only FIVE were resident and the sixth ran as a second round
    // -- 1.40 ms per 100 M rows instead of 0.97 (sweep of 3 .. 7 workgroups per CU, round 5; the same second-round cliff
    // the eighth workgroup showed).  LDS is handed out in granules: 512 bytes did not explain the measurement, 1280
    // (1/128 of the CU's LDS) does; counting one workgroup too few costs ~5 %, one too many 45 %.
    constexpr uint32_t kLdsGranule = 1280;
    const int by_lds = (int)((160u << 10) / ((lds.total + kLdsGranule - 1) / kLdsGranule * kLdsGranule));
    if (by_lds >= 1 && occ > by_lds) occ = by_lds;
    if (const char *e = getenv("TGX_REGEX_RESIDENT")) occ = std::max(1, atoi(e));  // (experiments: workgroups per CU)
    cache[{id, lds.total}] = occ;
    return occ;
  };
  int64_t max_blocks = (d.length + 511) / 512;  // 128 rows per wave step, four waves
  if (max_blocks < 1) max_blocks = 1;
  const dim3 block(256);
#define TGX_RX(LDS, DIRECT, VIEW, ID)                                                                            \
  do {                                                                                                           \
    auto k = regex_match_kernel<LDS, DIRECT, VIEW>;                                                              \
    const int64_t blocks = std::min<int64_t>(max_blocks, (int64_t)n_cu * resident(k, ID));                       \
    if (lds.total > (64u << 10))                                                                                 \
      (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds.total);    \
    hipLaunchKernelGGL(k, dim3((int)blocks), block, lds.total, stream, d, dfa, lds, d_counters);                 \
  } while (0)
#define TGX_RXM(LDS, DIRECT, VIEW, ID)                                                                           \
  do {                                                                                                           \
    auto k = regex_match_kernel<LDS, DIRECT, VIEW, true>;                                                        \
    const int64_t blocks = std::min<int64_t>(max_blocks, (int64_t)n_cu * resident(k, ID));                       \
    hipLaunchKernelGGL(k, dim3((int)blocks), block, lds.total, stream, d, dfa, lds, d_counters);                 \
  } while (0)
  if (d.n_pat > 1) {  // a product automaton: several patterns in one walk
    if (direct) {
      if (view) TGX_RXM(true, true, true, 6); else TGX_RXM(true, true, false, 7);
    } else if (in_lds) {
      if (view) TGX_RXM(true, false, true, 8); else TGX_RXM(true, false, false, 9);
    } else {
      if (view) TGX_RXM(false, false, true, 10); else TGX_RXM(false, false, false, 11);
    }
    return;
  }
  if (direct) {
    if (view) TGX_RX(true, true, true, 0); else TGX_RX(true, true, false, 1);
  } else if (in_lds) {
    if (view) TGX_RX(true, false, true, 2); else TGX_RX(true, false, false, 3);
  } else {
    if (view) TGX_RX(false, false, true, 4); else TGX_RX(false, false, false, 5);
  }
#undef TGX_RX
#undef TGX_RXM
}

}  // namespace tgx
