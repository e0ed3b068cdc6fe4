// Views of the device-resident key sets behind TGX_CHECK_DISTINCT (see distinct.hip).
#pragma once
#include <stdint.h>

namespace tgx {

constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFULL;

struct DistinctColDesc {
  const void *values;       // int64 / float64 bit patterns
  const uint8_t *validity;  // or nullptr
  int64_t offset;
  int64_t length;
  int32_t want_multiplicity;
  int32_t pad;
};

struct HashSetView {
  uint64_t *keys;  // capacity = mask + 1 slots, kEmptyKey = free; 128-bit sets use two words per slot
  uint32_t *dup;   // 1 bit per slot: key seen at least twice (only with multiplicity)
  uint64_t mask;
  // EXACT key sets (TGX_FLAG_EXACT_KEYS; distinct128.hip "exact"): a slot is (first fingerprint word, reference into
  // the key store) and the store holds, per key, a 16-byte header (second fingerprint word, length, kind) followed by
  // the key's bytes -- equal fingerprints are confirmed byte by byte.  nullptr: a fingerprint set (two words = the
  // fingerprint).
  uint64_t *store;                     // 8-byte words
  unsigned long long *store_cursor;    // next free word (device)
  uint64_t store_words;                // capacity; an entry that would not fit raises counters[kCntStoreFull]
  // the batch's new keys between its two phases (distinct128.hip): (slot, second fingerprint word) pairs, every WAVE of
  // the insert kernel filling a region of its own (a fill shared by all waves is one address: ~50 ns per bump, 80 ms per
  // 100 M new keys -- measured twice, for the list and for the store cursor)
  uint64_t *pending;
  uint32_t *pending_counts;            // [waves]: pairs in each wave's region
  uint64_t pending_region;             // pairs a region holds
  uint32_t pending_waves, pad_;
};
// the launch shape of the insert kernels of an exact set (and of the commit kernels that follow them wave for wave)
inline uint32_t exact_blocks(uint64_t items) {
  uint64_t blocks = (items + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  return (uint32_t)blocks;
}
inline uint32_t exact_waves(uint64_t items) { return exact_blocks(items) * 4; }
inline uint64_t exact_region(uint64_t items) {  // rows a wave of the grid-stride loop sees at most
  const uint64_t threads = (uint64_t)exact_blocks(items) * 256;
  return (items + threads - 1) / threads * 64;
}

// The 128-bit key of the fingerprint function (Chaskey's K, K1 = 2K, K2 = 4K in GF(2^128); distinct128.hip).  Drawn from
// the OS at tgx_plan_create (or set by the caller: every rank / every state that exchanges keys must hold the same one).
struct FpKey {
  uint32_t k[4], k1[4], k2[4];
};
inline void fp_key_times_two(const uint32_t in[4], uint32_t out[4]) {
  const uint32_t carry = (in[3] >> 31) ? 0x87u : 0u;
  out[3] = (in[3] << 1) | (in[2] >> 31);
  out[2] = (in[2] << 1) | (in[1] >> 31);
  out[1] = (in[1] << 1) | (in[0] >> 31);
  out[0] = (in[0] << 1) ^ carry;
}
inline FpKey fp_key_expand(const uint32_t k[4]) {
  FpKey key;
  for (int i = 0; i < 4; i++) key.k[i] = k[i];
  fp_key_times_two(key.k, key.k1);
  fp_key_times_two(key.k1, key.k2);
  return key;
}

struct BitmapView {
  uint32_t *seen;   // bit (key - base)
  uint32_t *twice;  // same indexing; only with multiplicity
  int64_t base;
  uint64_t range;   // number of representable keys
};

// Range-partitioned population of the bitmap (distinct.hip, partition_kernel / bucket_apply_kernel):
// phase 1 scatters (key - base) into P = ceil(range / 2^sub_bits) bucket lists of 32-bit in-bucket
// offsets; phase 2 replays each list against its 2^sub_bits-bit slice of the bitmap held in LDS.
constexpr uint32_t kMaxPartitions = 2048;
constexpr uint32_t kListPad = 0xFFFFFFFFu;  // filler of the padded runs in the bucket lists
// a (tile, bucket) run is padded to a whole number of these many slots: 4-byte entries / 2-byte entries
#ifndef TGX_RUN_PAD4
#define TGX_RUN_PAD4 16
#endif
#ifndef TGX_RUN_PAD2
#define TGX_RUN_PAD2 32
#endif
constexpr int kRunPad4 = TGX_RUN_PAD4, kRunPad2 = TGX_RUN_PAD2;
constexpr int kPartitionThreads = 1024;
constexpr int kPartitionKeysPerThread = 32;
constexpr int kPartitionTile = kPartitionThreads * kPartitionKeysPerThread;  // 32768 keys
constexpr int kSliceWordsLds = 32768;  // 128 KiB of LDS bitmap per workgroup

struct PartitionParams {
  const void *values;
  const uint8_t *validity;
  int64_t offset;
  int64_t length;
  int64_t base;
  uint64_t range;
  uint32_t sub_bits;   // log2(keys per bucket)
  uint32_t n_buckets;  // P
  uint64_t cap;        // list capacity per bucket in slots (4-byte slots, multiple of 16; key16: 2-byte slots, of 32)
  uint32_t *lists;     // n_lists x cap slots: the lists of buckets [bucket0, bucket0 + n_lists)
  // [0,P): next free slot per list (zeroed per batch); [P,2P): valid-length limits (all-ones); [2P]: 1 when keys next
  // to each other in the column share buckets (partition_init_kernel's probe): partition_kernel's CLUSTERED passes
  unsigned long long *cursors;
  uint32_t *seen;      // global bitmap (rounded up to whole slices)
  uint32_t *twice;     // or nullptr
  int32_t want_multiplicity;
  int32_t key16;       // 1: sub_bits <= 16 and list entries are 2 bytes; 2: 20-bit entries, three to an 8-byte word, cap
                       //    and every run a multiple of 24 (64 bytes) -- both never with multiplicity; 0: 4-byte entries
  // not nullptr: the pass also produces the column's COUNT / MIN / MAX / SUM (one ScanPartial per workgroup, folded
  // by scan_reduce_kernel) -- the numeric scan then skips the column: a unique-key column crosses HBM once for its
  // range checks and its uniqueness check together
  struct ScanPartial *stats;
  struct OutlierStats *outliers;  // with stats: the aggregates of the (rare) keys outside [base, base + range)
  // the buckets this batch is expected to touch (all of them unless the batch's own value range is known -- a flush of
  // a stream whose keys grow covers a few slices of a bitmap that has grown with the stream): lists are laid out and
  // sized for these only; a key that lands in another bucket after all takes the spill path (exact, slow)
  uint32_t bucket0, n_lists;
  int32_t probe;  // 0: partition_init_kernel leaves the clustered flag at 0 (TGX_NO_CLUSTERED_PROBE=1: for A/B runs)
  // 0: the probe's flag picks the form and BOTH forms of partition_kernel are launched (the other one leaves at once:
  // ~5 us of a 1.6 ms step); 1 / 2: the host remembers which form the column's last batch took (counters[kCntForm]) and
  // launches only that one -- plain / CLUSTERED -- whatever the probe says: both forms are exact for any keys, a wrong
  // guess costs time once and is corrected by the next look at the counters
  int32_t force_form;
};

// keys outside the bitmap's range enter the column's aggregates through global atomics (they are rare: the range
// was sampled from the column): SUM as the two halves sum(key & 0xFFFFFFFF) + 2^32 * sum(key >> 32)
struct OutlierStats {
  long long mn, mx;
  unsigned long long lo32_sum;
  long long hi32_sum;
  unsigned long long count;
};

// a strided sample of an Int64 column: the value range a DISTINCT pass can expect BEFORE the column has been scanned
struct DistinctSample {
  int64_t min_v, max_v;
  unsigned long long count;  // valid values sampled
  unsigned long long pad;
};

// 16-byte record used by merge / serialize / the cross-rank key exchange.
struct KeyRecord {
  uint64_t key;
  uint64_t count;  // saturates at 2
};

// 32-byte record of the 128-bit fingerprint sets (Utf8 columns, distinct128.hip)
struct KeyRecord128 {
  uint64_t a, b;   // the two 64-bit hashes
  uint64_t count;  // saturates at 2
  uint64_t pad;
};

// Big Utf8 batches (distinct128.hip, fp_* kernels): the fingerprints are partitioned twice, by 8 bits of their first
// word each time, into kFpFan^2 lists short enough to be deduplicated in LDS -- no global atomic per value.
constexpr int kFpFan = 256;              // lists per level
constexpr int kFpXcds = 8;               // level 1 keeps a set of lists per XCD (workgroup w runs on XCD w % 8): runs that
                                         // neighbour each other in a list then come out of ONE L2, where partial lines meet
constexpr int kFpTile = 2048;            // records a workgroup groups in LDS at a time
constexpr uint32_t kFpSlots = 4096;      // LDS table of one final list (16 KiB: eight workgroups a CU); lists of bigger
                                         // batches get 16384 or 32768 slots (fp_count_kernel<SLOTS>), load <= 3/4
constexpr uint32_t kFpListMax = 24576;   // records a final list may hold at most (3/4 of 32768; a 16-bit record index)
constexpr int64_t kFpMinRows = 1 << 21;  // smaller batches go straight into the global table (same time at 1-2 M rows)
struct FpLists {
  uint64_t *recs;     // [lists][cap] records of two words
  uint32_t *offered;  // [lists] records offered to the list; those beyond cap were dropped (kCntOutOfRange counts
                      // the workgroups that dropped some: the batch is then redone through the global table)
  uint64_t cap;
};

// counters[] slots shared by the kernels of distinct.hip
enum {
  kCntDistinct = 0,   // keys in the set (excluding the EMPTY stand-in)
  kCntTwice = 1,      // keys seen at least twice
  kCntEmptyRows = 2,  // rows (or record counts) carrying the all-ones key
  kCntValidRows = 3,  // non-null rows scanned
  kCntOutOfRange = 4, // bitmap mode: keys outside [base, base+range); fingerprint lists: overflows -- must stay 0
  kCntSpare = 5,
  kCntStoreFull = 6,  // exact key sets: entries that did not fit the key store -- must stay 0
  kCntForm = 7,       // the partition pass's last probe: 1 plain, 2 keys in order (PartitionParams::force_form)
  kNumDistinctCounters = 8
};

// one component of a tuple key (distinct_tuple_kernel)
struct TupleCol {
  int32_t kind;  // 0 Int64 / Float64 (bit pattern), 1 Utf8, 2 LargeUtf8, 3 Utf8View, 4 Dictionary<Int32, Utf8 | LargeUtf8>
  int32_t dict_large;             // kind 4: the dictionary's offsets are 64-bit
  const void *values;             // numeric values / Utf8View views / dictionary indices
  const void *offsets;            // Utf8 / LargeUtf8 (kind 4: the dictionary's)
  const uint8_t *data;            // (kind 4: the dictionary's)
  const uint8_t *const *buffers;  // Utf8View
  const uint8_t *validity;
  int64_t offset;
  const uint8_t *dict_validity;   // kind 4: a NULL entry makes the component NULL
  int64_t dict_offset;            // kind 4: the dictionary's own Arrow offset
};
constexpr int kMaxTupleCols = 8;
struct TupleDesc {
  TupleCol cols[kMaxTupleCols];
  int32_t n_cols;
  int32_t want_multiplicity;
  int64_t length;
  FpKey key;
};

}  // namespace tgx
