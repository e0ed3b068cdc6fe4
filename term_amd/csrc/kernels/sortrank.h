// sortrank.h -- the sample sort behind SQL RANK() (kernels/sortrank.hip): descriptors shared with the host code.
//
// RANK() OVER (ORDER BY c) of the reference's Spearman query (TG/analyzers/advanced/correlation.rs:334-350) is
// "1 + the number of keys that sort before mine".  The keys (IEEE totalOrder keys of CAST(c AS DOUBLE), unsigned) are
// partitioned by SPLITTERS drawn from a sorted random sample -- whatever the distribution, every bucket between two
// neighbouring splitters holds about the same number of keys -- in up to three passes of at most 256 ways each, until
// a bucket fits a small workgroup's LDS, where the rank inside the bucket is counted; a key EQUAL to a splitter goes
// to a bucket of its own ("equality bucket": heavy ties never need sorting, every key there has the bucket's first
// position as its rank).  No comparison sort, no radix passes over all 64 bits.
//
// A pass needs to know where every bucket starts before it moves a key.  Counting the keys first (a read of all keys
// per pass) is one way; the sample is the other: the share of the sample that falls into a bucket, plus 6.5 standard
// deviations, is room enough but once in ~10^6 jobs, and a pass that finds a bucket full says so in the job's status
// word -- the caller then runs the job again with counted buckets (SrJob::optimistic).  Only the last pass counts: its
// buckets are the work items of the ranking kernels and lie one behind the other.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tgx {

constexpr int kSrTile = 2048;        // keys per tile of a partition pass
constexpr int kSrPartThreads = 512;  // partition kernels: 8 waves, 4 keys per thread and tile
constexpr int kSrPartPer = kSrTile / kSrPartThreads;
constexpr int kSrMaxSplit = 255;     // splitters per part (an 8-step branch-free search in LDS)
constexpr int kSrMaxNb = 2 * kSrMaxSplit + 1;
constexpr int kSrXcds = 8;
constexpr int kSrFirstParts = 8;     // the first pass cuts its input into this many stretches, one per XCD
// last pass, small buckets: 256 threads, 8 keys each
constexpr int kSrFastThreads = 256;
constexpr int kSrFastPer = 8;
constexpr int kSrFastCap = kSrFastThreads * kSrFastPer;
constexpr int kSrFastBinBits = 11;
// last pass, buckets that outgrew the small one (and chunk against chunk beyond its own capacity)
constexpr int kSrSlowThreads = 512;
constexpr int kSrSlowPer = 8;
constexpr int kSrSlowCap = kSrSlowThreads * kSrSlowPer;
constexpr int kSrSlowBinBits = 12;
constexpr uint32_t kSrEqPiece = 8192;  // an equality bucket is handed out in pieces of this many keys
constexpr uint32_t kSrEqTiny = 32;     // ... unless it holds at most this many: then one thread walks it

// what a job's keys (and 8-byte payloads) are when pass 0 reads them: sort keys already, or column values that become
// sort keys as they are read -- CAST(c AS DOUBLE), then the IEEE total order as an unsigned number (the Spearman state
// ranks a lent batch straight from the caller's columns: no converted copy is written first)
enum SrSource : int { kSrKeys = 0, kSrInt64 = 1, kSrFloat64 = 2 };
__host__ __device__ inline uint64_t sr_source_key(uint64_t bits, int source) {
  if (source == kSrKeys) return bits;
  double d;
  if (source == kSrFloat64) {
    __builtin_memcpy(&d, &bits, 8);
  } else {
    d = (double)(int64_t)bits;
  }
  int64_t k;
  __builtin_memcpy(&k, &d, 8);
  k ^= (int64_t)(((uint64_t)(k >> 63)) >> 1);  // (f64_total_key, device_types.h)
  return (uint64_t)k ^ 0x8000000000000000ULL;  // signed total order -> unsigned order
}

struct SrTileRef {  // one tile of a partition pass: keys [begin, begin + len) of a piece of `part`, len <= kSrTile
  uint32_t part, begin, len;
};
struct SrItem {  // one unit of the last pass: a bucket (or a piece of an equality bucket)
  uint32_t start, count;
  uint32_t rank_base;  // keys that sort before the bucket
  uint32_t flags;      // bit 0: every key of the item is the same
  uint64_t lo, hi;     // the splitters the bucket lies between: lo < key < hi (small items only)
};

// what the last pass does with a key's rank
enum SrSink : int {
  kSrSorted = 0,       // keys (and payloads) written in order
  kSrRank32 = 1,       // rank32[slot of the key in the partitioned arrays] = RANK() - 1
  kSrSums = 2,         // payload = RANK(x) - 1 of the pair: the five rank sums
  kSrRankScatter = 3,  // rank_out[payload] = ext_base + RANK()   (payload = where the key came from)
};

struct RankSums {
  unsigned long long wrapped[5];   // UInt64 arithmetic of the reference: sums and products wrap modulo 2^64
  unsigned long long exact_lo[5];  // the same sums without wrapping, 128 bits
  unsigned long long exact_hi[5];
};

// One partition pass (tiles -> count or estimate -> offsets -> scatter), by value in the kernel-argument segment.
// A pass cuts PARTS (pass 0: kSrFirstParts stretches of the input, all with the same splitters; later passes: the
// buckets of the pass before, each with the splitters that lie between its bounds) into nb = 2 S + 1 buckets.  A part
// lies in PIECES: piece s of part p is [pbeg[s * pstride + p], pend[s * pstride + p]) -- one piece, but for the parts
// of pass 1, which are what the stretches of pass 0 each wrote of a bucket.
struct SrLevel {
  const uint64_t *keys_in;
  const void *pay_in;  // nullptr with a 4-byte payload: the payload is the key's index (iota)
  uint64_t *keys_out;
  void *pay_out;
  const uint64_t *fine;  // all splitters, in order
  int32_t key_source, pay_source;  // SrSource of keys_in / an 8-byte pay_in (pass 0 of a job over column values)
  int32_t level;         // 0, 1, 2
  uint32_t nbp;          // level 2: buckets per part of level 1 (a part is (d1, d2) = (p / nbp, p % nbp))
  uint32_t w1, w2;       // first splitter of a part's range: (d1 >> 1) * w1 + (d2 >> 1) * w2
  uint32_t stride;       // splitter k of a part: fine[first + (k + 1) * stride - 1]
  uint32_t split_count;  // S
  uint32_t nb;           // 2 S + 1
  const uint32_t *pbeg, *pend;  // the pieces of the parts
  uint32_t npieces, pstride;
  uint32_t nparts;
  uint32_t *part_size;   // [nparts]: keys of the part (the tiles kernel adds the pieces up)
  uint32_t *part_total;  // [nparts]: room the part's buckets take in the output (counted: the same)
  uint32_t *out_base;    // [nparts + 1]: where the part's first bucket starts in the output
  uint32_t *limit;       // [nparts][nb]: where the bucket's room ends
  uint32_t *pbegin;      // pass 0: [nparts][nb]: where each stretch's range of a bucket starts (the next pass's pieces)
  uint32_t out_cap;      // keys the output arrays hold
  uint32_t *status;      // the job's status word: != 0 -- every kernel returns.  Bit 4 * pass: the buckets' room
                         // together exceeds the output arrays; bit 4 * pass + 1: a bucket was full
  SrTileRef *tiles;      // [kSrXcds][tile_cap]: the tiles each XCD's workgroups take
  uint32_t tile_cap;
  uint32_t *tile_count;  // [kSrXcds]
  uint32_t *tot;         // [nparts][nb]: keys per bucket
  uint32_t *cursor;      // [nparts][nb]: where the bucket's next run goes
  uint32_t group;        // workgroups of an XCD that walk a stretch of its tile list side by side (0: all of them)
  uint32_t *bstart;      // pass 0: [nb + 1], later: [nparts * nb + 1]: where the buckets start
};

struct SrFinal {
  const uint64_t *keys;
  const void *pay;
  const SrItem *items;
  const uint32_t *n_items;
  const SrItem *tiny;  // equality buckets of a few keys (every splitter is a key of the input: at least its own bucket
  const uint32_t *n_tiny;  // holds one), a thread each
  uint64_t *out_keys;  // kSrSorted
  void *out_pay;
  uint32_t *rank32;    // kSrRank32
  uint64_t *rank_out;  // kSrRankScatter
  uint64_t ext_base;
  RankSums *partials;  // kSrSums: one per workgroup
  uint32_t cap;        // keys ranked in one go by the chunked kernel (<= kSrSlowCap)
  const uint32_t *status;
};

// knobs (environment, read per call): tests shrink them so that small inputs take every path
struct SrTuning {
  uint32_t target;      // keys per bucket aimed at            TGX_SORT_TARGET   (1024)
  uint32_t oversample;  // sample keys per bucket               TGX_SORT_SAMPLE   (16)
  uint32_t cap;         // largest bucket of the small kernel   TGX_SORT_CAP      (2048)
  uint32_t slow_cap;    // SrFinal::cap                         TGX_SORT_SLOWCAP  (4096)
  uint32_t max_split;   // splitters per pass                   TGX_SORT_SPLIT    (255)
  uint32_t wg_per_cu;   // workgroups per CU of a pass          TGX_SORT_WG       (3)
  uint32_t first_parts; // stretches of pass 0                  TGX_SORT_PARTS    (8)
  uint32_t optimistic;  // 0: every pass counts                 TGX_SORT_OPTIMISTIC (1)
  uint32_t sigmas_x2;   // twice the standard deviations of room TGX_SORT_SIGMAS_X2 (13; tests shrink it to see a job fail)
  uint32_t optimistic_min;  // callers ask for it from this many keys on  TGX_SORT_OPTIMISTIC_MIN (1 Mi)
};
SrTuning sr_tuning();

// A sort / ranking job.  Everything is queued on `stream`; nothing is waited for.
struct SrJob {
  const uint64_t *keys = nullptr;  // n keys (left as they are unless k[1] aliases them)
  const void *pay = nullptr;       // payloads of pay_bytes each, or nullptr (iota with pay_bytes == 4)
  uint64_t n = 0;
  int pay_bytes = 0;               // 0, 4 or 8
  int key_source = kSrKeys, pay_source = kSrKeys;  // SrSource (pay_source: 8-byte payloads only)
  // ping-pong space of the partition passes: pass i writes k[i & 1] / p[i & 1].  k[1] / p[1] may be the input arrays
  // (they are read by pass 0 only): the keys then come back permuted, each with its payload
  uint64_t *k[2] = {nullptr, nullptr};
  void *p[2] = {nullptr, nullptr};
  int sink = kSrSorted;
  uint64_t *out_keys = nullptr;
  void *out_pay = nullptr;
  uint32_t *rank32 = nullptr;
  uint64_t *rank_out = nullptr;
  uint64_t ext_base = 0;
  RankSums *partials = nullptr;  // kSrSums: sr_partials_count() entries
  // bucket sizes taken from the sample in all passes but the last: k[i] / p[i] then hold cap[i] >= sr_roomy_elems(n)
  // keys (the buckets lie apart), the input arrays are not among them, and *status (a device word, written by the job)
  // != 0 once the job is through means: nothing of its output is valid, run it again with optimistic = false
  bool optimistic = false;
  uint64_t cap[2] = {0, 0};
  uint32_t *status = nullptr;
};
struct SrPlaced {  // where the job's keys / payloads lay when the last pass read them (kSrRank32 is aligned to it)
  const uint64_t *keys;
  const void *pay;
};
size_t sr_workspace_bytes(uint64_t n);
uint64_t sr_roomy_elems(uint64_t n);  // elements of k[i] / p[i] an optimistic job wants
bool sr_optimistic_applies(uint64_t n);  // would a job of n keys with optimistic = true and that room take it?
int sr_partials_count();
hipError_t sr_run(const SrJob &job, void *workspace, size_t workspace_bytes, hipStream_t stream, SrPlaced *placed);

}  // namespace tgx
