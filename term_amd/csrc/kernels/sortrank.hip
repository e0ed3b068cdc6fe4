// sortrank.hip -- SQL RANK() over 64-bit keys on gfx950: a sample sort written for the chip, no library primitive.
//
// The reference ranks in SQL (TG/analyzers/advanced/correlation.rs:334-350: RANK() OVER (ORDER BY CAST(c AS DOUBLE)),
// twice); DataFusion sorts the partition and numbers the tie runs.  Here (sortrank.h has the outline):
//   sample     16 keys per aimed-at bucket, one from every stretch of n / ns keys at a hashed position; the sample is
//              sorted by this very machinery (a 1 G-key job sorts 16 M, that one 256 K, that one 4 K, that one in a
//              single workgroup) and every 16th sample key is a splitter
//   partition  up to three passes of at most 256 ways (2 S + 1 buckets for S splitters: the odd ones hold the keys
//              EQUAL to a splitter).  A pass is tiles -> count -> offsets -> scatter: buckets come out exactly sized
//              and contiguous.  The scatter pass is the repo's list pass (kernels/lists.h) with exact reservations: a
//              tile of 2048 keys is grouped by bucket in LDS, every (tile, bucket) run reserves its place in the bucket
//              with ONE atomic and leaves in consecutive lanes.  A store of 8 scattered bytes per lane costs a memory
//              transaction per lane AND leaves 56 bytes of every line for somebody else to fill -- from another
//              XCD's L2 that means a partial write per key (the first version did exactly that: 1.2 TB/s) -- so all
//              tiles of one part are taken by workgroups of ONE XCD: the runs that complete each other's lines meet in
//              that L2.  (Pass 0 has one part, the input; it is cut into stretches, each with its own
//              range in every bucket: one stretch per XCD.)
//              Where a bucket starts has to be known before a key moves.  The last pass counts (a read of all keys:
//              its buckets are the ranking kernels' work items and lie one behind the other); the passes before it
//              take a bucket's size from its share of the SAMPLE plus 6.5 standard deviations (sr_room*): buckets
//              then lie apart, a part is read in pieces, and a run that finds its bucket full raises the job's status
//              word -- every later kernel returns at once and the caller runs the job again, counted (SrJob).
//              These kernels are bound by the instructions they issue (~250 a key in a pass, ~110 in the count), not
//              by bytes: the splitter search steps by selects, the look for lanes that agree on a bucket is taken on
//              one tile in sixteen while it does not pay, the pass over column values (SrSource) is its own instance.
//   rank       one 256-thread workgroup per bucket of ~1024 keys: the keys are binned in LDS by the leading bits of
//              (key - lower splitter) -- a counting sort with as many bins as the bucket may hold keys; two keys meet
//              in a bin only if they are equal or agree in those bits, and those few are compared -- so a key's rank
//              is (keys before the bucket) + (keys in lower bins) + (smaller keys in its bin).  The next bucket's
//              keys are requested before the current one's are looked at.  A bucket that outgrew that kernel (more
//              than twice the aim: one in ~10^3 at 16 samples a bucket) goes to a larger, chunk-against-chunk kernel;
//              equality buckets are not looked at.
// What the last pass does with a rank is the job's sink (SrSink): the Spearman state ranks x with y as the payload,
// then y with RANK(x) as the payload, and sums -- no rank ever finds its way back to a row (spearman_device.cpp).
#include "sortrank.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <cmath>

namespace tgx {
int tgx_num_cus();

namespace {

__device__ __forceinline__ uint32_t sr_lane() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ uint32_t sr_below(uint64_t m) {  // set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// cursor[b] += 1 for every lane with `valid`, returning the value before the lane's own increment.  Lanes that agree
// on b are served by one LDS add of their count: the leader's group first, and on while the groups are worth it (keys
// in order, few distinct values: an LDS atomic on ONE address retires every ~10 cycles, 64 lanes on it cost 600);
// what is left (scattered keys: the first group is a lane or two) goes lane by lane.  Called by ALL lanes of the wave.
// `groups` (wave-uniform): look for lanes that agree at all; *paid: the first group was worth its add.  A partition
// kernel looks on one tile in sixteen and goes on looking while it pays (keys in order, heavy ties); on scattered keys
// the look costs a fifth of what the tile's keys cost otherwise.
__device__ __forceinline__ uint32_t sr_claim(uint32_t *cursor, uint32_t b, bool valid, bool groups = true,
                                             bool *paid = nullptr) {
  const uint32_t lane = sr_lane();
  uint64_t todo = __builtin_amdgcn_ballot_w64(valid);
  uint32_t pos = 0;
  while (groups && todo) {  // (wave-uniform)
    const int leader = (int)__builtin_ctzll(todo);
    const uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)b, leader);
    const uint64_t grp = __builtin_amdgcn_ballot_w64(valid && b == lb) & todo;
    const uint32_t cnt = (uint32_t)__builtin_popcountll(grp);
    uint32_t base = 0;
    if (lane == (uint32_t)leader) base = atomicAdd(&cursor[lb], cnt);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    if ((grp >> lane) & 1ull) pos = base + sr_below(grp);
    todo &= ~grp;
    if (cnt < 8) break;
    if (paid) *paid = true;
  }
  if ((todo >> lane) & 1ull) pos = atomicAdd(&cursor[b], 1u);
  return pos;
}

// the same without the positions (the counting pass): the lanes left over after the groups send adds that return nothing
__device__ __forceinline__ void sr_tally(uint32_t *hist, uint32_t b, bool valid, bool groups, bool *paid) {
  const uint32_t lane = sr_lane();
  uint64_t todo = __builtin_amdgcn_ballot_w64(valid);
  while (groups && todo) {  // (wave-uniform)
    const int leader = (int)__builtin_ctzll(todo);
    const uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)b, leader);
    const uint64_t grp = __builtin_amdgcn_ballot_w64(valid && b == lb) & todo;
    const uint32_t cnt = (uint32_t)__builtin_popcountll(grp);
    if (cnt < 8) break;
    *paid = true;
    if (lane == (uint32_t)leader) __hip_atomic_fetch_add(&hist[lb], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    todo &= ~grp;
  }
  if ((todo >> lane) & 1ull) __hip_atomic_fetch_add(&hist[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- parts and their splitters ------------------------------------------------------------------------------------
struct SrPartInfo {
  bool eq;         // an equality bucket of an earlier pass: one value, nothing to split
  uint32_t owner;  // XCD whose workgroups take the part's tiles
  uint64_t first;  // first splitter of the part's range in `fine`
};
__device__ __forceinline__ SrPartInfo sr_part_info(const SrLevel &L, uint32_t p) {
  SrPartInfo r;
  if (L.level == 0) {
    r.eq = false;
    r.owner = p & (kSrXcds - 1);
    r.first = 0;
  } else if (L.level == 1) {
    r.eq = p & 1u;
    r.owner = (p >> 1) & (kSrXcds - 1);
    r.first = (uint64_t)(p >> 1) * L.w1;
  } else {
    const uint32_t d1 = p / L.nbp, d2 = p % L.nbp;
    r.eq = (d1 | d2) & 1u;
    r.owner = ((d1 >> 1) + (d2 >> 1)) & (kSrXcds - 1);
    r.first = (uint64_t)(d1 >> 1) * L.w1 + (uint64_t)(d2 >> 1) * L.w2;
  }
  return r;
}
// the part's splitters into LDS, padded with the largest key so the search needs no bounds
__device__ __forceinline__ void sr_load_splitters(const SrLevel &L, const SrPartInfo &pi, uint32_t S, uint64_t *sp) {
  for (uint32_t k = threadIdx.x; k <= (uint32_t)kSrMaxSplit; k += blockDim.x)
    sp[k] = k < S ? L.fine[pi.first + (uint64_t)(k + 1) * L.stride - 1] : ~0ull;
}
// lower bound by halving steps over the padded splitters (first_step: the largest power of two <= S)
__device__ __forceinline__ uint32_t sr_lower_bound(const uint64_t *sp, uint32_t first_step, uint64_t key) {
  uint32_t pos = 0;
  for (uint32_t step = first_step; step; step >>= 1)
    if (sp[pos + step - 1] < key) pos += step;
  return pos;
}
__device__ __forceinline__ uint32_t sr_first_step(uint32_t S) { return S ? 1u << (31 - __builtin_clz(S)) : 0u; }

// The search of a part: a table over the key range the splitters span, kSrLutCells cells of equal width, says between
// which two splitter indices the lower bound of a cell's keys lies (splitters are quantiles: for most data a cell holds
// one or none, and the 8 dependent LDS reads of a plain binary search -- the partition kernels were bound by their
// instruction count, ~90 a key -- become one table read and a step or two).
constexpr int kSrLutCells = 256;
struct SrSearch {
  uint64_t dom_lo, dom_hi;  // the table's key range: splitters `edge` and S - 1 - edge
  uint32_t S, edge;
  int shift;
};
// sp[] holds the part's splitters (a barrier has passed); builds lut[] and passes a barrier.  The table spans the
// splitters without the outermost sixteenth on either side: quantiles of doubles are anything but evenly spread over
// the KEY range where the values cross many exponents (uniform [0, 1): half of all splitters lie in the top 1 / 1022 of
// it), between the 6th and the 94th percentile a few exponents remain; keys outside search the few splitters there.
__device__ __forceinline__ SrSearch sr_build_search(const uint64_t *sp, uint32_t S, uint32_t *lut) {
  SrSearch q;
  q.S = S;
  q.edge = 0;
  q.dom_lo = q.dom_hi = 0;
  q.shift = 0;
  if (S) {
    q.edge = S / 16;
    q.dom_lo = sp[q.edge];
    q.dom_hi = sp[S - 1 - q.edge];
    const uint64_t width = q.dom_hi - q.dom_lo;
    q.shift = width < (uint64_t)kSrLutCells ? 0 : (64 - (int)__builtin_clzll(width)) - 8;
    const uint32_t fs = sr_first_step(S);
    for (uint32_t c = threadIdx.x; c < (uint32_t)kSrLutCells; c += blockDim.x) {
      const uint64_t start = q.dom_lo + ((uint64_t)c << q.shift);
      const uint64_t next = start + (1ull << q.shift);
      const bool wraps = next < start || start < q.dom_lo;
      const uint32_t lb0 = start < q.dom_lo ? S : sr_lower_bound(sp, fs, start);
      const uint32_t lb1 = wraps ? S : sr_lower_bound(sp, fs, next);
      lut[c] = lb0 | (lb1 << 16);
    }
  }
  __syncthreads();
  return q;
}
// buckets of N keys at once (their LDS reads in flight together): 2 * (splitters below the key) + (it equals the next)
template <int N>
__device__ __forceinline__ void sr_buckets(const uint64_t *sp, const uint32_t *lut, const SrSearch &q,
                                           const uint64_t (&key)[N], uint32_t (&b)[N]) {
  uint32_t lo[N], hi[N];
#pragma unroll
  for (int u = 0; u < N; u++) {
    const bool in = key[u] >= q.dom_lo && key[u] <= q.dom_hi;
    const uint32_t e = lut[in ? (uint32_t)((key[u] - q.dom_lo) >> q.shift) : 0u];
    // (below the table: at most `edge` splitters are smaller; above it: all but the last `edge` are)
    lo[u] = in ? (e & 0xFFFFu) : (key[u] < q.dom_lo ? 0u : q.S - q.edge);
    hi[u] = in ? (e >> 16) : (key[u] < q.dom_lo ? q.edge : q.S);
  }
  // (selects, not branches: a step is a read and four selects for every key of the lane, open or not -- with an `if`
  // per key the loop was mostly exec-mask bookkeeping, and these kernels are bound by the instructions they issue)
  for (;;) {
    bool open = false;
#pragma unroll
    for (int u = 0; u < N; u++) open = open || lo[u] < hi[u];
    if (__builtin_amdgcn_ballot_w64(open) == 0) break;
#pragma unroll
    for (int u = 0; u < N; u++) {
      const bool go = lo[u] < hi[u];
      const uint32_t mid = (lo[u] + hi[u]) >> 1;  // (< S when go; <= S otherwise: sp[] is padded to kSrMaxSplit + 1)
      const bool less = sp[mid] < key[u];
      lo[u] = (go && less) ? mid + 1 : lo[u];
      hi[u] = (go && !less) ? mid : hi[u];
    }
  }
#pragma unroll
  for (int u = 0; u < N; u++) {
    const uint32_t at = lo[u] < q.S ? lo[u] : q.S;  // (sp[S] is the padding: the largest key, never a real match below)
    b[u] = 2u * lo[u] + ((lo[u] < q.S && sp[at] == key[u]) ? 1u : 0u);
  }
}

// exclusive scan of a[0 .. 8 * THREADS) in LDS (entries past n read as 0), `add` on top; a[n] = the total + add when
// `sentinel`.  A barrier has passed since a[] was written; one passes before this returns.
template <int THREADS>
__device__ __forceinline__ void sr_block_scan8(uint32_t *a, uint32_t n, uint32_t add, bool sentinel, uint32_t *wsum) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  uint32_t v[8], s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t i = tid * 8 + k;
    v[k] = i < n ? a[i] : 0u;
    s += v[k];
  }
  uint32_t incl = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d, 64);
    if (lane >= (uint32_t)d) incl += up;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t run = add + incl - s;
  for (uint32_t w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t i = tid * 8 + k;
    if (i < n) a[i] = run;
    run += v[k];
  }
  if (sentinel && tid == THREADS - 1) a[n] = run;  // (the last thread's run has passed every entry)
  __syncthreads();
}

// ---- sample / splitters -------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t sr_mix(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// one key from each of ns stretches that together cover ALL n positions (an input in order -- the second ranking of a
// correlated pair reads its keys grouped by the first -- has its largest keys at the end: a sample that stops short of
// it leaves them to one huge last bucket)
__global__ void sr_sample_kernel(const uint64_t *keys, uint64_t n, uint64_t ns, uint64_t *out, int source) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ns) return;
  // n = ns * q + r: the first r stretches hold q + 1 keys, the others q
  const uint64_t q = n / ns, r = n - q * ns;  // (wave-uniform: scalar divisions)
  const uint64_t b = k * q + (k < r ? k : r), len = q + (k < r ? 1 : 0);
  uint64_t pos = b + (len > 1 ? sr_mix(k + 0x9E3779B97F4A7C15ull) % len : 0);
  if (pos >= n) pos = n - 1;
  out[k] = sr_source_key(keys[pos], source);
}
__global__ void sr_pick_kernel(const uint64_t *sorted, uint64_t count, uint32_t every, uint64_t *fine) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < count) fine[j] = sorted[(j + 1) * every];
}

// ---- a partition pass ---------------------------------------------------------------------------------------------
// pass 0: the input as stretches of whole tiles; and the stretches of the sample that were drawn from them (`ns` sample
// keys, in the order of the positions they came from; a stretch's share of the sample starts with the first sample key
// drawn at or behind its first position -- sharp to one sample key, the room of a bucket is generous by more)
__global__ void sr_first_parts_kernel(uint32_t *part_start, uint32_t nparts, uint32_t per_part, uint32_t n,
                                      uint32_t *sample_start, uint32_t ns) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > nparts) return;
  const uint64_t at = (uint64_t)p * per_part;
  const uint32_t v = (p == nparts || at > n) ? n : (uint32_t)at;
  part_start[p] = v;
  if (sample_start) {
    // sr_sample_kernel: n = ns * q + r, the first r sample keys come from q + 1 positions each, the others from q
    const uint64_t q = ns ? n / ns : 0, r = ns ? n - q * ns : 0, edge = r * (q + 1);
    uint64_t k = ns;
    if (v < n && q) k = v < edge ? v / (q + 1) : r + (v - edge) / q;
    sample_start[p] = k > ns ? ns : (uint32_t)k;
  }
}

// the tiles of every part, listed per XCD: a part between two splitters belongs to ONE XCD (the runs that fill a
// bucket's lines then meet in one L2); an equality part is only copied, its tiles go round all of them.  One wave per
// piece of a part; the order of the lists does not matter.
__global__ __launch_bounds__(256) void sr_tiles_kernel(SrLevel L) {
  if (*(volatile const uint32_t *)L.status) return;
  const uint32_t w = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  if (w >= L.nparts * L.npieces) return;
  const uint32_t p = w % L.nparts, piece = w / L.nparts;
  const uint32_t b = L.pbeg[(size_t)piece * L.pstride + p], e = L.pend[(size_t)piece * L.pstride + p];
  if (e <= b) return;
  if (lane == 0) atomicAdd(&L.part_size[p], e - b);
  const uint32_t nt = (uint32_t)(((uint64_t)(e - b) + kSrTile - 1) / kSrTile);
  const SrPartInfo pi = sr_part_info(L, p);
  auto ref = [&](uint32_t k) {
    const uint32_t at = b + k * (uint32_t)kSrTile;
    return SrTileRef{p, at, e - at > (uint32_t)kSrTile ? (uint32_t)kSrTile : e - at};
  };
  if (!pi.eq) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&L.tile_count[pi.owner], nt);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    SrTileRef *out = L.tiles + (size_t)pi.owner * L.tile_cap + base;
    for (uint32_t k = lane; k < nt; k += 64) out[k] = ref(k);
  } else {
    for (uint32_t x = 0; x < (uint32_t)kSrXcds; x++) {
      if (nt <= x) break;
      const uint32_t ntx = (nt - x + kSrXcds - 1) / kSrXcds;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&L.tile_count[x], ntx);
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      SrTileRef *out = L.tiles + (size_t)x * L.tile_cap + base;
      for (uint32_t k = lane; k < ntx; k += 64) out[k] = ref(x + k * kSrXcds);
    }
  }
}

// keys per (part, bucket): a workgroup keeps counting in LDS while its tiles stay in one part; the next tile's keys
// are requested before this tile's are searched
__global__ __launch_bounds__(kSrPartThreads) void sr_count_kernel(SrLevel L) {
  __shared__ uint64_t sp[kSrMaxSplit + 1];
  __shared__ uint32_t lut[kSrLutCells];
  __shared__ uint32_t hist[kSrMaxNb + 1];
  if (*(volatile const uint32_t *)L.status) return;
  const uint32_t tid = threadIdx.x;
  const uint32_t x = blockIdx.x & (kSrXcds - 1), W = gridDim.x >> 3;
  const uint32_t nt = L.tile_count[x];
  // `group` workgroups walk a stretch of their XCD's list side by side
  const uint32_t jstep = (L.group == 0 || L.group > W) ? W : L.group, groups = (W + jstep - 1) / jstep;
  const uint32_t len = (nt + groups - 1) / groups, g0 = ((blockIdx.x >> 3) / jstep) * len;
  const uint32_t j0 = g0 + (blockIdx.x >> 3) % jstep, j1 = g0 + len < nt ? g0 + len : nt;
  const SrTileRef *tiles = L.tiles + (size_t)x * L.tile_cap;
  uint32_t cur = ~0u;
  SrSearch q;
  q.S = 0;
  auto tile_end = [&](const SrTileRef &r) { return r.begin + r.len; };
  auto fetch = [&](const SrTileRef &r, uint32_t end, uint64_t (&k)[kSrPartPer]) {
#pragma unroll
    for (int u = 0; u < kSrPartPer; u++) {
      const uint64_t i = (uint64_t)r.begin + u * kSrPartThreads + tid;
      k[u] = i < end ? sr_source_key(__builtin_nontemporal_load(L.keys_in + i), L.key_source) : 0ull;
    }
  };
  auto flush = [&]() {
    __syncthreads();
    if (q.S) {
      for (uint32_t v = tid; v < 2 * q.S + 1; v += kSrPartThreads) {
        const uint32_t h = hist[v];
        if (h) atomicAdd(&L.tot[(size_t)cur * L.nb + v], h);
      }
      __syncthreads();
    }
  };
  bool pays = true;   // (wave-uniform) lanes of this wave have lately agreed on buckets: sr_claim / sr_tally
  uint32_t trip = 0;
  SrTileRef ref = j0 < j1 ? tiles[j0] : SrTileRef{0, 0, 0};
  uint32_t end = j0 < j1 ? tile_end(ref) : 0;
  uint64_t key[kSrPartPer];
  if (j0 < j1) fetch(ref, end, key);
  for (uint32_t j = j0; j < j1; j += jstep) {
    const bool more = j + jstep < j1;
    const SrTileRef nref = more ? tiles[j + jstep] : SrTileRef{0, 0, 0};
    const uint32_t nend = more ? tile_end(nref) : 0;
    uint64_t nkey[kSrPartPer];
    if (more) fetch(nref, nend, nkey);
    if (ref.part != cur) {
      flush();
      cur = ref.part;
      const SrPartInfo pi = sr_part_info(L, cur);
      const uint32_t S = pi.eq ? 0u : L.split_count;
      sr_load_splitters(L, pi, S, sp);
      for (uint32_t v = tid; v <= (uint32_t)kSrMaxNb; v += kSrPartThreads) hist[v] = 0;
      __syncthreads();
      q = sr_build_search(sp, S, lut);
    }
    if (q.S) {  // (an equality part is one bucket: the offsets kernel knows its size)
      uint32_t bk[kSrPartPer];
      sr_buckets<kSrPartPer>(sp, lut, q, key, bk);
      const bool look = pays || (trip & 15u) == 0;
      bool paid = false;
      trip++;
#pragma unroll
      for (int u = 0; u < kSrPartPer; u++)
        sr_tally(hist, bk[u], (uint64_t)ref.begin + u * kSrPartThreads + tid < end, look, &paid);
      if (look) pays = paid;
    }
    ref = nref;
    end = nend;
#pragma unroll
    for (int u = 0; u < kSrPartPer; u++) key[u] = nkey[u];
  }
  flush();
}

// ---- room from the sample instead of a count ----------------------------------------------------------------------
// m sample keys fell into a bucket out of `of` drawn from `size` keys: the bucket holds about size * m / of keys; the
// room is that plus sigmas_x2 / 2 standard deviations of m (a Poisson count: sqrt(m)) plus a few keys' worth for the
// buckets no sample key fell into -- and never more than the keys there are
__device__ __forceinline__ uint32_t sr_room(uint32_t m, uint32_t of, uint32_t size, uint32_t sigmas_x2) {
  if (of == 0) return size;
  const double mm = (double)m;
  const double want = (mm + 0.5 * (double)sigmas_x2 * sqrt(mm) + 2.0 * (double)sigmas_x2) * ((double)size / (double)of);
  return want >= (double)size ? size : (uint32_t)want + 1u;
}
// pass 0: tot[s][v] holds the sample keys of stretch s that fall into bucket v (the counting kernel, run over the
// sample): the room of the stretch's range in the bucket
__global__ void sr_room_first_kernel(SrLevel L, const uint32_t *sample_start, uint32_t sigmas_x2) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= L.nparts * L.nb) return;
  const uint32_t sidx = e / L.nb;
  const uint32_t size = L.pend[sidx] - L.pbeg[sidx], of = sample_start[sidx + 1] - sample_start[sidx];
  L.tot[e] = sr_room(L.tot[e], of, size, sigmas_x2);
}
__device__ __forceinline__ uint32_t sr_sample_lower(const uint64_t *sorted, uint32_t ns, uint64_t v) {  // keys < v
  uint32_t lo = 0, hi = ns;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (sorted[mid] < v)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}
__device__ __forceinline__ uint32_t sr_sample_upper(const uint64_t *sorted, uint32_t ns, uint64_t v) {  // keys <= v
  uint32_t lo = 0, hi = ns;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (sorted[mid] <= v)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}
// later passes: a part is the keys between two splitters whatever order they came in, so the sorted sample's share
// between a bucket's bounds is the bucket's share of the part.  One thread per (part, bucket).
__global__ void sr_room_kernel(SrLevel L, const uint64_t *sorted, uint32_t ns, uint64_t n_fine, uint32_t sigmas_x2) {
  if (*(volatile const uint32_t *)L.status) return;
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (uint64_t)L.nparts * L.nb) return;
  const uint32_t p = (uint32_t)(e / L.nb), v = (uint32_t)(e % L.nb);
  const SrPartInfo pi = sr_part_info(L, p);
  const uint32_t size = L.part_size[p];
  if (pi.eq || size == 0) {
    L.tot[e] = (pi.eq && v == 0) ? size : 0u;
    return;
  }
  const uint32_t S = L.split_count;
  if (v > 2 * S) {
    L.tot[e] = 0;
    return;
  }
  const uint64_t width = (uint64_t)(S + 1) * L.stride;
  // sample keys strictly inside the part
  const bool has_lo = pi.first > 0, has_hi = pi.first + width - 1 < n_fine;
  const uint32_t part_a = has_lo ? sr_sample_upper(sorted, ns, L.fine[pi.first - 1]) : 0u;
  const uint32_t part_b = has_hi ? sr_sample_lower(sorted, ns, L.fine[pi.first + width - 1]) : ns;
  const uint32_t of = part_b > part_a ? part_b - part_a : 0u;
  const uint32_t k = v >> 1;
  uint32_t a, b;
  if (v & 1u) {  // the keys equal to splitter k
    const uint64_t val = L.fine[pi.first + (uint64_t)(k + 1) * L.stride - 1];
    a = sr_sample_lower(sorted, ns, val);
    b = sr_sample_upper(sorted, ns, val);
  } else {
    a = k == 0 ? part_a : sr_sample_upper(sorted, ns, L.fine[pi.first + (uint64_t)k * L.stride - 1]);
    b = k == S ? part_b : sr_sample_lower(sorted, ns, L.fine[pi.first + (uint64_t)(k + 1) * L.stride - 1]);
  }
  L.tot[e] = sr_room(b > a ? b - a : 0u, of, size, sigmas_x2);
}

// ---- where the buckets start ----------------------------------------------------------------------------------------
// pass 0: bucket v starts at (room of the lower buckets); inside it the stretches' ranges follow one another
__global__ __launch_bounds__(512) void sr_offsets_first_kernel(SrLevel L) {
  __shared__ uint32_t start[kSrMaxNb + 9];
  __shared__ uint32_t wsum[8];
  __shared__ unsigned long long all;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) all = 0;
  uint32_t total = 0;
  if (tid < L.nb)
    for (uint32_t p = 0; p < L.nparts; p++) total += L.tot[(size_t)p * L.nb + tid];
  for (uint32_t v = tid; v <= (uint32_t)kSrMaxNb + 8; v += 512) start[v] = 0;
  __syncthreads();
  if (tid < L.nb) {
    start[tid] = total;
    atomicAdd(&all, (unsigned long long)total);
  }
  __syncthreads();
  if (all > (unsigned long long)L.out_cap) {  // (the sum in 64 bits: the scan below would wrap)
    if (tid == 0) atomicOr(L.status, 1u << (4 * L.level));
    return;
  }
  sr_block_scan8<512>(start, L.nb, 0u, true, wsum);
  if (tid < L.nb) {
    uint32_t run = start[tid];
    L.bstart[tid] = run;
    for (uint32_t p = 0; p < L.nparts; p++) {
      const size_t at = (size_t)p * L.nb + tid;
      const uint32_t room = L.tot[at];
      L.cursor[at] = run;
      L.pbegin[at] = run;
      run += room;
      L.limit[at] = run;
    }
    L.tot[tid] = total;  // (row 0 now holds the buckets' sizes: what the last pass's work list reads)
  }
  if (tid == 0) L.bstart[L.nb] = start[L.nb];
}

// later passes, one wave per part: the room its buckets take (an equality part is one bucket: its keys)
__global__ __launch_bounds__(512) void sr_part_totals_kernel(SrLevel L) {
  if (*(volatile const uint32_t *)L.status) return;
  const uint32_t p = (blockIdx.x * 512u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  if (p >= L.nparts) return;
  const SrPartInfo pi = sr_part_info(L, p);
  uint32_t *tot = L.tot + (size_t)p * L.nb;
  unsigned long long s = 0;
  if (pi.eq) {
    for (uint32_t i = lane; i < L.nb; i += 64) tot[i] = i == 0 ? L.part_size[p] : 0u;
    s = lane == 0 ? L.part_size[p] : 0;
  } else {
    for (uint32_t i = lane; i < L.nb; i += 64) s += tot[i];
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
  if (lane == 0) {
    if (s > 0xFFFFFFFFull) {
      atomicOr(L.status, 1u << (4 * L.level));
      s = 0;
    }
    L.part_total[p] = (uint32_t)s;
  }
}
// ... one after the other in the output (a single workgroup: a few ten thousand parts at most)
__global__ __launch_bounds__(1024) void sr_scan_parts_kernel(SrLevel L) {
  __shared__ unsigned long long wsum[16];
  __shared__ unsigned long long carry;
  if (*(volatile const uint32_t *)L.status) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < L.nparts; base += 1024) {
    const uint32_t p = base + tid;
    const unsigned long long v = p < L.nparts ? L.part_total[p] : 0ull;
    unsigned long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long up = __shfl_up(incl, d, 64);
      if (lane >= (uint32_t)d) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned long long before = carry;
    for (uint32_t w = 0; w < wave; w++) before += wsum[w];
    const unsigned long long mine = before + incl - v;
    if (p < L.nparts) L.out_base[p] = mine > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)mine;
    __syncthreads();
    if (tid == 1023) carry = before + incl;
    __syncthreads();
  }
  if (tid == 0) {
    L.out_base[L.nparts] = carry > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)carry;
    if (carry > (unsigned long long)L.out_cap) atomicOr(L.status, 1u << (4 * L.level));
  }
}
// ... and the part's buckets from there on.  One wave per part.
__global__ __launch_bounds__(512) void sr_offsets_kernel(SrLevel L) {
  if (*(volatile const uint32_t *)L.status) return;
  const uint32_t p = (blockIdx.x * 512u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  if (p >= L.nparts) return;
  const uint32_t b = L.out_base[p];
  const size_t row = (size_t)p * L.nb;
  const uint32_t *tot = L.tot + row;
  uint32_t *cur = L.cursor + row, *bs = L.bstart + row, *lim = L.limit + row;
  uint32_t v[8], s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t i = lane * 8 + k;
    v[k] = i < L.nb ? tot[i] : 0u;
    s += v[k];
  }
  uint32_t incl = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d, 64);
    if (lane >= (uint32_t)d) incl += up;
  }
  uint32_t run = b + incl - s;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t i = lane * 8 + k;
    if (i < L.nb) {
      cur[i] = run;
      bs[i] = run;
      lim[i] = run + v[k];
    }
    run += v[k];
  }
}

// the pass proper: a tile's keys (and payloads) grouped by bucket in LDS, every run placed with one reservation; the
// next tile's keys are requested before this tile's are searched
// (three workgroups a CU are six waves a SIMD: 80 registers each -- at 82 the third workgroup does not fit and the
// pass takes 10.5 instead of 7 ms)
// KS / PS: what the keys / the 8-byte payloads are (SrSource; column values in pass 0 of a job over a lent batch) -- an
// instance per kind: with the kind a run-time value the Int64 conversion's registers pushed the kernel over its 80
template <int PB, int KS, int PS>
__global__ __launch_bounds__(kSrPartThreads) __attribute__((amdgpu_waves_per_eu(6, 8))) void sr_scatter_kernel(SrLevel L) {
  __shared__ uint64_t sp[kSrMaxSplit + 1];
  __shared__ uint64_t stage_k[kSrTile];
  __shared__ uint64_t stage_p8[PB == 8 ? kSrTile : 1];
  __shared__ uint32_t stage_p4[PB == 4 ? kSrTile : 1];
  __shared__ uint16_t ids[kSrTile];
  __shared__ uint32_t lut[kSrLutCells];
  __shared__ uint32_t hist[kSrMaxNb + 1];
  __shared__ uint32_t delta[kSrMaxNb + 1];
  __shared__ uint32_t room_end[kSrMaxNb + 1];  // where the buckets of the part at hand end
  __shared__ uint32_t wsum[kSrPartThreads / 64];
  __shared__ uint32_t s_full, s_base;
  if (*(volatile const uint32_t *)L.status) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) s_full = 0;
  const uint32_t x = blockIdx.x & (kSrXcds - 1), W = gridDim.x >> 3;
  const uint32_t nt = L.tile_count[x];
  // `group` workgroups walk a stretch of their XCD's list side by side: the runs that complete a bucket's lines follow
  // one another closely, and few buckets are open per L2 at a time
  const uint32_t jstep = (L.group == 0 || L.group > W) ? W : L.group, groups = (W + jstep - 1) / jstep;
  const uint32_t len = (nt + groups - 1) / groups, g0 = ((blockIdx.x >> 3) / jstep) * len;
  const uint32_t j0 = g0 + (blockIdx.x >> 3) % jstep, j1 = g0 + len < nt ? g0 + len : nt;
  const SrTileRef *tiles = L.tiles + (size_t)x * L.tile_cap;
  const uint64_t *pin8 = (const uint64_t *)L.pay_in;
  const uint32_t *pin4 = (const uint32_t *)L.pay_in;
  uint64_t *pout8 = (uint64_t *)L.pay_out;
  uint32_t *pout4 = (uint32_t *)L.pay_out;
  uint32_t cur = ~0u, S = 0;
  SrSearch q;
  q.S = 0;
  auto tile_end = [&](const SrTileRef &r) { return r.begin + r.len; };
  auto fetch = [&](const SrTileRef &r, uint32_t end, uint64_t (&k)[kSrPartPer], uint64_t (&q8)[PB == 8 ? kSrPartPer : 1],
                   uint32_t (&q4)[PB == 4 ? kSrPartPer : 1]) {
#pragma unroll
    for (int u = 0; u < kSrPartPer; u++) {
      const uint64_t i = (uint64_t)r.begin + u * kSrPartThreads + tid;
      const bool ok = i < end;
      k[u] = ok ? __builtin_nontemporal_load(L.keys_in + i) : 0ull;
      if (PB == 8) q8[u] = ok ? __builtin_nontemporal_load(pin8 + i) : 0ull;
      if (KS != kSrKeys) k[u] = sr_source_key(k[u], KS);
      if (PB == 8 && PS != kSrKeys) q8[u] = sr_source_key(q8[u], PS);
      if (PB == 4) q4[u] = (ok && pin4) ? __builtin_nontemporal_load(pin4 + i) : (uint32_t)i;
    }
  };
  bool pays = true;   // (wave-uniform) lanes of this wave have lately agreed on buckets: sr_claim / sr_tally
  uint32_t trip = 0;
  SrTileRef ref = j0 < j1 ? tiles[j0] : SrTileRef{0, 0, 0};
  uint32_t end = j0 < j1 ? tile_end(ref) : 0;
  uint64_t key[kSrPartPer];
  uint64_t p8[PB == 8 ? kSrPartPer : 1];
  uint32_t p4[PB == 4 ? kSrPartPer : 1];
  if (j0 < j1) fetch(ref, end, key, p8, p4);
  for (uint32_t j = j0; j < j1; j += jstep) {
    const bool more = j + jstep < j1;
    const SrTileRef nref = more ? tiles[j + jstep] : SrTileRef{0, 0, 0};
    const uint32_t nend = more ? tile_end(nref) : 0;
    uint64_t nkey[kSrPartPer];
    uint64_t np8[PB == 8 ? kSrPartPer : 1];
    uint32_t np4[PB == 4 ? kSrPartPer : 1];
    if (more) fetch(nref, nend, nkey, np8, np4);
    if (ref.part != cur) {
      __syncthreads();
      cur = ref.part;
      const SrPartInfo pi = sr_part_info(L, cur);
      S = pi.eq ? 0u : L.split_count;
      sr_load_splitters(L, pi, S, sp);
      for (uint32_t v = tid; v <= (uint32_t)kSrMaxNb; v += kSrPartThreads) {
        hist[v] = 0;
        room_end[v] = v < L.nb ? L.limit[(size_t)cur * L.nb + v] : 0u;
      }
      __syncthreads();
      q = sr_build_search(sp, S, lut);
    }
    if (S == 0) {  // one bucket: the tile is one run
      __syncthreads();
      if (tid == 0) {
        const size_t at = (size_t)cur * L.nb;
        const uint32_t base = atomicAdd(&L.cursor[at], ref.len);
        s_base = base;
        if (base + ref.len > L.limit[at] || base + ref.len < base) s_full = 1;
      }
      __syncthreads();
      if (s_full) break;
      const uint32_t shift = s_base - ref.begin;
#pragma unroll
      for (int u = 0; u < kSrPartPer; u++) {
        const uint64_t i = (uint64_t)ref.begin + u * kSrPartThreads + tid;
        if (i < end) {
          const uint32_t at = (uint32_t)i + shift;
          L.keys_out[at] = key[u];
          if (PB == 8) pout8[at] = p8[u];
          if (PB == 4) pout4[at] = p4[u];
        }
      }
    } else {
      uint32_t bk[kSrPartPer], rk[kSrPartPer];
      sr_buckets<kSrPartPer>(sp, lut, q, key, bk);
      // (not with 8-byte payloads: that instance has no register to spare for the two words of state)
      const bool look = PB == 8 || pays || (trip & 15u) == 0;
      bool paid = false;
      trip++;
#pragma unroll
      for (int u = 0; u < kSrPartPer; u++)  // (the key's place inside its run)
        rk[u] = sr_claim(hist, bk[u], (uint64_t)ref.begin + u * kSrPartThreads + tid < end, look, &paid);
      if (look) pays = paid;
      __syncthreads();
      // one reservation per run -- its round trip runs under the scan and the regrouping: only the stores need it
      const uint32_t h = tid < 2 * S + 1 ? hist[tid] : 0u;
      uint32_t reserved = 0;
      if (h) {
        const size_t cell = (size_t)cur * L.nb + tid;
        reserved = atomicAdd(&L.cursor[cell], h);
        if (reserved + h > room_end[tid] || reserved + h < reserved) s_full = 1;  // (the barrier below publishes it)
      }
      uint32_t incl = h;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += up;
      }
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      if (s_full) break;  // a bucket had less room than keys: nothing of this job counts any more
      uint32_t excl = incl - h;
      for (uint32_t w = 0; w < wave; w++) excl += wsum[w];
      hist[tid] = excl;  // (tid <= kSrMaxNb)
      __syncthreads();
#pragma unroll
      for (int u = 0; u < kSrPartPer; u++) {
        if ((uint64_t)ref.begin + u * kSrPartThreads + tid < end) {
          const uint32_t at = hist[bk[u]] + rk[u];
          stage_k[at] = key[u];
          if (PB == 8) stage_p8[at] = p8[u];
          if (PB == 4) stage_p4[at] = p4[u];
          ids[at] = (uint16_t)bk[u];
        }
      }
      delta[tid] = reserved - excl;
      __syncthreads();
      hist[tid] = 0;  // (nobody reads the counts any more; the barrier below stands before the next tile's)
      const uint32_t count = end - ref.begin;
      for (uint32_t q = tid; q < count; q += kSrPartThreads) {
        const uint32_t at = q + delta[ids[q]];
        L.keys_out[at] = stage_k[q];
        if (PB == 8) pout8[at] = stage_p8[q];
        if (PB == 4) pout4[at] = stage_p4[q];
      }
      __syncthreads();
    }
    ref = nref;
    end = nend;
#pragma unroll
    for (int u = 0; u < kSrPartPer; u++) {
      key[u] = nkey[u];
      if (PB == 8) p8[u] = np8[u];
      if (PB == 4) p4[u] = np4[u];
    }
  }
  if (s_full && tid == 0) atomicOr(L.status, 2u << (4 * L.level));  // (sortrank.h: the status word's bits)
}

// ---- the last pass's work lists ------------------------------------------------------------------------------------
// `small`: buckets the 256-thread kernel ranks (and the pieces of equality buckets), `large`: the others
__global__ __launch_bounds__(256) void sr_items_kernel(SrLevel L, uint64_t n_buckets, uint32_t small_cap, SrItem *small,
                                                        uint32_t *n_small, SrItem *large, uint32_t *n_large, SrItem *tiny,
                                                        uint32_t *n_tiny) {
  __shared__ uint32_t wsum_t[4];
  __shared__ uint32_t block_base_t;
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t block_base;
  if (*(volatile const uint32_t *)L.status) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint64_t rows = L.level == 0 ? 1ull : (uint64_t)L.nparts;  // (pass 0: row 0 holds the buckets' sizes)
  const uint32_t e = blockIdx.x * 256u + tid, total = (uint32_t)(rows * L.nb);  // (at most 511^3 entries)
  uint32_t cnt = 0, st = 0;
  bool eq = false;
  if (e < total) {
    const uint32_t p = e / L.nb, v = e - p * L.nb;
    const bool part_eq = L.level == 0 ? false : sr_part_info(L, p).eq;
    if (!(part_eq && v > 0)) cnt = L.tot[e];
    eq = part_eq || (v & 1u);
    st = L.bstart[e];
  }
  // a bucket between two splitters knows its key range (the small kernel bins by it); the first and the last bucket of
  // all do not: they go with the large ones, whose kernel takes the range from the keys
  uint64_t lo = 0, hi = 0;
  bool edge = false;
  if (cnt && !eq) {
    const uint32_t p = e / L.nb, v = e - p * L.nb;
    const uint64_t g = (L.level == 0 ? 0ull : sr_part_info(L, p).first) + (v >> 1);  // (the last pass: stride 1)
    edge = g == 0 || g + 1 >= n_buckets;
    if (!edge) {
      lo = L.fine[g - 1];
      hi = L.fine[g];
    }
  }
  const bool big = (cnt > small_cap || edge) && !eq;
  const bool is_tiny = eq && cnt > 0 && cnt <= kSrEqTiny;
  const uint32_t ni = (cnt == 0 || big || is_tiny) ? 0u : (eq ? (cnt + kSrEqPiece - 1) / kSrEqPiece : 1u);
  uint32_t incl = ni, incl_t = is_tiny ? 1u : 0u;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d, 64), up_t = __shfl_up(incl_t, d, 64);
    if (lane >= (uint32_t)d) {
      incl += up;
      incl_t += up_t;
    }
  }
  if (lane == 63) {
    wsum[wave] = incl;
    wsum_t[wave] = incl_t;
  }
  __syncthreads();
  uint32_t off = incl - ni, off_t = incl_t - (is_tiny ? 1u : 0u);
  for (uint32_t w = 0; w < wave; w++) {
    off += wsum[w];
    off_t += wsum_t[w];
  }
  if (tid == 0) {
    const uint32_t all = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    block_base = all ? atomicAdd(n_small, all) : 0u;
    const uint32_t all_t = wsum_t[0] + wsum_t[1] + wsum_t[2] + wsum_t[3];
    block_base_t = all_t ? atomicAdd(n_tiny, all_t) : 0u;
  }
  __syncthreads();
  if (big) large[atomicAdd(n_large, 1u)] = SrItem{st, cnt, st, 0u, 0ull, 0ull};
  if (is_tiny) tiny[block_base_t + off_t] = SrItem{st, cnt, st, 1u, 0ull, 0ull};
  if (ni == 0) return;
  SrItem *out = small + block_base + off;
  if (!eq) {
    out[0] = SrItem{st, cnt, st, 0u, lo, hi};
  } else {
    for (uint32_t k = 0; k < ni; k++) {
      const uint32_t at = k * kSrEqPiece;
      out[k] = SrItem{st + at, cnt - at < kSrEqPiece ? cnt - at : kSrEqPiece, st, 1u, 0ull, 0ull};
    }
  }
}
// (a job of one bucket: no splitters, hence no key range: the large kernel's)
__global__ void sr_one_item_kernel(SrItem *large, uint32_t *counters, uint32_t n) {
  large[0] = SrItem{0u, n, 0u, 0u, 0ull, 0ull};
  counters[0] = 0;
  counters[1] = 1;
  counters[2] = 0;
}

// ---- the last pass -------------------------------------------------------------------------------------------------
struct SrSumAcc {
  // sum(a), sum(b) stay below 2^64 (ranks <= n < 2^32: at most n (n + 1) / 2); the products need 128 bits
  unsigned long long lo[5], hi[3];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int k = 0; k < 5; k++) lo[k] = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) hi[k] = 0;
  }
  __device__ __forceinline__ void add(unsigned long long a, unsigned long long b) {
    lo[0] += a;
    lo[1] += b;
    const unsigned __int128 t[3] = {(unsigned __int128)a * a, (unsigned __int128)b * b, (unsigned __int128)a * b};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const unsigned long long tl = (unsigned long long)t[k], th = (unsigned long long)(t[k] >> 64);
      const unsigned long long s = lo[2 + k] + tl;
      hi[k] += th + (s < tl ? 1ull : 0ull);
      lo[2 + k] = s;
    }
  }
};
// the workgroup's sums: waves through shuffles, then LDS; every workgroup writes its slot
template <int THREADS>
__device__ __forceinline__ void sr_store_sums(const SrSumAcc &acc, RankSums *slot) {
  __shared__ RankSums sh[THREADS / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 5; k++) {
    unsigned long long lo = acc.lo[k], hi = k >= 2 ? acc.hi[k - 2] : 0ull;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const unsigned long long olo = __shfl_down(lo, d, 64), ohi = __shfl_down(hi, d, 64);
      const unsigned long long s = lo + olo;
      hi += ohi + (s < lo ? 1ull : 0ull);
      lo = s;
    }
    if (lane == 0) {
      sh[tid >> 6].exact_lo[k] = lo;
      sh[tid >> 6].exact_hi[k] = hi;
    }
  }
  __syncthreads();
  if (tid == 0) {
    RankSums r = sh[0];
    for (int wv = 1; wv < THREADS / 64; wv++)
      for (int k = 0; k < 5; k++) {
        const unsigned long long s = r.exact_lo[k] + sh[wv].exact_lo[k];
        r.exact_hi[k] += sh[wv].exact_hi[k] + (s < r.exact_lo[k] ? 1ull : 0ull);
        r.exact_lo[k] = s;
      }
    // UInt64 arithmetic that wraps (the reference's) is the exact sum modulo 2^64: ranks are below 2^33, and a
    // product reduced modulo 2^64 before or after the summation is the same residue
    for (int k = 0; k < 5; k++) r.wrapped[k] = r.exact_lo[k];
    *slot = r;
  }
}

// does the sink read the payload where the key lies?
template <int PB, int SINK>
struct SrWantsPay {
  static constexpr bool k4 = (SINK == kSrSums || SINK == kSrRankScatter || (SINK == kSrSorted && PB == 4));
  static constexpr bool k8 = SINK == kSrSorted && PB == 8;
};

template <int PB, int SINK>
__device__ __forceinline__ void sr_emit(const SrFinal &F, const SrItem &it, uint64_t g, uint64_t key, uint32_t pay4,
                                        uint64_t pay8, uint32_t lt, uint32_t eqb, SrSumAcc &acc) {
  if (SINK == kSrSorted) {
    const uint64_t at = (it.flags & 1u) ? g : (uint64_t)it.start + lt + eqb;
    F.out_keys[at] = key;
    if (PB == 8) ((uint64_t *)F.out_pay)[at] = pay8;
    if (PB == 4) ((uint32_t *)F.out_pay)[at] = pay4;
  } else if (SINK == kSrRank32) {
    F.rank32[g] = it.rank_base + lt;
  } else if (SINK == kSrSums) {
    acc.add((unsigned long long)pay4 + 1ull, (unsigned long long)it.rank_base + lt + 1ull);
  } else {
    F.rank_out[pay4] = F.ext_base + it.rank_base + lt + 1ull;
  }
}
template <int PB, int SINK>
__device__ __forceinline__ uint32_t sr_pay4(const SrFinal &F, uint64_t g) {
  if (!SrWantsPay<PB, SINK>::k4) return 0u;
  const uint32_t *p = (const uint32_t *)F.pay;
  return p ? __builtin_nontemporal_load(p + g) : (uint32_t)g;
}
template <int PB, int SINK>
__device__ __forceinline__ uint64_t sr_pay8(const SrFinal &F, uint64_t g) {
  if (!SrWantsPay<PB, SINK>::k8) return 0ull;
  return __builtin_nontemporal_load((const uint64_t *)F.pay + g);
}

// buckets of up to kSrFastCap keys between two known splitters, one per workgroup and trip; the next bucket's keys are
// on their way while this one's are ranked.  Five barriers a bucket: the bins of the NEXT bucket are cleared while this
// one's keys are compared (two sets of bins take turns).
template <int PB, int SINK>
__global__ __launch_bounds__(kSrFastThreads) void sr_rank_small_kernel(SrFinal F) {
  constexpr bool kWantOrder = SINK == kSrSorted;  // positions among EQUAL keys matter only when the keys are laid out
  constexpr int kBins = 1 << kSrFastBinBits;
  constexpr bool kPay8 = SrWantsPay<PB, SINK>::k8;
  __shared__ uint64_t skey[kSrFastCap];
  __shared__ uint32_t bins2[2][kBins + 8];
  __shared__ uint32_t wsum[kSrFastThreads / 64];
  const uint32_t tid = threadIdx.x;
  const bool off = *(volatile const uint32_t *)F.status != 0;  // (a failed job: the sums' slots are still written)
  const uint32_t n_items = off ? 0u : *F.n_items, G = gridDim.x;
  SrSumAcc acc;
  acc.clear();
  for (uint32_t v = tid; v < 2u * (kBins + 8); v += kSrFastThreads) (&bins2[0][0])[v] = 0;
  __syncthreads();

  {  // equality buckets of a few keys: RANK() is the bucket's first position for all of them; a thread per bucket
    const uint32_t n_tiny = off ? 0u : *F.n_tiny;
    for (uint32_t i = blockIdx.x * kSrFastThreads + tid; i < n_tiny; i += G * kSrFastThreads) {
      const SrItem t = F.tiny[i];
      for (uint32_t e = 0; e < t.count; e++) {
        const uint64_t g = (uint64_t)t.start + e;
        sr_emit<PB, SINK>(F, t, g, SINK == kSrSorted ? F.keys[g] : 0ull, sr_pay4<PB, SINK>(F, g), sr_pay8<PB, SINK>(F, g),
                          0u, 0u, acc);
      }
    }
  }
  uint32_t w = blockIdx.x, par = 0;
  uint32_t dirty[2] = {0, 0};  // entries of each set of bins that are not zero
  const SrItem none = SrItem{0, 0, 0, 1, 0, 0};
  SrItem it = w < n_items ? F.items[w] : none;
  SrItem it_next = w + G < n_items ? F.items[w + G] : none;
  uint64_t key[kSrFastPer];
  uint32_t pay4[kSrFastPer];
  uint64_t pay8[kPay8 ? kSrFastPer : 1];
  auto fetch = [&](const SrItem &t, uint64_t (&k)[kSrFastPer], uint32_t (&p4)[kSrFastPer],
                   uint64_t (&p8)[kPay8 ? kSrFastPer : 1]) {
    if (t.flags & 1u) return;  // (equality pieces are walked where they lie)
#pragma unroll
    for (int u = 0; u < kSrFastPer; u++) {
      const uint32_t e = tid + u * kSrFastThreads;
      const uint64_t g = (uint64_t)t.start + e;
      k[u] = 0;
      p4[u] = 0;
      if (e < t.count) {
        k[u] = __builtin_nontemporal_load(F.keys + g);
        p4[u] = sr_pay4<PB, SINK>(F, g);
        if (kPay8) p8[u] = sr_pay8<PB, SINK>(F, g);
      }
    }
  };
  fetch(it, key, pay4, pay8);

  for (; w < n_items; w += G) {
    // the trip after this one: its descriptor was requested a trip ago, its keys are requested now
    const SrItem it_next2 = w + 2 * G < n_items ? F.items[w + 2 * G] : none;
    uint64_t nkey[kSrFastPer];
    uint32_t npay4[kSrFastPer];
    uint64_t npay8[kPay8 ? kSrFastPer : 1];
    fetch(it_next, nkey, npay4, npay8);

    if (it.flags & 1u) {  // every key the same: RANK() is the bucket's first position for all of them
      for (uint32_t e = tid; e < it.count; e += kSrFastThreads) {
        const uint64_t g = (uint64_t)it.start + e;
        sr_emit<PB, SINK>(F, it, g, SINK == kSrSorted ? F.keys[g] : 0ull, sr_pay4<PB, SINK>(F, g), sr_pay8<PB, SINK>(F, g),
                          0u, 0u, acc);
      }
    } else {
      uint32_t *bins = bins2[par], *other = bins2[par ^ 1u];
      const uint32_t dirty_other = dirty[par ^ 1u];
      par ^= 1u;
      const uint32_t m = it.count;
      const uint64_t kbase = it.lo + 1, span = it.hi - it.lo - 2;  // (lo < key < hi)
      // as many bins as the bucket may hold keys, but no fewer than 512: scanning and clearing 2048 bins costs what
      // ranking a thousand keys does, and a bucket of the aimed-at size has a thousand
      const int bits = m > 1024u ? kSrFastBinBits : (m > 512u ? kSrFastBinBits - 1 : kSrFastBinBits - 2);
      const uint32_t nbins = 1u << bits;
      dirty[par] = 0;           // (par now names the other set: cleared below)
      dirty[par ^ 1u] = nbins + 1;
      const int shift = span < (uint64_t)nbins ? 0 : (64 - (int)__builtin_clzll(span)) - bits;
      uint32_t slot[kSrFastPer];
#pragma unroll
      for (int u = 0; u < kSrFastPer; u++)
        if (tid + u * kSrFastThreads < m) slot[u] = atomicAdd(&bins[(uint32_t)((key[u] - kbase) >> shift)], 1u);
      __syncthreads();
      sr_block_scan8<kSrFastThreads>(bins, nbins, 0u, true, wsum);
#pragma unroll
      for (int u = 0; u < kSrFastPer; u++)
        if (tid + u * kSrFastThreads < m) {
          slot[u] += bins[(uint32_t)((key[u] - kbase) >> shift)];
          skey[slot[u]] = key[u];
        }
      __syncthreads();
      // (the next bucket's bins, as far as the bucket before this one wrote them)
      for (uint32_t v = tid; v < dirty_other; v += kSrFastThreads) other[v] = 0;
#pragma unroll
      for (int u = 0; u < kSrFastPer; u++) {
        if (tid + u * kSrFastThreads >= m) continue;
        const uint32_t d = (uint32_t)((key[u] - kbase) >> shift);
        const uint32_t b0 = bins[d], b1 = bins[d + 1];
        uint32_t lt = b0, eqb = 0;  // everything in a lower bin is smaller
        if (shift == 0) {          // a bin is one value
          if (kWantOrder) eqb = slot[u] - b0;
        } else {
          for (uint32_t j = b0; j < b1; j++) {
            const uint64_t kk = skey[j];
            lt += kk < key[u] ? 1u : 0u;
            if (kWantOrder) eqb += (kk == key[u] && j < slot[u]) ? 1u : 0u;
          }
        }
        sr_emit<PB, SINK>(F, it, (uint64_t)it.start + tid + u * kSrFastThreads, key[u], pay4[u], kPay8 ? pay8[u] : 0ull,
                          lt, eqb, acc);
      }
      __syncthreads();
    }
    it = it_next;
    it_next = it_next2;
#pragma unroll
    for (int u = 0; u < kSrFastPer; u++) {
      key[u] = nkey[u];
      pay4[u] = npay4[u];
      if (kPay8) pay8[u] = npay8[u];
    }
  }
  if (SINK == kSrSums) sr_store_sums<kSrFastThreads>(acc, F.partials + blockIdx.x);
}

// buckets of any size: F.cap keys at a time, every chunk ranked against every chunk
template <int PB, int SINK>
__global__ __launch_bounds__(kSrSlowThreads) void sr_rank_large_kernel(SrFinal F) {
  constexpr bool kWantOrder = SINK == kSrSorted;
  constexpr int kBins = 1 << kSrSlowBinBits;
  __shared__ uint64_t skey[kSrSlowCap];
  __shared__ uint32_t bins[kBins + 8];
  __shared__ unsigned long long s_mn, s_mx;
  __shared__ uint32_t wsum[kSrSlowThreads / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t n_items = *(volatile const uint32_t *)F.status ? 0u : *F.n_items;
  SrSumAcc acc;
  acc.clear();

  for (uint32_t w = blockIdx.x; w < n_items; w += gridDim.x) {
    const SrItem it = F.items[w];
    const uint32_t C = F.cap, m = it.count, nch = (m + C - 1) / C;
    if (tid == 0) {
      s_mn = ~0ull;
      s_mx = 0ull;
    }
    __syncthreads();
    {
      unsigned long long mn = ~0ull, mx = 0ull;
      for (uint32_t e = tid; e < m; e += kSrSlowThreads) {
        const unsigned long long k = F.keys[(uint64_t)it.start + e];
        mn = k < mn ? k : mn;
        mx = k > mx ? k : mx;
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long omn = __shfl_xor(mn, d, 64), omx = __shfl_xor(mx, d, 64);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
      }
      if (lane == 0) {
        atomicMin(&s_mn, mn);
        atomicMax(&s_mx, mx);
      }
    }
    __syncthreads();
    const uint64_t kbase = s_mn, span = s_mx - s_mn;
    const int shift = span < (uint64_t)kBins ? 0 : (64 - (int)__builtin_clzll(span)) - kSrSlowBinBits;

    for (uint32_t ci = 0; ci < nch; ci++) {
      const uint32_t ilen = m - ci * C < C ? m - ci * C : C;
      uint64_t key[kSrSlowPer];
      uint32_t okm = 0;
#pragma unroll
      for (int u = 0; u < kSrSlowPer; u++) {
        const uint32_t e = tid + u * kSrSlowThreads;
        key[u] = 0;
        if (e < ilen) {
          key[u] = F.keys[(uint64_t)it.start + (uint64_t)ci * C + e];
          okm |= 1u << u;
        }
      }
      uint32_t lt[kSrSlowPer], eqb[kSrSlowPer];
#pragma unroll
      for (int u = 0; u < kSrSlowPer; u++) lt[u] = eqb[u] = 0;

      for (uint32_t cj = 0; cj < nch; cj++) {
        const bool own = cj == ci;
        const uint32_t jlen = m - cj * C < C ? m - cj * C : C;
        uint64_t kj[kSrSlowPer];
        uint32_t okj = 0;
        if (own) {
#pragma unroll
          for (int u = 0; u < kSrSlowPer; u++) kj[u] = key[u];
          okj = okm;
        } else {
#pragma unroll
          for (int u = 0; u < kSrSlowPer; u++) {
            const uint32_t e = tid + u * kSrSlowThreads;
            kj[u] = 0;
            if (e < jlen) {
              kj[u] = F.keys[(uint64_t)it.start + (uint64_t)cj * C + e];
              okj |= 1u << u;
            }
          }
        }
        for (uint32_t v = tid; v < (uint32_t)kBins + 1; v += kSrSlowThreads) bins[v] = 0;
        __syncthreads();
        uint32_t slot[kSrSlowPer];
#pragma unroll
        for (int u = 0; u < kSrSlowPer; u++) {
          const bool ok = (okj >> u) & 1u;
          const uint32_t d = ok ? (uint32_t)((kj[u] - kbase) >> shift) : 0u;
          slot[u] = sr_claim(bins, d, ok);  // (arrival order inside the bin)
        }
        __syncthreads();
        sr_block_scan8<kSrSlowThreads>(bins, kBins, 0u, true, wsum);
#pragma unroll
        for (int u = 0; u < kSrSlowPer; u++)
          if ((okj >> u) & 1u) {
            slot[u] += bins[(uint32_t)((kj[u] - kbase) >> shift)];
            skey[slot[u]] = kj[u];
          }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kSrSlowPer; u++) {
          if (!((okm >> u) & 1u)) continue;
          const uint32_t d = (uint32_t)((key[u] - kbase) >> shift);
          const uint32_t b0 = bins[d], b1 = bins[d + 1];
          lt[u] += b0;  // everything in a lower bin is smaller
          if (shift == 0) {  // a bin is one value
            if (kWantOrder) eqb[u] += cj < ci ? b1 - b0 : (own ? slot[u] - b0 : 0u);
          } else {
            for (uint32_t j = b0; j < b1; j++) {
              const uint64_t kk = skey[j];
              lt[u] += kk < key[u] ? 1u : 0u;
              if (kWantOrder) eqb[u] += (kk == key[u] && (cj < ci || (own && j < slot[u]))) ? 1u : 0u;
            }
          }
        }
        __syncthreads();  // bins / skey are rebuilt by the next chunk (or item)
      }
#pragma unroll
      for (int u = 0; u < kSrSlowPer; u++)
        if ((okm >> u) & 1u) {
          const uint64_t g = (uint64_t)it.start + (uint64_t)ci * C + tid + u * kSrSlowThreads;
          sr_emit<PB, SINK>(F, it, g, key[u], sr_pay4<PB, SINK>(F, g), sr_pay8<PB, SINK>(F, g), lt[u], eqb[u], acc);
        }
    }
  }
  if (SINK == kSrSums) sr_store_sums<kSrSlowThreads>(acc, F.partials + blockIdx.x);
}

// ---- host side ----------------------------------------------------------------------------------------------------
uint32_t env_u32(const char *name, uint32_t dflt, uint32_t lo, uint32_t hi) {
  const char *e = getenv(name);
  if (!e || !*e) return dflt;
  const long long v = atoll(e);
  if (v < (long long)lo) return lo;
  if (v > (long long)hi) return hi;
  return (uint32_t)v;
}

struct SrShape {
  int levels = 0;
  uint32_t f[3] = {1, 1, 1};
  uint32_t every = 1;
  uint64_t buckets = 1, ns = 0, sample_stride = 1;
  uint32_t nparts[3] = {1, 1, 1}, nb[3] = {1, 1, 1}, tile_cap[3] = {1, 1, 1};
  uint32_t per_part = kSrTile;
  uint64_t max_small = 1, max_large = 1;
};

SrShape sr_shape(uint64_t n, const SrTuning &t) {
  SrShape s;
  if (n <= t.cap) return s;
  const uint64_t max_f = (uint64_t)t.max_split + 1;
  uint64_t b = (n + t.target - 1) / t.target;
  if (b < 2) b = 2;
  s.levels = b <= max_f ? 1 : (b <= max_f * max_f ? 2 : 3);
  uint64_t f = (uint64_t)std::ceil(std::pow((double)b, 1.0 / s.levels));
  while (s.levels == 2 && f * f < b) f++;
  while (s.levels == 3 && f * f * f < b) f++;
  f = std::min(std::max<uint64_t>(f, 2), max_f);
  uint64_t below = 1;
  for (int l = 0; l + 1 < s.levels; l++) {
    s.f[l] = (uint32_t)f;
    below *= f;
  }
  uint64_t last = (b + below - 1) / below;
  last = std::min(std::max<uint64_t>(last, 2), max_f);
  s.f[s.levels - 1] = (uint32_t)last;
  s.buckets = below * last;
  // (the sample is sorted by the same machinery: it has to be a fraction of the job)
  uint64_t every = std::min<uint64_t>(t.oversample, n / (4 * s.buckets));
  if (every < 1) every = 1;
  s.every = (uint32_t)every;
  s.ns = s.buckets * every;  // <= n when buckets <= n
  if (s.ns > n) s.ns = n;    // (cannot happen: buckets <= ceil(n / target) * (rounding of three roots); kept honest)
  s.sample_stride = std::max<uint64_t>(1, n / s.ns);
  const uint64_t first_parts = t.first_parts;
  uint64_t per = (n + first_parts - 1) / first_parts;
  per = (per + kSrTile - 1) / kSrTile * kSrTile;
  s.per_part = (uint32_t)std::min<uint64_t>(per, 0x80000000ull);
  uint64_t parts = first_parts;
  for (int l = 0; l < s.levels; l++) {
    s.nparts[l] = (uint32_t)parts;
    s.nb[l] = 2 * (s.f[l] - 1) + 1;
    // (a part of pass 1 lies in as many pieces as pass 0 has stretches, each with a last, short tile)
    const uint64_t pieces = std::min<uint64_t>(parts, n) * (l == 1 ? first_parts : 1);
    s.tile_cap[l] = (uint32_t)std::min<uint64_t>(n / kSrTile + pieces + 16, 0xFFFFFFF0ull);
    parts = (l == 0 ? 1 : parts) * s.nb[l];
  }
  s.max_small = parts + n / kSrEqPiece + 2;  // (`parts` is now the number of buckets of the last pass)
  s.max_large = n / (t.cap + 1) + 4;  // (+ the two buckets at the ends of the key range)
  return s;
}

struct Carver {  // hands out pieces of the workspace; with base == nullptr it only adds up
  char *base;
  size_t off = 0;
  template <class T>
  T *take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T *p = base ? (T *)(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

int sr_small_grid() { return std::max(1, tgx_num_cus()) * 6; }
int sr_large_grid() { return std::max(1, tgx_num_cus()); }

template <int PB, int SINK>
void launch_rank(const SrFinal &small, const SrFinal &large, hipStream_t stream) {
  hipLaunchKernelGGL((sr_rank_small_kernel<PB, SINK>), dim3(sr_small_grid()), dim3(kSrFastThreads), 0, stream, small);
  hipLaunchKernelGGL((sr_rank_large_kernel<PB, SINK>), dim3(sr_large_grid()), dim3(kSrSlowThreads), 0, stream, large);
}

// queues the whole job; `dry`: only walks the workspace (sizes)
hipError_t sr_run_impl(const SrJob &job, Carver &ws, hipStream_t stream, const SrTuning &tune, bool dry, SrPlaced *placed,
                       int depth) {
  const uint64_t n = job.n;
  if (placed) *placed = SrPlaced{job.keys, job.pay};
  if (n == 0) return hipSuccess;
  if (depth > 40) return hipErrorUnknown;
  const SrShape sh = sr_shape(n, tune);
  SrItem *small = ws.take<SrItem>(sh.max_small), *large = ws.take<SrItem>(sh.max_large);
  SrItem *tiny = ws.take<SrItem>(sh.max_small);
  uint32_t *counters = ws.take<uint32_t>(8);  // [0] small items, [1] large items, [2] tiny items
  uint32_t *own_status = ws.take<uint32_t>(1);
  uint32_t *status = job.status ? job.status : own_status;
  if (!dry && depth == 0) {  // (the sorts of the samples share the job's word: a failure anywhere is the job's)
    hipError_t e0 = hipMemsetAsync(status, 0, sizeof(uint32_t), stream);
    if (e0 != hipSuccess) return e0;
  }
  // bucket sizes from the sample in all passes but the last, if the caller gave the room
  const uint64_t roomy = sr_roomy_elems(n);
  const bool optimistic = job.optimistic && tune.optimistic && sh.levels >= 2 && job.cap[0] >= roomy &&
                          job.cap[1] >= roomy && roomy <= 0xFFFFFFF0ull;
  // (column values as keys: only pass 0 reads them, and only a job that never writes where it reads may be given any)
  if ((job.key_source != kSrKeys || job.pay_source != kSrKeys) && !optimistic) return hipErrorInvalidValue;
  const uint64_t *keys = job.keys;
  const void *pay = job.pay;
  if (sh.levels == 0) {
    if (!dry)
      hipLaunchKernelGGL(sr_one_item_kernel, dim3(1), dim3(1), 0, stream, large, counters, (uint32_t)n);
  } else {
    // ---- splitters: a sorted sample
    uint64_t *samp = ws.take<uint64_t>(sh.ns), *sorted = ws.take<uint64_t>(sh.ns);
    uint64_t *fine = ws.take<uint64_t>(sh.buckets);
    SrJob sj;
    sj.keys = samp;
    sj.n = sh.ns;
    sj.pay_bytes = 0;
    sj.k[0] = ws.take<uint64_t>(sh.ns);
    sj.k[1] = ws.take<uint64_t>(sh.ns);
    sj.sink = kSrSorted;
    sj.out_keys = sorted;
    sj.status = status;
    if (!dry)
      hipLaunchKernelGGL(sr_sample_kernel, dim3((unsigned)((sh.ns + 255) / 256)), dim3(256), 0, stream, job.keys, n, sh.ns,
                         samp, job.key_source);
    hipError_t e = sr_run_impl(sj, ws, stream, tune, dry, nullptr, depth + 1);
    if (e != hipSuccess) return e;
    if (!dry)
      hipLaunchKernelGGL(sr_pick_kernel, dim3((unsigned)((sh.buckets - 1 + 255) / 256)), dim3(256), 0, stream, sorted,
                         sh.buckets - 1, sh.every, fine);
    // ---- the partition passes
    const int per_xcd = std::max(1, tgx_num_cus() / kSrXcds);
    const int grid = kSrXcds * per_xcd * (int)tune.wg_per_cu;
    const int count_grid = kSrXcds * per_xcd * 4;  // (little LDS, few registers: four 512-thread workgroups fill a CU)
    SrLevel L{}, prev{};
    for (int lv = 0; lv < sh.levels; lv++) {
      prev = L;
      const bool guess = optimistic && lv + 1 < sh.levels;  // this pass takes its buckets' room from the sample
      L.keys_in = keys;
      L.pay_in = pay;
      L.key_source = lv == 0 ? job.key_source : (int)kSrKeys;
      L.pay_source = lv == 0 ? job.pay_source : (int)kSrKeys;
      L.keys_out = job.k[lv & 1];
      L.pay_out = job.p[lv & 1];
      L.out_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(job.cap[lv & 1], n), 0xFFFFFFF0ull);
      L.status = status;
      L.fine = fine;
      L.level = lv;
      L.nbp = lv == 2 ? sh.nb[1] : 1;
      // splitter k of a part: fine[first + (k + 1) * stride - 1], stride = the ways of the passes below
      uint32_t below[3] = {1, 1, 1};
      for (int l = sh.levels - 2; l >= 0; l--) below[l] = below[l + 1] * sh.f[l + 1];
      L.stride = below[lv];
      L.w1 = below[0];
      L.w2 = sh.levels > 1 ? below[1] : 1;
      L.split_count = sh.f[lv] - 1;
      L.nb = sh.nb[lv];
      L.nparts = sh.nparts[lv];
      L.tile_cap = sh.tile_cap[lv];
      // long parts (the first passes): all workgroups of an XCD walk a part side by side -- one part's buckets open per
      // L2 (a workgroup on a part of its own: 96 x 199 x 2 open lines per L2, and the last pass of three took 10.3
      // instead of 6.1 ms); short parts (that pass: ~50 tiles each): eight together, or every tile would bring its own
      // splitters and table
      {
        const uint64_t tiles_per_part = (n / kSrTile) / std::max<uint64_t>(1, L.nparts / 2 + 1);
        L.group = tiles_per_part < 4ull * (uint64_t)(grid / kSrXcds) ? 8u : 0u;
      }
      L.tiles = ws.take<SrTileRef>((size_t)kSrXcds * L.tile_cap);
      L.tile_count = ws.take<uint32_t>(kSrXcds);
      const size_t table = (size_t)L.nparts * L.nb;
      L.tot = ws.take<uint32_t>(table);
      L.cursor = ws.take<uint32_t>(table);
      L.limit = ws.take<uint32_t>(table);
      L.bstart = ws.take<uint32_t>(table + 2);
      L.part_size = ws.take<uint32_t>(L.nparts);
      L.part_total = ws.take<uint32_t>(L.nparts);
      L.out_base = ws.take<uint32_t>((size_t)L.nparts + 1);
      L.pbegin = lv == 0 ? ws.take<uint32_t>(table) : nullptr;
      uint32_t *sample_start = nullptr;
      if (lv == 0) {
        uint32_t *part_start = ws.take<uint32_t>((size_t)L.nparts + 2);
        sample_start = ws.take<uint32_t>((size_t)L.nparts + 2);
        L.pbeg = part_start;
        L.pend = part_start + 1;
        L.npieces = 1;
        L.pstride = 0;
        if (!dry)
          hipLaunchKernelGGL(sr_first_parts_kernel, dim3(1), dim3(128), 0, stream, part_start, L.nparts, sh.per_part,
                             (uint32_t)n, sample_start, (uint32_t)sh.ns);
      } else if (lv == 1) {  // what each stretch of pass 0 wrote of a bucket
        L.pbeg = prev.pbegin;
        L.pend = prev.cursor;
        L.npieces = prev.nparts;
        L.pstride = prev.nb;
      } else {
        L.pbeg = prev.bstart;
        L.pend = prev.cursor;
        L.npieces = 1;
        L.pstride = 0;
      }
      // (pass 0 with room from the sample: the counting kernel runs over the sample's stretches)
      SrLevel LS = L;
      uint32_t *sample_size = nullptr;
      if (lv == 0) {
        LS.keys_in = samp;
        LS.key_source = kSrKeys;  // (the sample holds sort keys)
        LS.pbeg = sample_start;
        LS.pend = sample_start + 1;
        LS.tile_cap = (uint32_t)(sh.ns / kSrTile + L.nparts + 16);
        LS.tiles = ws.take<SrTileRef>((size_t)kSrXcds * LS.tile_cap);
        LS.tile_count = ws.take<uint32_t>(kSrXcds);
        sample_size = ws.take<uint32_t>(L.nparts);
        LS.part_size = sample_size;
      }
      if (!dry) {
        hipError_t e2 = hipMemsetAsync(L.tot, 0, table * sizeof(uint32_t), stream);
        if (e2 == hipSuccess) e2 = hipMemsetAsync(L.tile_count, 0, kSrXcds * sizeof(uint32_t), stream);
        if (e2 == hipSuccess) e2 = hipMemsetAsync(L.part_size, 0, L.nparts * sizeof(uint32_t), stream);
        if (e2 != hipSuccess) return e2;
        const unsigned tile_waves = L.nparts * L.npieces;
        hipLaunchKernelGGL(sr_tiles_kernel, dim3((tile_waves + 3) / 4), dim3(256), 0, stream, L);
        if (!guess) {
          hipLaunchKernelGGL(sr_count_kernel, dim3(count_grid), dim3(kSrPartThreads), 0, stream, L);
        } else if (lv == 0) {
          e2 = hipMemsetAsync(LS.tile_count, 0, kSrXcds * sizeof(uint32_t), stream);
          if (e2 == hipSuccess) e2 = hipMemsetAsync(sample_size, 0, L.nparts * sizeof(uint32_t), stream);
          if (e2 != hipSuccess) return e2;
          hipLaunchKernelGGL(sr_tiles_kernel, dim3((L.nparts + 3) / 4), dim3(256), 0, stream, LS);
          hipLaunchKernelGGL(sr_count_kernel, dim3(count_grid), dim3(kSrPartThreads), 0, stream, LS);
          hipLaunchKernelGGL(sr_room_first_kernel, dim3((unsigned)((table + 255) / 256)), dim3(256), 0, stream, L,
                             sample_start, tune.sigmas_x2);
        } else {
          hipLaunchKernelGGL(sr_room_kernel, dim3((unsigned)((table + 255) / 256)), dim3(256), 0, stream, L, sorted,
                             (uint32_t)sh.ns, sh.buckets - 1, tune.sigmas_x2);
        }
        if (lv == 0) {
          hipLaunchKernelGGL(sr_offsets_first_kernel, dim3(1), dim3(512), 0, stream, L);
        } else {
          hipLaunchKernelGGL(sr_part_totals_kernel, dim3((L.nparts + 7) / 8), dim3(512), 0, stream, L);
          hipLaunchKernelGGL(sr_scan_parts_kernel, dim3(1), dim3(1024), 0, stream, L);
          hipLaunchKernelGGL(sr_offsets_kernel, dim3((L.nparts + 7) / 8), dim3(512), 0, stream, L);
        }
        // (an instance per kind of source: see sr_scatter_kernel)
#define TGX_SR_SCATTER(PB, KS, PS) \
  hipLaunchKernelGGL((sr_scatter_kernel<PB, KS, PS>), dim3(grid), dim3(kSrPartThreads), 0, stream, L)
        const int ks = L.key_source, ps = job.pay_bytes == 8 ? L.pay_source : (int)kSrKeys;
        if (job.pay_bytes == 8) {
          if (ks == kSrKeys && ps == kSrKeys) TGX_SR_SCATTER(8, kSrKeys, kSrKeys);
          else if (ks == kSrFloat64 && ps == kSrFloat64) TGX_SR_SCATTER(8, kSrFloat64, kSrFloat64);
          else if (ks == kSrFloat64 && ps == kSrInt64) TGX_SR_SCATTER(8, kSrFloat64, kSrInt64);
          else if (ks == kSrInt64 && ps == kSrFloat64) TGX_SR_SCATTER(8, kSrInt64, kSrFloat64);
          else if (ks == kSrInt64 && ps == kSrInt64) TGX_SR_SCATTER(8, kSrInt64, kSrInt64);
          else return hipErrorInvalidValue;  // (keys that are sort keys beside payloads that are not: nobody asks)
        } else if (job.pay_bytes == 4) {
          if (ks == kSrKeys) TGX_SR_SCATTER(4, kSrKeys, kSrKeys);
          else if (ks == kSrFloat64) TGX_SR_SCATTER(4, kSrFloat64, kSrKeys);
          else TGX_SR_SCATTER(4, kSrInt64, kSrKeys);
        } else {
          if (ks == kSrKeys) TGX_SR_SCATTER(0, kSrKeys, kSrKeys);
          else if (ks == kSrFloat64) TGX_SR_SCATTER(0, kSrFloat64, kSrKeys);
          else TGX_SR_SCATTER(0, kSrInt64, kSrKeys);
        }
#undef TGX_SR_SCATTER
      }
      keys = L.keys_out;
      pay = L.pay_out;
    }
    if (!dry) {
      hipError_t e2 = hipMemsetAsync(counters, 0, 8 * sizeof(uint32_t), stream);
      if (e2 != hipSuccess) return e2;
      const uint64_t entries = (uint64_t)(L.level == 0 ? 1 : L.nparts) * L.nb;
      hipLaunchKernelGGL(sr_items_kernel, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, stream, L, sh.buckets,
                         tune.cap, small, counters, large, counters + 1, tiny, counters + 2);
    }
  }
  if (placed) *placed = SrPlaced{keys, pay};
  if (dry) return hipSuccess;
  SrFinal F;
  F.keys = keys;
  F.pay = pay;
  F.items = small;
  F.n_items = counters;
  F.tiny = tiny;
  F.n_tiny = counters + 2;
  F.out_keys = job.out_keys;
  F.out_pay = job.out_pay;
  F.rank32 = job.rank32;
  F.rank_out = job.rank_out;
  F.ext_base = job.ext_base;
  F.partials = job.partials;
  F.cap = tune.slow_cap;
  F.status = status;
  SrFinal FL = F;
  FL.items = large;
  FL.n_items = counters + 1;
  FL.partials = job.partials ? job.partials + sr_small_grid() : nullptr;
  switch (job.sink) {
    case kSrSorted:
      if (job.pay_bytes == 8)
        launch_rank<8, kSrSorted>(F, FL, stream);
      else if (job.pay_bytes == 4)
        launch_rank<4, kSrSorted>(F, FL, stream);
      else
        launch_rank<0, kSrSorted>(F, FL, stream);
      break;
    case kSrRank32:
      launch_rank<0, kSrRank32>(F, FL, stream);
      break;
    case kSrSums:
      launch_rank<4, kSrSums>(F, FL, stream);
      break;
    default:
      launch_rank<4, kSrRankScatter>(F, FL, stream);
      break;
  }
  if (depth == 0 && getenv("TGX_SORT_DEBUG")) {  // (diagnostics: waits for the job)
    uint32_t c[3] = {0, 0, 0};
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(c, counters, sizeof(c), hipMemcpyDeviceToHost);
    uint32_t stw = 0;
    (void)hipMemcpy(&stw, status, sizeof(stw), hipMemcpyDeviceToHost);
    fprintf(stderr, "tgx sort: n %llu, %d passes of %u x %u x %u ways (%s), %llu splitters from %llu sample keys, %u small "
                    "+ %u large + %u tiny items, status %u\n", (unsigned long long)n, sh.levels, sh.f[0], sh.f[1], sh.f[2],
            optimistic ? "room from the sample" : "counted", (unsigned long long)sh.buckets - 1,
            (unsigned long long)sh.ns, c[0], c[1], c[2], stw);
  }
  return hipGetLastError();
}

}  // namespace

SrTuning sr_tuning() {
  SrTuning t;
  t.target = env_u32("TGX_SORT_TARGET", 1024, 2, 1u << 20);
  t.oversample = env_u32("TGX_SORT_SAMPLE", 16, 1, 256);
  t.cap = env_u32("TGX_SORT_CAP", kSrFastCap, 8, kSrFastCap);
  t.slow_cap = env_u32("TGX_SORT_SLOWCAP", kSrSlowCap, 8, kSrSlowCap);
  t.max_split = env_u32("TGX_SORT_SPLIT", kSrMaxSplit, 1, kSrMaxSplit);
  t.wg_per_cu = env_u32("TGX_SORT_WG", 3, 1, 16);
  t.first_parts = env_u32("TGX_SORT_PARTS", kSrFirstParts, 8, 64) & ~7u;
  t.optimistic = env_u32("TGX_SORT_OPTIMISTIC", 1, 0, 1);
  t.sigmas_x2 = env_u32("TGX_SORT_SIGMAS_X2", 13, 0, 64);
  t.optimistic_min = env_u32("TGX_SORT_OPTIMISTIC_MIN", 1u << 20, 1, 0xFFFFFFFFu);
  return t;
}

int sr_partials_count() { return sr_small_grid() + sr_large_grid(); }

// the buckets of a pass that takes their room from the sample lie apart: about a quarter more than the keys
uint64_t sr_roomy_elems(uint64_t n) { return n + n / 3 + (8u << 20); }
bool sr_optimistic_applies(uint64_t n) {
  const SrTuning t = sr_tuning();
  return t.optimistic && n <= 0xFFFFFFF0ull && sr_roomy_elems(n) <= 0xFFFFFFF0ull && sr_shape(n, t).levels >= 2;
}

size_t sr_workspace_bytes(uint64_t n) {
  Carver c{nullptr};
  SrJob j;
  j.n = n;
  (void)sr_run_impl(j, c, nullptr, sr_tuning(), true, nullptr, 0);
  return c.off + 256;
}

hipError_t sr_run(const SrJob &job, void *workspace, size_t workspace_bytes, hipStream_t stream, SrPlaced *placed) {
  if (job.n > 0xFFFFFFF0ull) return hipErrorInvalidValue;
  const SrTuning tune = sr_tuning();
  Carver dry{nullptr};
  hipError_t e = sr_run_impl(job, dry, stream, tune, true, nullptr, 0);
  if (e != hipSuccess) return e;
  if (dry.off + 256 > workspace_bytes) return hipErrorOutOfMemory;
  Carver c{(char *)workspace};
  return sr_run_impl(job, c, stream, tune, false, placed, 0);
}

}  // namespace tgx
