/*
 * suite_mt.c -- CPU ORACLE, TEST INFRASTRUCTURE ONLY (see tgx_oracle.h): the "null + range + unique" suite of
 * BASELINE.json run the way the reference's engine runs it on a multi-core host, for bench.py's `cpu_baseline`.
 *
 * The reference evaluates one SQL aggregate query per constraint (TG/core/suite.rs:67-100) and DataFusion runs each
 * with `target_partitions` = the host's cores: every partition folds its row range into a partial state, the
 * partials are merged (`AnalyzerState::merge`, TG/analyzers/traits.rs:160-170).  COUNT(DISTINCT c)
 * (TG/constraints/uniqueness.rs:612-617) is a per-partition hash set of the non-NULL values, re-partitioned by hash
 * and counted -- a hash set, not a sort.  This file restates exactly that shape over the oracle's scalar kernels:
 *   - T threads, contiguous 64-row aligned row ranges (the shards bench.py's GPU ranks get);
 *   - per column: orc_count (completeness) and orc_stats_* (min / max / mean) on the range, merged at the end;
 *   - per unique column: phase 1 inserts the range's keys into a thread-local open-addressing set, then groups the set's
 *     keys by owner(key); phase 2 gives thread t the runs addressed to it from every partition and counts its
 *     disjoint part.
 * The merged results equal the single-threaded oracle's (tests/test_oracle_golden.py checks it).
 */
#define _GNU_SOURCE /* pthread_barrier_t under -std=c11 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "tgx_oracle.h"

typedef struct {
  uint64_t *slots; /* open addressing, linear probing; EMPTY = all ones, which is tracked by `has_ones` */
  uint64_t mask;
  int has_ones;
  int64_t used;
} keyset_t;

#define KS_EMPTY (~(uint64_t)0)

static inline uint64_t mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}

static inline int32_t owner_of(uint64_t key, int32_t T) {
  return (int32_t)((mix64(key ^ 0x9e3779b97f4a7c15ULL) >> 32) % (uint64_t)T);
}

static int ks_init(keyset_t *s, int64_t expected) {
  uint64_t cap = 1024;
  while (cap < (uint64_t)expected * 2) cap <<= 1;
  s->slots = (uint64_t *)malloc(cap * sizeof(uint64_t));
  if (!s->slots) return -1;
  memset(s->slots, 0xFF, cap * sizeof(uint64_t));
  s->mask = cap - 1;
  s->has_ones = 0;
  s->used = 0;
  return 0;
}

static inline void ks_insert(keyset_t *s, uint64_t key) {
  if (key == KS_EMPTY) {
    s->used += !s->has_ones;
    s->has_ones = 1;
    return;
  }
  uint64_t i = mix64(key) & s->mask;
  for (;;) {
    const uint64_t cur = s->slots[i];
    if (cur == key) return;
    if (cur == KS_EMPTY) {
      s->slots[i] = key;
      s->used++;
      return;
    }
    i = (i + 1) & s->mask;
  }
}

typedef struct {
  const orc_suite_column_t *cols;
  int32_t n_cols;
  const int32_t *unique_cols;
  int32_t n_unique;
  int64_t n;
  int32_t n_threads;
  /* per thread x column partials */
  orc_count_t *counts;
  orc_stats_t *stats;
  keyset_t *local;      /* [thread][unique] */
  int64_t *owned;       /* [thread][unique]: size of the disjoint part thread t counted */
  int64_t *non_null_u;  /* [thread][unique] */
  uint64_t **out_keys;  /* [thread][unique]: the partition's distinct keys grouped by owner */
  int64_t *out_start;   /* [thread][unique][T + 1]: start of each owner's run in out_keys */
  pthread_barrier_t barrier;
  int failed;
} suite_job_t;

typedef struct {
  suite_job_t *job;
  int32_t tid;
} worker_t;

static void range_of(int64_t n, int32_t threads, int32_t t, int64_t *lo, int64_t *hi) {
  const int64_t per = (n / threads) / 64 * 64;
  *lo = (int64_t)t * per;
  *hi = t == threads - 1 ? n : *lo + per;
}

static void *worker(void *arg) {
  worker_t *w = (worker_t *)arg;
  suite_job_t *j = w->job;
  const int32_t t = w->tid, T = j->n_threads;
  int64_t lo, hi;
  range_of(j->n, T, t, &lo, &hi);
  for (int32_t c = 0; c < j->n_cols; c++) {
    const orc_suite_column_t *col = &j->cols[c];
    /* one scan per constraint, as the reference does: completeness, then the statistics */
    orc_count(col->validity, lo, hi - lo, &j->counts[(size_t)t * j->n_cols + c]);
    orc_stats_t *st = &j->stats[(size_t)t * j->n_cols + c];
    if (col->is_float)
      orc_stats_f64((const double *)col->values, col->validity, lo, hi - lo, st);
    else
      orc_stats_i64((const int64_t *)col->values, col->validity, lo, hi - lo, st);
  }
  /* COUNT(DISTINCT): partial hash set of this partition */
  for (int32_t u = 0; u < j->n_unique; u++) {
    const orc_suite_column_t *col = &j->cols[j->unique_cols[u]];
    keyset_t *ks = &j->local[(size_t)t * j->n_unique + u];
    if (ks_init(ks, hi - lo) != 0) {
      j->failed = 1;
      continue;
    }
    const uint64_t *bits = (const uint64_t *)col->values;
    int64_t nn = 0;
    for (int64_t i = lo; i < hi; i++) {
      if (col->validity && !((col->validity[i >> 3] >> (i & 7)) & 1)) continue;
      ks_insert(ks, bits[i]);
      nn++;
    }
    j->non_null_u[(size_t)t * j->n_unique + u] = nn;
    /* re-partition by owner (DataFusion's hash repartition between the partial and the final aggregate): this
     * partition's distinct keys, grouped by the thread that will count them */
    int64_t *start = &j->out_start[((size_t)t * j->n_unique + u) * (size_t)(T + 1)];
    memset(start, 0, (size_t)(T + 1) * sizeof(int64_t));
    for (uint64_t i = 0; i <= ks->mask; i++)
      if (ks->slots[i] != KS_EMPTY) start[owner_of(ks->slots[i], T) + 1]++;
    if (ks->has_ones) start[owner_of(KS_EMPTY, T) + 1]++;
    for (int32_t o = 0; o < T; o++) start[o + 1] += start[o];
    uint64_t *out = (uint64_t *)malloc((size_t)(ks->used > 0 ? ks->used : 1) * sizeof(uint64_t));
    int64_t *cur = (int64_t *)malloc((size_t)T * sizeof(int64_t));
    if (!out || !cur) {
      j->failed = 1;
      free(out);
      free(cur);
      continue;
    }
    memcpy(cur, start, (size_t)T * sizeof(int64_t));
    for (uint64_t i = 0; i <= ks->mask; i++)
      if (ks->slots[i] != KS_EMPTY) out[cur[owner_of(ks->slots[i], T)]++] = ks->slots[i];
    if (ks->has_ones) out[cur[owner_of(KS_EMPTY, T)]++] = KS_EMPTY;
    free(cur);
    j->out_keys[(size_t)t * j->n_unique + u] = out;
    free(ks->slots);
    ks->slots = NULL;
  }
  pthread_barrier_wait(&j->barrier);
  /* final aggregate: thread t unites the runs addressed to it */
  for (int32_t u = 0; u < j->n_unique && !j->failed; u++) {
    int64_t expect = 0;
    for (int32_t s = 0; s < T; s++) {
      const int64_t *start = &j->out_start[((size_t)s * j->n_unique + u) * (size_t)(T + 1)];
      expect += start[t + 1] - start[t];
    }
    keyset_t fin;
    if (ks_init(&fin, expect) != 0) {
      j->failed = 1;
      break;
    }
    for (int32_t s = 0; s < T; s++) {
      const int64_t *start = &j->out_start[((size_t)s * j->n_unique + u) * (size_t)(T + 1)];
      const uint64_t *keys = j->out_keys[(size_t)s * j->n_unique + u];
      for (int64_t i = start[t]; i < start[t + 1]; i++) ks_insert(&fin, keys[i]);
    }
    j->owned[(size_t)t * j->n_unique + u] = fin.used;
    free(fin.slots);
  }
  return NULL;
}

int orc_suite_mt(const orc_suite_column_t *cols, int32_t n_cols, const int32_t *unique_cols, int32_t n_unique,
                 int64_t n, int32_t n_threads, orc_count_t *counts, orc_stats_t *stats, orc_distinct_t *distinct) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 1024) n_threads = 1024;
  suite_job_t j;
  memset(&j, 0, sizeof(j));
  j.cols = cols;
  j.n_cols = n_cols;
  j.unique_cols = unique_cols;
  j.n_unique = n_unique;
  j.n = n;
  j.n_threads = n_threads;
  j.counts = (orc_count_t *)calloc((size_t)n_threads * (size_t)(n_cols > 0 ? n_cols : 1), sizeof(orc_count_t));
  j.stats = (orc_stats_t *)calloc((size_t)n_threads * (size_t)(n_cols > 0 ? n_cols : 1), sizeof(orc_stats_t));
  const size_t nu = (size_t)n_threads * (size_t)(n_unique > 0 ? n_unique : 1);
  j.local = (keyset_t *)calloc(nu, sizeof(keyset_t));
  j.owned = (int64_t *)calloc(nu, sizeof(int64_t));
  j.non_null_u = (int64_t *)calloc(nu, sizeof(int64_t));
  j.out_keys = (uint64_t **)calloc(nu, sizeof(uint64_t *));
  j.out_start = (int64_t *)calloc(nu * (size_t)(n_threads + 1), sizeof(int64_t));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  worker_t *ws = (worker_t *)calloc((size_t)n_threads, sizeof(worker_t));
  if (!j.counts || !j.stats || !j.local || !j.owned || !j.non_null_u || !j.out_keys || !j.out_start || !th || !ws)
    return -1;
  pthread_barrier_init(&j.barrier, NULL, (unsigned)n_threads);
  int started = 0;
  for (int32_t t = 0; t < n_threads; t++) {
    ws[t].job = &j;
    ws[t].tid = t;
    if (pthread_create(&th[t], NULL, worker, &ws[t]) != 0) break;
    started++;
  }
  if (started != n_threads) return -2; /* (cannot happen below the process limit; the barrier would hang) */
  for (int32_t t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  pthread_barrier_destroy(&j.barrier);
  /* merge the partials in partition order (AnalyzerState::merge) */
  for (int32_t c = 0; c < n_cols; c++) {
    orc_count_t cnt = {0, 0};
    orc_stats_t m;
    memset(&m, 0, sizeof(m));
    long double sum = 0.0L;
    for (int32_t t = 0; t < n_threads; t++) {
      const orc_count_t *pc = &j.counts[(size_t)t * n_cols + c];
      const orc_stats_t *ps = &j.stats[(size_t)t * n_cols + c];
      cnt.total += pc->total;
      cnt.non_null += pc->non_null;
      m.total += ps->total;
      m.is_float = ps->is_float;
      if (!ps->has_value) continue;
      if (!m.has_value) {
        m.min_i = ps->min_i;
        m.max_i = ps->max_i;
        m.min_f = ps->min_f;
        m.max_f = ps->max_f;
        m.has_value = 1;
      } else {
        if (ps->min_i < m.min_i) m.min_i = ps->min_i;
        if (ps->max_i > m.max_i) m.max_i = ps->max_i;
        if (ps->min_f < m.min_f) m.min_f = ps->min_f; /* (no NaN / signed zeros in the bench table) */
        if (ps->max_f > m.max_f) m.max_f = ps->max_f;
      }
      m.non_null += ps->non_null;
      m.sum_i_wrapping = (int64_t)((uint64_t)m.sum_i_wrapping + (uint64_t)ps->sum_i_wrapping);
      sum += (long double)ps->sum_hi;
    }
    m.sum_f = m.sum_hi = (double)sum;
    m.mean = m.non_null > 0 ? (double)(sum / (long double)m.non_null) : 0.0;
    counts[c] = cnt;
    stats[c] = m;
  }
  for (int32_t u = 0; u < n_unique; u++) {
    orc_distinct_t d;
    memset(&d, 0, sizeof(d));
    d.total = n;
    for (int32_t t = 0; t < n_threads; t++) {
      d.non_null += j.non_null_u[(size_t)t * n_unique + u];
      d.distinct += j.owned[(size_t)t * n_unique + u];
      free(j.local[(size_t)t * n_unique + u].slots);
      free(j.out_keys[(size_t)t * n_unique + u]);
    }
    distinct[u] = d; /* groups_once is not part of FullUniqueness (uniqueness.rs:612-617): left 0 */
  }
  const int failed = j.failed;
  free(j.counts);
  free(j.stats);
  free(j.local);
  free(j.owned);
  free(j.non_null_u);
  free(j.out_keys);
  free(j.out_start);
  free(th);
  free(ws);
  return failed ? -1 : 0;
}
