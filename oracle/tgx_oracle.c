/*
 * tgx_oracle.c -- CPU ORACLE. TEST INFRASTRUCTURE ONLY (see tgx_oracle.h).
 *
 * Scalar restatement of the aggregates term-guard's constraints emit as SQL and
 * of its hand-written KllSketch.  Third-party arithmetic (DataFusion 50.3.0 /
 * arrow 56.2.0, Cargo.lock:127-128, 998-999) is restated from its published
 * algorithms; each function names the reference call site it serves.
 */
#include "tgx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int bit_is_set(const uint8_t *bm, int64_t i) {
  return bm == NULL ? 1 : (bm[i >> 3] >> (i & 7)) & 1;
}

/* ------------------------------------------------------------------ K1 */
/* TG/constraints/completeness.rs:158-163: SELECT COUNT(*), COUNT(col).
 * DataFusion's count = len - null_count; we count bit by bit. */
void orc_count(const uint8_t *validity, int64_t offset, int64_t n, orc_count_t *out) {
  int64_t nn = 0;
  for (int64_t i = 0; i < n; i++) nn += bit_is_set(validity, offset + i);
  out->total = n;
  out->non_null = nn;
}

/* ------------------------------------------------------------------ K2-K4 */
/* IEEE-754 totalOrder on doubles, as f64::total_cmp (arrow-arith aggregate min/max
 * route floats through ArrowNativeTypeOp::is_lt/is_gt = total_cmp). */
static inline int64_t f64_total_key(double d) {
  int64_t b;
  memcpy(&b, &d, 8);
  return b ^ (int64_t)(((uint64_t)(b >> 63)) >> 1);
}

typedef struct {
  /* DataFusion VarianceAccumulator (Welford) */
  uint64_t count;
  double mean, m2;
} welford_t;

static inline void welford_add(welford_t *w, double v) {
  w->count += 1;
  double delta1 = v - w->mean;
  double new_mean = delta1 / (double)w->count + w->mean;
  double delta2 = v - new_mean;
  w->m2 += delta1 * delta2;
  w->mean = new_mean;
}

static void stats_finish(orc_stats_t *o, const welford_t *w, long double sum_ld) {
  o->has_value = o->non_null > 0;
  o->sum_hi = (double)sum_ld;
  o->mean = o->non_null > 0 ? o->sum_f / (double)o->non_null : NAN;
  o->has_variance = w->count >= 2;
  o->var_samp = w->count >= 2 ? w->m2 / (double)(w->count - 1) : NAN;
  o->stddev_samp = w->count >= 2 ? sqrt(o->var_samp) : NAN;
}

/* TG/constraints/statistics.rs:263 `SELECT MIN|MAX|AVG|SUM|STDDEV|VARIANCE("c")`.
 * SUM(Int64) is a wrapping i64 add; AVG(Int64) coerces each value to Float64 first. */
void orc_stats_i64(const int64_t *values, const uint8_t *validity, int64_t offset, int64_t n,
                   orc_stats_t *out) {
  memset(out, 0, sizeof(*out));
  out->total = n;
  out->min_i = INT64_MAX;
  out->max_i = INT64_MIN;
  out->min_f = NAN;
  out->max_f = NAN;
  uint64_t wrap = 0;
  double sf = 0.0, sq = 0.0;
  long double sl = 0.0L;
  welford_t w = {0, 0.0, 0.0};
  for (int64_t i = 0; i < n; i++) {
    if (!bit_is_set(validity, offset + i)) continue;
    int64_t v = values[offset + i];
    out->non_null++;
    if (v < out->min_i) out->min_i = v;
    if (v > out->max_i) out->max_i = v;
    wrap += (uint64_t)v;
    double d = (double)v;
    sf += d;
    sq += d * d;
    sl += (long double)v;
    welford_add(&w, d);
  }
  out->sum_i_wrapping = (int64_t)wrap;
  out->sum_f = sf;
  out->sumsq_f = sq;
  stats_finish(out, &w, sl);
  if (out->has_value) {
    out->min_f = (double)out->min_i;
    out->max_f = (double)out->max_i;
  }
}

void orc_stats_f64(const double *values, const uint8_t *validity, int64_t offset, int64_t n,
                   orc_stats_t *out) {
  memset(out, 0, sizeof(*out));
  out->total = n;
  out->is_float = 1;
  out->min_f = NAN;
  out->max_f = NAN;
  int64_t kmin = INT64_MAX, kmax = INT64_MIN;
  double sf = 0.0, sq = 0.0;
  long double sl = 0.0L;
  welford_t w = {0, 0.0, 0.0};
  for (int64_t i = 0; i < n; i++) {
    if (!bit_is_set(validity, offset + i)) continue;
    double v = values[offset + i];
    out->non_null++;
    int64_t k = f64_total_key(v);
    if (out->non_null == 1 || k < kmin) {
      kmin = k;
      out->min_f = v;
    }
    if (out->non_null == 1 || k > kmax) {
      kmax = k;
      out->max_f = v;
    }
    sf += v;
    sq += v * v;
    sl += (long double)v;
    welford_add(&w, v);
  }
  out->sum_f = sf;
  out->sumsq_f = sq;
  stats_finish(out, &w, sl);
}

/* ------------------------------------------------------------------ K5 / K6 */
static int cmp_u64(const void *a, const void *b) {
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : x > y;
}

/* TG/constraints/uniqueness.rs:612-617 COUNT(DISTINCT c) (NULLs excluded),
 * :671-681 GROUP BY c / SUM(CASE WHEN cnt = 1 ...) where NULL forms one group. */
int orc_distinct_bits64(const uint64_t *bits, const uint8_t *validity, int64_t offset, int64_t n,
                        orc_distinct_t *out) {
  memset(out, 0, sizeof(*out));
  out->total = n;
  uint64_t *tmp = (uint64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(uint64_t));
  if (!tmp) return -1;
  int64_t m = 0;
  for (int64_t i = 0; i < n; i++)
    if (bit_is_set(validity, offset + i)) tmp[m++] = bits[offset + i];
  out->non_null = m;
  qsort(tmp, (size_t)m, sizeof(uint64_t), cmp_u64);
  int64_t i = 0;
  while (i < m) {
    int64_t j = i + 1;
    while (j < m && tmp[j] == tmp[i]) j++;
    out->distinct++;
    if (j - i == 1) out->groups_once++;
    i = j;
  }
  if (n - m == 1) out->groups_once++; /* a NULL group of exactly one row */
  free(tmp);
  return 0;
}

/* TG/constraints/approx_count_distinct.rs:56-66 `APPROX_DISTINCT(col)`: a HyperLogLog sketch.  DataFusion's
 * (datafusion-functions-aggregate 50.3.0, src/hyperloglog.rs -- absent from /root/reference, a Cargo.lock dependency)
 * has 2^14 one-byte registers over ahash values; its `count()` is Ertl's improved estimator ("New cardinality
 * estimation algorithms for HyperLogLog sketches", 2017, algorithm 6).  The hash is third-party and seeded inside the
 * crate; the product's lane mixes a value's 64 bits with the bijection restated here (index = low 14 bits of `a`, rank =
 * leading zeros of `b` + 1, 1..33), so the REGISTERS are a pure function of the set of values: the device must
 * reproduce them byte for byte, and the estimate with them.  NULL rows are skipped (:229-255). */
#define ORC_HLL_REGISTERS 16384
static uint32_t orc_rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
void orc_hll_registers(const uint64_t *bits, const uint8_t *validity, int64_t offset, int64_t n, uint8_t *registers) {
  for (int64_t i = 0; i < n; i++) {
    if (!bit_is_set(validity, offset + i)) continue;
    const uint64_t v = bits[offset + i];
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    uint32_t a = lo ^ orc_rotl32(hi * 0x9E3779B1u, 15);
    a ^= a >> 16;
    a *= 0x85EBCA6Bu;
    a ^= a >> 13;
    a *= 0xC2B2AE35u;
    a ^= a >> 16;
    uint32_t b = (hi ^ orc_rotl32(a, 16)) * 0x27D4EB2Fu;
    uint32_t rank = 1;
    while (rank <= 32 && !(b & 0x80000000u)) {
      b <<= 1;
      rank++;
    }
    uint8_t *r = &registers[a & (ORC_HLL_REGISTERS - 1)];
    if (*r < rank) *r = (uint8_t)rank;
  }
}
static double orc_hll_sigma(double x) {
  if (x == 1.0) return INFINITY;
  double y = 1.0, z = x;
  for (;;) {
    x *= x;
    double z0 = z;
    z += x * y;
    y += y;
    if (z0 == z) return z;
  }
}
static double orc_hll_tau(double x) {
  if (x == 0.0 || x == 1.0) return 0.0;
  double y = 1.0, z = 1.0 - x;
  for (;;) {
    x = sqrt(x);
    double z0 = z;
    y *= 0.5;
    z -= (1.0 - x) * (1.0 - x) * y;
    if (z0 == z) return z / 3.0;
  }
}
/* q = 32: ranks 0 .. 33 (DataFusion: q = 50 with a 64-bit hash; the formula is the same) */
uint64_t orc_hll_estimate(const uint8_t *registers) {
  enum { Q = 32 };
  double hist[Q + 2] = {0};
  for (int i = 0; i < ORC_HLL_REGISTERS; i++) hist[registers[i] > Q + 1 ? Q + 1 : registers[i]] += 1.0;
  const double m = (double)ORC_HLL_REGISTERS;
  double z = m * orc_hll_tau((m - hist[Q + 1]) / m);
  for (int k = Q; k >= 1; k--) z = 0.5 * (z + hist[k]);
  z += m * orc_hll_sigma(hist[0] / m);
  const double e = 0.5 / log(2.0) * m * m / z;
  return isfinite(e) ? (uint64_t)llround(e) : 0;
}

typedef struct {
  const uint8_t *p;
  int32_t len;
} strref_t;
static int cmp_strref(const void *a, const void *b) {
  const strref_t *x = (const strref_t *)a, *y = (const strref_t *)b;
  int32_t m = x->len < y->len ? x->len : y->len;
  int c = m ? memcmp(x->p, y->p, (size_t)m) : 0;
  if (c) return c;
  return x->len < y->len ? -1 : x->len > y->len;
}

int orc_distinct_utf8(const int32_t *offsets, const uint8_t *data, const uint8_t *validity,
                      int64_t offset, int64_t n, orc_distinct_t *out) {
  memset(out, 0, sizeof(*out));
  out->total = n;
  strref_t *tmp = (strref_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(strref_t));
  if (!tmp) return -1;
  int64_t m = 0;
  for (int64_t i = 0; i < n; i++) {
    if (!bit_is_set(validity, offset + i)) continue;
    int32_t a = offsets[offset + i], b = offsets[offset + i + 1];
    tmp[m].p = data + a;
    tmp[m].len = b - a;
    m++;
  }
  out->non_null = m;
  qsort(tmp, (size_t)m, sizeof(strref_t), cmp_strref);
  int64_t i = 0;
  while (i < m) {
    int64_t j = i + 1;
    while (j < m && cmp_strref(&tmp[j], &tmp[i]) == 0) j++;
    out->distinct++;
    if (j - i == 1) out->groups_once++;
    i = j;
  }
  if (n - m == 1) out->groups_once++;
  free(tmp);
  return 0;
}

/* ------------------------------------------------------------------ K10 / K11 / K12 */
static inline double num_at(const void *p, int is_float, int64_t i) {
  return is_float ? ((const double *)p)[i] : (double)((const int64_t *)p)[i];
}

/* TG/analyzers/advanced/correlation.rs:239-249:
 * COUNT(*), SUM(x), SUM(y), SUM(x*x), SUM(y*y), SUM(x*y) WHERE x,y NOT NULL, all CAST AS DOUBLE */
void orc_comoments(const void *x, int x_is_float, const uint8_t *x_validity, int64_t x_offset,
                   const void *y, int y_is_float, const uint8_t *y_validity, int64_t y_offset,
                   int64_t n, orc_comoments_t *out) {
  memset(out, 0, sizeof(*out));
  for (int64_t i = 0; i < n; i++) {
    if (!bit_is_set(x_validity, x_offset + i) || !bit_is_set(y_validity, y_offset + i)) continue;
    double a = num_at(x, x_is_float, x_offset + i), b = num_at(y, y_is_float, y_offset + i);
    out->n++;
    out->sum_x += a;
    out->sum_y += b;
    out->sum_x2 += a * a;
    out->sum_y2 += b * b;
    out->sum_xy += a * b;
  }
}

/* TG/analyzers/advanced/correlation.rs:407-427 */
double orc_pearson_from_state(const orc_comoments_t *s) {
  if (s->n < 2) return NAN;
  double n = (double)s->n;
  double numerator = n * s->sum_xy - s->sum_x * s->sum_y;
  double denominator =
      sqrt((n * s->sum_x2 - s->sum_x * s->sum_x) * (n * s->sum_y2 - s->sum_y * s->sum_y));
  if (denominator == 0.0) return 0.0;
  return numerator / denominator;
}

/* TG/analyzers/advanced/correlation.rs:428-432 */
double orc_covariance_from_state(const orc_comoments_t *s) {
  if (s->n < 2) return NAN;
  double n = (double)s->n;
  return (s->sum_xy - (s->sum_x * s->sum_y) / n) / (n - 1.0);
}

/* TG/constraints/correlation.rs:260-275 CORR(a,b) / COVAR_SAMP(a,b): DataFusion's
 * CovarianceAccumulator (online co-moment) and population StddevAccumulators. */
void orc_corr_online(const void *x, int x_is_float, const uint8_t *x_validity, int64_t x_offset,
                     const void *y, int y_is_float, const uint8_t *y_validity, int64_t y_offset,
                     int64_t n, orc_corr_t *out) {
  memset(out, 0, sizeof(*out));
  uint64_t count = 0;
  double mean1 = 0, mean2 = 0, c = 0;
  welford_t w1 = {0, 0, 0}, w2 = {0, 0, 0};
  for (int64_t i = 0; i < n; i++) {
    if (!bit_is_set(x_validity, x_offset + i) || !bit_is_set(y_validity, y_offset + i)) continue;
    double v1 = num_at(x, x_is_float, x_offset + i), v2 = num_at(y, y_is_float, y_offset + i);
    count += 1;
    double delta1 = v1 - mean1;
    double new_mean1 = delta1 / (double)count + mean1;
    double delta2 = v2 - mean2;
    double new_mean2 = delta2 / (double)count + mean2;
    c += delta1 * (v2 - new_mean2);
    mean1 = new_mean1;
    mean2 = new_mean2;
    welford_add(&w1, v1);
    welford_add(&w2, v2);
  }
  out->n = count;
  out->corr = NAN;
  out->covar_samp = NAN;
  if (count >= 1) {
    double covar_pop = c / (double)count;
    double s1 = sqrt(w1.m2 / (double)count), s2 = sqrt(w2.m2 / (double)count);
    out->corr_has_value = 1;
    out->corr = (s1 == 0.0 || s2 == 0.0) ? 0.0 : covar_pop / s1 / s2;
  }
  if (count >= 2) {
    out->covar_has_value = 1;
    out->covar_samp = c / (double)(count - 1);
  }
}

typedef struct {
  double v;
  int64_t idx;
} rank_item_t;
static int cmp_rank_item(const void *a, const void *b) {
  /* ORDER BY CAST(c AS DOUBLE) ascending; DataFusion sorts floats by total order */
  int64_t x = f64_total_key(((const rank_item_t *)a)->v),
          y = f64_total_key(((const rank_item_t *)b)->v);
  return x < y ? -1 : x > y;
}

/* min-rank (SQL RANK()) of v[0..m) into r[0..m) */
static int min_ranks(const double *v, int64_t m, uint64_t *r) {
  rank_item_t *it = (rank_item_t *)malloc((size_t)(m > 0 ? m : 1) * sizeof(rank_item_t));
  if (!it) return -1;
  for (int64_t i = 0; i < m; i++) {
    it[i].v = v[i];
    it[i].idx = i;
  }
  qsort(it, (size_t)m, sizeof(rank_item_t), cmp_rank_item);
  int64_t i = 0;
  while (i < m) {
    int64_t j = i + 1;
    while (j < m && f64_total_key(it[j].v) == f64_total_key(it[i].v)) j++;
    for (int64_t t = i; t < j; t++) r[it[t].idx] = (uint64_t)(i + 1);
    i = j;
  }
  free(it);
  return 0;
}

/* TG/analyzers/advanced/correlation.rs:334-350. RANK() is UInt64; SUM over UInt64 and the
 * products rank*rank wrap modulo 2^64 in DataFusion (arrow wrapping arithmetic). */
int orc_spearman_state(const void *x, int x_is_float, const uint8_t *x_validity, int64_t x_offset,
                       const void *y, int y_is_float, const uint8_t *y_validity, int64_t y_offset,
                       int64_t n, orc_comoments_t *out) {
  memset(out, 0, sizeof(*out));
  size_t cap = (size_t)(n > 0 ? n : 1);
  double *xv = (double *)malloc(cap * sizeof(double));
  double *yv = (double *)malloc(cap * sizeof(double));
  uint64_t *rx = (uint64_t *)malloc(cap * sizeof(uint64_t));
  uint64_t *ry = (uint64_t *)malloc(cap * sizeof(uint64_t));
  if (!xv || !yv || !rx || !ry) {
    free(xv); free(yv); free(rx); free(ry);
    return -1;
  }
  int64_t m = 0;
  for (int64_t i = 0; i < n; i++) {
    if (!bit_is_set(x_validity, x_offset + i) || !bit_is_set(y_validity, y_offset + i)) continue;
    xv[m] = num_at(x, x_is_float, x_offset + i);
    yv[m] = num_at(y, y_is_float, y_offset + i);
    m++;
  }
  int rc = min_ranks(xv, m, rx) | min_ranks(yv, m, ry);
  uint64_t sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
  for (int64_t i = 0; i < m; i++) {
    sx += rx[i];
    sy += ry[i];
    sxx += rx[i] * rx[i];
    syy += ry[i] * ry[i];
    sxy += rx[i] * ry[i];
  }
  out->n = (uint64_t)m;
  out->sum_x = (double)sx;
  out->sum_y = (double)sy;
  out->sum_x2 = (double)sxx;
  out->sum_y2 = (double)syy;
  out->sum_xy = (double)sxy;
  free(xv); free(yv); free(rx); free(ry);
  return rc;
}

/* ------------------------------------------------------------------ K9: KllSketch */
/* SipHash (c_rounds, d_rounds); Rust's std DefaultHasher is SipHash-1-3 with k0=k1=0. */
#define ROTL64(x, b) (((x) << (b)) | ((x) >> (64 - (b))))
#define SIPROUND(v0, v1, v2, v3) \
  do {                           \
    v0 += v1; v1 = ROTL64(v1, 13); v1 ^= v0; v0 = ROTL64(v0, 32); \
    v2 += v3; v3 = ROTL64(v3, 16); v3 ^= v2;                      \
    v0 += v3; v3 = ROTL64(v3, 21); v3 ^= v0;                      \
    v2 += v1; v1 = ROTL64(v1, 17); v1 ^= v2; v2 = ROTL64(v2, 32); \
  } while (0)

uint64_t orc_siphash(int c_rounds, int d_rounds, uint64_t k0, uint64_t k1, const uint8_t *msg,
                     size_t len) {
  uint64_t v0 = k0 ^ 0x736f6d6570736575ULL, v1 = k1 ^ 0x646f72616e646f6dULL;
  uint64_t v2 = k0 ^ 0x6c7967656e657261ULL, v3 = k1 ^ 0x7465646279746573ULL;
  size_t full = len & ~(size_t)7;
  for (size_t i = 0; i < full; i += 8) {
    uint64_t m = 0;
    for (int j = 0; j < 8; j++) m |= (uint64_t)msg[i + j] << (8 * j);
    v3 ^= m;
    for (int r = 0; r < c_rounds; r++) SIPROUND(v0, v1, v2, v3);
    v0 ^= m;
  }
  uint64_t b = (uint64_t)len << 56;
  for (size_t j = 0; j < (len & 7); j++) b |= (uint64_t)msg[full + j] << (8 * j);
  v3 ^= b;
  for (int r = 0; r < c_rounds; r++) SIPROUND(v0, v1, v2, v3);
  v0 ^= b;
  v2 ^= 0xff;
  for (int r = 0; r < d_rounds; r++) SIPROUND(v0, v1, v2, v3);
  return v0 ^ v1 ^ v2 ^ v3;
}

typedef struct {
  uint64_t capacity;
  double *items;
  uint64_t len, cap_alloc;
  int sorted;
} compactor_t;

struct orc_kll {
  uint64_t k;
  compactor_t *levels;
  uint64_t n_levels, cap_levels;
  uint64_t n;
  double min_value, max_value;
  int parity_mode;
  uint64_t rng;
};

/* kll_sketch.rs:183-192 */
uint64_t orc_kll_level_capacity(uint64_t k, uint64_t level) {
  uint64_t c;
  switch (level) {
    case 0: return k;
    case 1: c = (k * 2) / 3; return c > 8 ? c : 8;
    case 2: c = k / 2; return c > 4 ? c : 4;
    case 3: c = k / 4; return c > 4 ? c : 4;
    case 4: c = k / 8; return c > 4 ? c : 4;
    default: return 4;
  }
}

static void comp_push(compactor_t *c, double v) {
  if (c->len == c->cap_alloc) {
    c->cap_alloc = c->cap_alloc ? c->cap_alloc * 2 : 16;
    c->items = (double *)realloc(c->items, c->cap_alloc * sizeof(double));
  }
  c->items[c->len++] = v;
}

static void kll_push_level(orc_kll *s, uint64_t capacity) {
  if (s->n_levels == s->cap_levels) {
    s->cap_levels = s->cap_levels ? s->cap_levels * 2 : 8;
    s->levels = (compactor_t *)realloc(s->levels, s->cap_levels * sizeof(compactor_t));
  }
  compactor_t *c = &s->levels[s->n_levels++];
  c->capacity = capacity;
  c->items = NULL;
  c->len = c->cap_alloc = 0;
  c->sorted = 1;
}

orc_kll *orc_kll_new(uint64_t k, int parity_mode, uint64_t seed) {
  if (k < 2) return NULL; /* kll_sketch.rs:167-169 panics */
  orc_kll *s = (orc_kll *)calloc(1, sizeof(orc_kll));
  s->k = k;
  s->min_value = INFINITY;
  s->max_value = -INFINITY;
  s->parity_mode = parity_mode;
  s->rng = seed ? seed : 0x9E3779B97F4A7C15ULL;
  kll_push_level(s, k);
  return s;
}

void orc_kll_free(orc_kll *s) {
  if (!s) return;
  for (uint64_t i = 0; i < s->n_levels; i++) free(s->levels[i].items);
  free(s->levels);
  free(s);
}

/* slice::sort_by(partial_cmp().unwrap_or(Equal)) is a stable sort; NaN never enters
 * (update() drops it), so an insertion-stable merge sort on plain `<` is equivalent. */
static void stable_sort_f64(double *a, uint64_t n) {
  if (n < 2) return;
  double *tmp = (double *)malloc(n * sizeof(double));
  for (uint64_t w = 1; w < n; w *= 2) {
    for (uint64_t lo = 0; lo < n; lo += 2 * w) {
      uint64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      uint64_t i = lo, j = mid, o = lo;
      while (i < mid && j < hi) tmp[o++] = (a[j] < a[i]) ? a[j++] : a[i++];
      while (i < mid) tmp[o++] = a[i++];
      while (j < hi) tmp[o++] = a[j++];
    }
    memcpy(a, tmp, n * sizeof(double));
  }
  free(tmp);
}

static void comp_ensure_sorted(compactor_t *c) {
  if (!c->sorted) {
    stable_sort_f64(c->items, c->len);
    c->sorted = 1;
  }
}

/* Rust `f64 as u64`: saturating, NaN -> 0 */
static uint64_t f64_as_u64(double d) {
  if (!(d > 0.0)) return 0;
  if (d >= 18446744073709551616.0) return UINT64_MAX;
  return (uint64_t)d;
}

/* kll_sketch.rs:80-102 */
static int select_keep_odd(orc_kll *s, const compactor_t *c) {
  if (s->parity_mode == 1) {
    uint64_t x = s->rng;
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    s->rng = x;
    return (int)((x >> 33) & 1);
  }
  uint8_t msg[16];
  uint64_t len = c->len;
  size_t m = 8;
  memcpy(msg, &len, 8); /* usize::hash -> write(&to_ne_bytes) */
  if (c->len > 0) {
    uint64_t f = f64_as_u64(c->items[0]);
    memcpy(msg + 8, &f, 8);
    m = 16;
  }
  return (int)(orc_siphash(1, 3, 0, 0, msg, m) % 2 == 1);
}

/* kll_sketch.rs:57-76: the items at the NON-selected parity are handed to the next level,
 * the selected parity stays at this level (nothing is discarded). */
static void kll_compact_into(orc_kll *s, uint64_t level) {
  compactor_t *c = &s->levels[level];
  comp_ensure_sorted(c);
  int keep_odd = select_keep_odd(s, c);
  uint64_t kept = 0, len = c->len;
  for (uint64_t i = 0; i < len; i++) {
    double item = s->levels[level].items[i];
    if (((i % 2) == 1) == (keep_odd != 0)) {
      s->levels[level].items[kept++] = item;
    } else {
      comp_push(&s->levels[level + 1], item);
      s->levels[level + 1].sorted = 0;
    }
  }
  s->levels[level].len = kept;
  s->levels[level].sorted = 1;
}

/* kll_sketch.rs:213-229 */
static void kll_cascade(orc_kll *s) {
  uint64_t level = 0;
  while (level < s->n_levels && s->levels[level].len >= s->levels[level].capacity) {
    if (level + 1 >= s->n_levels) kll_push_level(s, orc_kll_level_capacity(s->k, level + 1));
    kll_compact_into(s, level);
    level += 1;
  }
}

/* kll_sketch.rs:195-210 */
void orc_kll_update(orc_kll *s, double v) {
  if (isnan(v)) return;
  s->n += 1;
  s->min_value = fmin(s->min_value, v);
  s->max_value = fmax(s->max_value, v);
  comp_push(&s->levels[0], v);
  s->levels[0].sorted = 0;
  kll_cascade(s);
}

/* the callers' loop: tests/advanced_analytics_integration.rs:75-99 (skip NULL slots) */
void orc_kll_update_many(orc_kll *s, const double *v, const uint8_t *validity, int64_t offset,
                         int64_t n) {
  for (int64_t i = 0; i < n; i++)
    if (bit_is_set(validity, offset + i)) orc_kll_update(s, v[offset + i]);
}

/* kll_sketch.rs:327-366 */
int orc_kll_merge(orc_kll *dst, const orc_kll *src) {
  if (dst->k != src->k) return -1;
  dst->n += src->n;
  dst->min_value = fmin(dst->min_value, src->min_value);
  dst->max_value = fmax(dst->max_value, src->max_value);
  for (uint64_t level = 0; level < src->n_levels; level++) {
    while (level >= dst->n_levels) kll_push_level(dst, orc_kll_level_capacity(dst->k, level));
    for (uint64_t i = 0; i < src->levels[level].len; i++)
      comp_push(&dst->levels[level], src->levels[level].items[i]);
    dst->levels[level].sorted = 0;
  }
  /* `for level in 0..self.compactors.len()`: the range is fixed before the loop runs, so
   * levels appended while cascading are not themselves compacted (kll_sketch.rs:353) */
  uint64_t levels_at_entry = dst->n_levels;
  for (uint64_t level = 0; level < levels_at_entry; level++) {
    while (dst->levels[level].len >= dst->levels[level].capacity) {
      if (level + 1 >= dst->n_levels)
        kll_push_level(dst, orc_kll_level_capacity(dst->k, level + 1));
      kll_compact_into(dst, level);
    }
  }
  return 0;
}

typedef struct {
  double v;
  uint64_t w;
  uint64_t seq;
} witem_t;
static int cmp_witem(const void *a, const void *b) {
  const witem_t *x = (const witem_t *)a, *y = (const witem_t *)b;
  if (x->v < y->v) return -1;
  if (x->v > y->v) return 1;
  return x->seq < y->seq ? -1 : x->seq > y->seq; /* stable */
}
static inline uint64_t sat_add(uint64_t a, uint64_t b) {
  uint64_t r = a + b;
  return r < a ? UINT64_MAX : r;
}

/* kll_sketch.rs:246-322 */
int orc_kll_quantile(const orc_kll *s, double phi, double *out) {
  if (s->n == 0) return -1;
  if (!(phi >= 0.0 && phi <= 1.0)) return -1;
  if (phi == 0.0) { *out = s->min_value; return 0; }
  if (phi == 1.0) { *out = s->max_value; return 0; }
  uint64_t total_items = 0;
  for (uint64_t l = 0; l < s->n_levels; l++) total_items += s->levels[l].len;
  if (total_items == 0) return -1;
  witem_t *it = (witem_t *)malloc(total_items * sizeof(witem_t));
  uint64_t m = 0;
  for (uint64_t l = 0; l < s->n_levels; l++) {
    uint64_t weight = l >= 63 ? UINT64_MAX / 2 : (uint64_t)1 << l;
    /* the reference sorts a clone of each compactor first (stable), then the whole list
     * (stable); sorting once with (value, level, position-after-level-sort) is the same order */
    uint64_t len = s->levels[l].len;
    double *tmp = (double *)malloc((len ? len : 1) * sizeof(double));
    memcpy(tmp, s->levels[l].items, len * sizeof(double));
    if (!s->levels[l].sorted) stable_sort_f64(tmp, len);
    for (uint64_t i = 0; i < len; i++) {
      it[m].v = tmp[i];
      it[m].w = weight;
      it[m].seq = m;
      m++;
    }
    free(tmp);
  }
  qsort(it, (size_t)m, sizeof(witem_t), cmp_witem);
  uint64_t total_weight = 0;
  for (uint64_t i = 0; i < m; i++) total_weight = sat_add(total_weight, it[i].w);
  double target_rank = ceil(phi * (double)total_weight);
  uint64_t cum = 0;
  for (uint64_t i = 0; i < m; i++) {
    cum = sat_add(cum, it[i].w);
    if ((double)cum >= target_rank) {
      *out = it[i].v;
      free(it);
      return 0;
    }
  }
  free(it);
  *out = s->max_value;
  return 0;
}

uint64_t orc_kll_count(const orc_kll *s) { return s->n; }
uint64_t orc_kll_num_levels(const orc_kll *s) { return s->n_levels; }
uint64_t orc_kll_num_retained(const orc_kll *s) {
  uint64_t t = 0;
  for (uint64_t l = 0; l < s->n_levels; l++) t += s->levels[l].len;
  return t;
}
double orc_kll_min(const orc_kll *s) { return s->min_value; }
double orc_kll_max(const orc_kll *s) { return s->max_value; }
/* kll_sketch.rs:397-399 */
double orc_kll_relative_error_bound(const orc_kll *s) { return 1.65 / sqrt((double)s->k); }
uint64_t orc_kll_level_items(const orc_kll *s, uint64_t level, double *out, uint64_t cap) {
  if (level >= s->n_levels) return 0;
  uint64_t len = s->levels[level].len;
  for (uint64_t i = 0; i < len && i < cap; i++) out[i] = s->levels[level].items[i];
  return len;
}
