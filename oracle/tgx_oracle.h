/*
 * tgx_oracle.h -- CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * A scalar, plain-C restatement of the per-row arithmetic behind term-guard's
 * Arrow-batch check evaluator (kernels K1..K12).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; nothing under term_amd/ links, imports or calls it.
 *
 * Pinning: the reference (Rust + DataFusion 50.3.0 / arrow 56.2.0 /
 * regex 1.12.2, none of which is vendored under /root/reference and none of
 * which can be compiled here -- no cargo/rustc) cannot be run in this image.
 * The oracle is therefore pinned against the reference's own known-answer
 * unit tests, restated as tests/golden/ JSON fixtures (see tests/test_oracle_golden.py).
 * What no reference test constrains is listed as "parity unpinned" in
 * DESIGN.md.
 *
 * All column arguments use the Arrow layout: `validity` is an LSB-first bitmap
 * (NULL pointer = no nulls), `offset` is the logical start (applies to validity
 * bits and to value / offsets slots alike), `n` the number of rows.
 *
 * Paths cited as TG/... are /root/reference/term-guard/src/...
 */
#ifndef TGX_ORACLE_H
#define TGX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- K1: COUNT(*) / COUNT(col)  (TG/constraints/completeness.rs:158-163) ---- */
typedef struct {
  int64_t total;
  int64_t non_null;
} orc_count_t;
void orc_count(const uint8_t *validity, int64_t offset, int64_t n, orc_count_t *out);

/* ---- K2/K3/K4: MIN MAX SUM AVG STDDEV VARIANCE (TG/constraints/statistics.rs:45-74) ---- */
typedef struct {
  int64_t total;
  int64_t non_null;
  int32_t has_value;        /* 0 => every aggregate is SQL NULL */
  int32_t is_float;
  int64_t min_i, max_i;     /* int64 columns */
  double min_f, max_f;      /* float64 columns, IEEE totalOrder (arrow-arith aggregate) */
  int64_t sum_i_wrapping;   /* SUM(int64): two's-complement wrapping add */
  double sum_f;             /* SUM(float64), or SUM(CAST(int64 AS DOUBLE)) = AVG numerator */
  double sum_hi;            /* same sum in long double, rounded once: accuracy yardstick */
  double mean;              /* sum_f / non_null */
  double sumsq_f;           /* SUM(x*x) over doubles (TG/analyzers/advanced/standard_deviation.rs:172-179) */
  int32_t has_variance;     /* non_null >= 2 */
  double var_samp;          /* Welford, as DataFusion's VarianceAccumulator */
  double stddev_samp;
} orc_stats_t;
void orc_stats_i64(const int64_t *values, const uint8_t *validity, int64_t offset, int64_t n,
                   orc_stats_t *out);
void orc_stats_f64(const double *values, const uint8_t *validity, int64_t offset, int64_t n,
                   orc_stats_t *out);

/* ---- K5/K6: COUNT(DISTINCT c) and the GROUP BY value-count query
 * (TG/constraints/uniqueness.rs:612-617, 671-681, 709-715) ---- */
typedef struct {
  int64_t total;
  int64_t non_null;
  int64_t distinct;         /* COUNT(DISTINCT c): NULLs excluded */
  int64_t groups_once;      /* #GROUP BY groups with cnt == 1; NULL is a group of its own */
} orc_distinct_t;
/* 64-bit fixed width keys compared by bit pattern (int64, and float64 as DataFusion hashes it) */
int orc_distinct_bits64(const uint64_t *bits, const uint8_t *validity, int64_t offset, int64_t n,
                        orc_distinct_t *out);
/* ---- the null + range + unique suite on T threads (suite_mt.c): what bench.py's cpu_baseline times.
 * Row-range partitions, per-partition partial states, hash-set COUNT(DISTINCT) re-partitioned by owner, merge in
 * partition order -- the shape DataFusion gives the reference's per-constraint queries (TG/core/suite.rs:67-100,
 * TG/analyzers/traits.rs:160-170).  counts / stats: one per column; distinct: one per entry of unique_cols.
 * Returns 0, -1 on allocation failure. */
typedef struct {
  const void *values;       /* int64 or float64, 8 bytes per row */
  const uint8_t *validity;  /* NULL = no nulls */
  int32_t is_float;
  int32_t reserved;
} orc_suite_column_t;
int orc_suite_mt(const orc_suite_column_t *cols, int32_t n_cols, const int32_t *unique_cols, int32_t n_unique,
                 int64_t n, int32_t n_threads, orc_count_t *counts, orc_stats_t *stats, orc_distinct_t *distinct);

/* Utf8 (int32 offsets). */
int orc_distinct_utf8(const int32_t *offsets, const uint8_t *data, const uint8_t *validity,
                      int64_t offset, int64_t n, orc_distinct_t *out);

/* ---- K11: raw co-moments (TG/analyzers/advanced/correlation.rs:239-249) ---- */
typedef struct {
  uint64_t n;
  double sum_x, sum_y, sum_x2, sum_y2, sum_xy;
} orc_comoments_t;
/* x_is_float / y_is_float: 0 => int64 values CAST AS DOUBLE, 1 => float64 */
void orc_comoments(const void *x, int x_is_float, const uint8_t *x_validity, int64_t x_offset,
                   const void *y, int y_is_float, const uint8_t *y_validity, int64_t y_offset,
                   int64_t n, orc_comoments_t *out);
/* metric formulas (TG/analyzers/advanced/correlation.rs:407-432) */
double orc_pearson_from_state(const orc_comoments_t *s);
double orc_covariance_from_state(const orc_comoments_t *s);

/* K10: CORR(a,b) / COVAR_SAMP(a,b) as DataFusion's online accumulators compute them
 * (TG/constraints/correlation.rs:260-275). has_value=0 => SQL NULL. */
typedef struct {
  uint64_t n;
  int32_t corr_has_value;
  double corr;
  int32_t covar_has_value;
  double covar_samp;
} orc_corr_t;
void orc_corr_online(const void *x, int x_is_float, const uint8_t *x_validity, int64_t x_offset,
                     const void *y, int y_is_float, const uint8_t *y_validity, int64_t y_offset,
                     int64_t n, orc_corr_t *out);

/* K12: Spearman via SQL RANK() (min-rank ties, UInt64 arithmetic that wraps)
 * (TG/analyzers/advanced/correlation.rs:334-350). Fills the state with the rank sums as f64. */
int orc_spearman_state(const void *x, int x_is_float, const uint8_t *x_validity, int64_t x_offset,
                       const void *y, int y_is_float, const uint8_t *y_validity, int64_t y_offset,
                       int64_t n, orc_comoments_t *out);

/* ---- K9: KllSketch (TG/analyzers/advanced/kll_sketch.rs) ---- */
typedef struct orc_kll orc_kll;
/* parity_mode: 0 = SipHash-1-3 branch (cfg(not(feature="test-utils")), kll_sketch.rs:87-101)
 *              1 = coin flips from a seeded xorshift (stands in for rand::rng(), :80-85)  */
orc_kll *orc_kll_new(uint64_t k, int parity_mode, uint64_t seed);
void orc_kll_free(orc_kll *s);
void orc_kll_update(orc_kll *s, double v);
void orc_kll_update_many(orc_kll *s, const double *v, const uint8_t *validity, int64_t offset,
                         int64_t n);
int orc_kll_merge(orc_kll *dst, const orc_kll *src);       /* -1 if k differs */
int orc_kll_quantile(const orc_kll *s, double phi, double *out); /* -1 on error (empty / phi) */
uint64_t orc_kll_count(const orc_kll *s);
uint64_t orc_kll_num_levels(const orc_kll *s);
uint64_t orc_kll_num_retained(const orc_kll *s);
double orc_kll_min(const orc_kll *s);
double orc_kll_max(const orc_kll *s);
double orc_kll_relative_error_bound(const orc_kll *s);
/* copy level `level` items into out (cap slots); returns item count */
uint64_t orc_kll_level_items(const orc_kll *s, uint64_t level, double *out, uint64_t cap);
uint64_t orc_kll_level_capacity(uint64_t k, uint64_t level);
/* exposed for a structural known-answer test of the hash */
uint64_t orc_siphash(int c_rounds, int d_rounds, uint64_t k0, uint64_t k1, const uint8_t *msg,
                     size_t len);

/* ---- APPROX_DISTINCT (TG/constraints/approx_count_distinct.rs:56-66): HyperLogLog, 2^14 one-byte registers ---- */
void orc_hll_registers(const uint64_t *bits, const uint8_t *validity, int64_t offset, int64_t n, uint8_t *registers);
uint64_t orc_hll_estimate(const uint8_t *registers);

/* ---- K7: pattern checks (TG/constraints/format.rs:750-776) ---- */
typedef struct orc_regex orc_regex;
/* compile with Rust-regex syntax; case_insensitive = the SQL `~*` operator.
 * Returns NULL and fills err (cap bytes) on a pattern the engine does not cover. */
orc_regex *orc_regex_compile(const char *pattern, size_t len, int case_insensitive, char *err,
                             size_t cap);
void orc_regex_free(orc_regex *re);
/* unanchored search, like regex::Regex::is_match */
int orc_regex_is_match(const orc_regex *re, const uint8_t *s, size_t len);
typedef struct {
  int64_t total;
  int64_t matches;
} orc_match_t;
/* COUNT(CASE WHEN [TRIM(]c[)] ~ pat [OR c IS NULL] THEN 1 END), COUNT(*) */
void orc_length_count_utf8(const int32_t *offsets, const uint8_t *data, const uint8_t *validity, int64_t offset,
                           int64_t n, uint64_t min_chars, uint64_t max_chars, orc_match_t *out);
void orc_regex_count_utf8(const orc_regex *re, const int32_t *offsets, const uint8_t *data,
                          const uint8_t *validity, int64_t offset, int64_t n, int trim,
                          int null_is_valid, orc_match_t *out);

#ifdef __cplusplus
}
#endif
#endif
