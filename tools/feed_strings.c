/* feed_strings.c -- a plain-C consumer of include/tgx.h for HOST string columns: streams 8 Mi e-mail addresses through
 * completeness + FormatType::Email + LENGTH bounds as DataFusion hands a column out -- 8192-row RecordBatches
 * (TG/core/context.rs:28-38) -- and as 65 536-row batches and as one batch, in the three layouts a Parquet string column
 * arrives in: Utf8 (offsets + bytes), Utf8View (16-byte views into data buffers of 64 Ki rows each, shared by the batches
 * cut from them) and Dictionary<Int32, Utf8> (one dictionary of 131 072 entries for all batches).  No Python between the
 * calls (tools/bench_host_strings.py is the same stream through the ctypes binding: 4-10 us per call, which is most of a
 * dictionary batch's cost).
 *
 *   build:  make -C tools            (gcc; links term_amd/libtgx.so and the HIP runtime)
 *   run:    build/feed_strings [layout]      layout: utf8 | view | dict (default: all three)
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/tgx.h"

#define CHECK_HIP(x)                                                  \
  do {                                                                \
    hipError_t e_ = (x);                                              \
    if (e_ != hipSuccess) {                                           \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));         \
      exit(1);                                                        \
    }                                                                 \
  } while (0)
#define CHECK_TGX(x)                                                  \
  do {                                                                \
    tgx_status s_ = (x);                                              \
    if (s_ != TGX_OK) {                                               \
      fprintf(stderr, "%s: %s: %s\n", #x, tgx_status_name(s_), err.msg); \
      exit(1);                                                        \
    }                                                                 \
  } while (0)

/* FormatType::Email (TG/constraints/format.rs): the pattern the reference's e-mail check compiles */
static const char kEmail[] = "^[a-zA-Z0-9.!#$%&'*+/=?^_`{|}~-]+@[a-zA-Z0-9](?:[a-zA-Z0-9-]{0,61}[a-zA-Z0-9])?"
                             "(?:\\.[a-zA-Z0-9](?:[a-zA-Z0-9-]{0,61}[a-zA-Z0-9])?)*$";

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static uint64_t mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}

enum { kDistinct = 131072, kRowsPerBuf = 65536 };

int main(int argc, char **argv) {
  const char *only = argc > 1 ? argv[1] : NULL;
  const long only_rows = argc > 2 ? atol(argv[2]) : 0; /* one batch size only (for a profiler) */
  const int64_t n = (int64_t)8192 * 1024;
  tgx_error err;
  memset(&err, 0, sizeof(err));
  tgx_options opts = {-1, 0, 0};
  CHECK_TGX(tgx_init(&opts, &err));

  /* 131 072 distinct addresses "user<k>@example<k % 1000>.com", repeated to n rows */
  int32_t *d_off = malloc((kDistinct + 1) * sizeof(int32_t));
  uint8_t *d_data = malloc((size_t)kDistinct * 40 + 64);
  d_off[0] = 0;
  for (int i = 0; i < kDistinct; i++) {
    const uint64_t u = mix64(5 + (uint64_t)i) % 1000000000ull;
    const int len = sprintf((char *)d_data + d_off[i], "user%llu@example%llu.com", (unsigned long long)u,
                            (unsigned long long)(u % 1000));
    d_off[i + 1] = d_off[i] + len;
  }
  const int64_t dict_bytes = d_off[kDistinct];
  const int64_t reps = n / kDistinct;
  /* Utf8: offsets + bytes of all n rows */
  int32_t *offsets = malloc((size_t)(n + 1) * sizeof(int32_t));
  uint8_t *data = malloc((size_t)dict_bytes * reps + 64);
  for (int64_t r = 0; r < reps; r++) {
    memcpy(data + r * dict_bytes, d_data, (size_t)dict_bytes);
    for (int i = 0; i < kDistinct; i++) offsets[r * kDistinct + i] = (int32_t)(r * dict_bytes + d_off[i]);
  }
  offsets[n] = (int32_t)(reps * dict_bytes);
  /* Utf8View: {length, 4-byte prefix, buffer index, offset}; one data buffer per 64 Ki rows */
  const int n_bufs = (int)((n + kRowsPerBuf - 1) / kRowsPerBuf);
  int32_t *views = malloc((size_t)n * 16);
  const uint8_t **bufs = malloc(n_bufs * sizeof(uint8_t *));
  int64_t *buf_sizes = malloc(n_bufs * sizeof(int64_t));
  for (int b = 0; b < n_bufs; b++) {
    const int64_t r0 = (int64_t)b * kRowsPerBuf, r1 = r0 + kRowsPerBuf < n ? r0 + kRowsPerBuf : n;
    buf_sizes[b] = (int64_t)offsets[r1] - offsets[r0];
    uint8_t *p = malloc((size_t)buf_sizes[b] + 16);
    memcpy(p, data + offsets[r0], (size_t)buf_sizes[b]);
    bufs[b] = p;
    for (int64_t i = r0; i < r1; i++) {
      views[4 * i] = offsets[i + 1] - offsets[i];
      memcpy(&views[4 * i + 1], data + offsets[i], 4);
      views[4 * i + 2] = b;
      views[4 * i + 3] = offsets[i] - offsets[r0];
    }
  }
  /* Dictionary<Int32, Utf8> */
  int32_t *indices = malloc((size_t)n * sizeof(int32_t));
  for (int64_t i = 0; i < n; i++) indices[i] = (int32_t)(i % kDistinct);
  tgx_column dictionary;
  memset(&dictionary, 0, sizeof(dictionary));
  dictionary.type = TGX_UTF8;
  dictionary.mem = TGX_MEM_HOST;
  dictionary.length = kDistinct;
  dictionary.null_count = 0;
  dictionary.offsets = d_off;
  dictionary.data = d_data;

  tgx_check_spec specs[3];
  memset(specs, 0, sizeof(specs));
  for (int k = 0; k < 3; k++) {
    specs[k].column = 0;
    specs[k].column2 = -1;
  }
  specs[0].kind = TGX_CHECK_COUNT;
  specs[1].kind = TGX_CHECK_REGEX_MATCH;
  specs[1].pattern = kEmail;
  specs[1].pattern_len = strlen(kEmail);
  specs[2].kind = TGX_CHECK_LENGTH;
  specs[2].length_min = 5;
  specs[2].length_max = 64;
  tgx_plan *plan = NULL;
  CHECK_TGX(tgx_plan_create(specs, 3, &plan, &err));
  /* the fourth leg: the Utf8 column again with a uniqueness check BY VALUE on top (an exact key set: what the shim's
   * `uniqueness` asks for; small batches take the table path, where the keys' bytes are kept) */
  tgx_check_spec specs_keys[4];
  memcpy(specs_keys, specs, sizeof(specs));
  memset(&specs_keys[3], 0, sizeof(specs_keys[3]));
  specs_keys[3].kind = TGX_CHECK_DISTINCT;
  specs_keys[3].column = 0;
  specs_keys[3].column2 = -1;
  specs_keys[3].flags = getenv("TGX_FEED_FINGERPRINT_KEYS") ? 0 : TGX_FLAG_EXACT_KEYS; /* (for A/B runs) */
  tgx_plan *plan_keys = NULL;
  CHECK_TGX(tgx_plan_create(specs_keys, 4, &plan_keys, &err));
  int64_t n_distinct = 0; /* (the generator's 131 072 draws repeat now and then) */
  {
    uint64_t *us = malloc(kDistinct * sizeof(uint64_t));
    for (int i = 0; i < kDistinct; i++) us[i] = mix64(5 + (uint64_t)i) % 1000000000ull;
    /* (a shell sort: no libc callback needed) */
    for (int gap = kDistinct / 2; gap > 0; gap /= 2)
      for (int i = gap; i < kDistinct; i++) {
        const uint64_t v = us[i];
        int j = i;
        for (; j >= gap && us[j - gap] > v; j -= gap) us[j] = us[j - gap];
        us[j] = v;
      }
    for (int i = 0; i < kDistinct; i++) n_distinct += i == 0 || us[i] != us[i - 1];
    free(us);
  }
  tgx_result res[4];

  const char *layouts[4] = {"utf8", "view", "dict", "utf8keys"};
  const char *names[4] = {"Utf8", "Utf8View", "Dictionary<Int32, Utf8>", "Utf8"};
  const double row_bytes[4] = {(double)offsets[n] / (double)n + 4.0, (double)offsets[n] / (double)n + 16.0,
                               4.0 + ((double)dict_bytes + 4.0 * kDistinct) / (double)n, (double)offsets[n] / (double)n + 4.0};
  for (int l = 0; l < 4; l++) {
    if (only && strcmp(only, layouts[l]) != 0) continue;
    const int with_keys = l == 3;
    tgx_plan *const plan_l = with_keys ? plan_keys : plan;
    tgx_state *st = NULL;
    CHECK_TGX(tgx_state_create(plan_l, NULL, &st, &err));
    /* the last two legs: the same batches as TGX_MEM_HOST_RETAINED -- buffers the caller keeps as they are until
     * tgx_finalize (what a consumer that holds its RecordBatches can promise): the copies wait for the flush */
    const int64_t batch_sizes[5] = {n, 65536, 8192, 65536, 8192};
    for (int b = 0; b < 5; b++) {
      const int64_t rows = batch_sizes[b];
      const int kept = b >= 3;
      if (only_rows > 0 && (rows != only_rows || kept)) continue;
      if (only_rows < 0 && (rows != -only_rows || !kept)) continue; /* negative: the kept leg of that size */
      double best = 1e30;
      for (int rep = 0; rep < 4; rep++) { /* the first pass allocates: best of the rest */
        CHECK_TGX(tgx_state_reset(plan_l, st, &err));
        CHECK_HIP(hipDeviceSynchronize());
        const double t0 = now();
        for (int64_t lo = 0; lo < n; lo += rows) {
          tgx_column c;
          memset(&c, 0, sizeof(c));
          c.mem = kept ? TGX_MEM_HOST_RETAINED : TGX_MEM_HOST;
          dictionary.mem = c.mem;
          c.length = lo + rows <= n ? rows : n - lo;
          c.offset = lo; /* a slice of the column's buffers, as Arrow hands them out */
          c.null_count = 0;
          if (l == 0 || l == 3) {
            c.type = TGX_UTF8;
            c.offsets = offsets;
            c.data = data;
          } else if (l == 1) {
            c.type = TGX_UTF8_VIEW;
            c.values = views;
            c.variadic = bufs;
            c.variadic_sizes = buf_sizes;
            c.n_variadic = n_bufs;
          } else {
            c.type = TGX_DICT32_UTF8;
            c.values = indices;
            c.dictionary = &dictionary;
          }
          CHECK_TGX(tgx_update(plan_l, st, &c, 1, &err));
        }
        CHECK_TGX(tgx_finalize(plan_l, st, res, with_keys ? 4 : 3, &err));
        const double dt = now() - t0;
        if (rep > 0 && dt < best) best = dt;
      }
      const int ok = res[0].total == n && res[0].non_null == n && res[1].matches == n && res[2].matches == n &&
                     (!with_keys || res[3].distinct == n_distinct);
      const int64_t updates = (n + rows - 1) / rows;
      printf("{\"workload\": \"HOST %s column%s%s, %lld rows x %.0f B, completeness + e-mail format + length (plain C)\", "
             "\"batch_rows\": %lld, \"updates\": %lld, \"total_ms\": %.3f, \"us_per_update\": %.3f, \"rows_per_s\": %.4g, "
             "\"host_to_device_GBs\": %.3g, \"verified\": %s}\n",
             names[l], with_keys ? " + uniqueness by value" : "", kept ? " (kept until finalize)" : "", (long long)n, row_bytes[l], (long long)rows, (long long)updates, best * 1e3,
             best * 1e6 / (double)updates, (double)n / best, (double)n * row_bytes[l] / best / 1e9, ok ? "true" : "false");
      fflush(stdout);
    }
    tgx_state_destroy(st);
  }
  tgx_plan_destroy(plan);
  tgx_plan_destroy(plan_keys);
  return 0;
}
