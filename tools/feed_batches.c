/* feed_batches.c -- a plain-C consumer of include/tgx.h: streams one table through tgx_update as DataFusion-sized
 * RecordBatches (8192 rows, TG/core/context.rs:28-38) and as one batch, from HOST and from DEVICE buffers, and prints
 * rows/s of each.  No Python between the calls: a tgx_update that only notes a small batch costs well under a
 * microsecond, which a ctypes call (2-10 us) would hide.  Also the smallest example of a binding: what a Rust shim does
 * through FFI is what this file does in C.
 *
 *   build:  make -C tools            (hipcc, links term_amd/libtgx.so and the HIP runtime)
 *   run:    build/feed_batches [rows] [cols] [only] [gap_us]   (defaults: 8 Mi rows, 8 columns; only = one case, e.g.
 *           "u1-host-8192": with the uniqueness check, HOST buffers, 8192-row batches -- for a profiler)
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/tgx.h"

#define CHECK_HIP(x)                                                                 \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                        \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)
#define CHECK_TGX(x)                                                                 \
  do {                                                                               \
    tgx_status s_ = (x);                                                             \
    if (s_ != TGX_OK) {                                                              \
      fprintf(stderr, "%s: %s: %s\n", #x, tgx_status_name(s_), err.msg);             \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

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

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : (int64_t)8192 * 1024;
  const int n_cols = argc > 2 ? atoi(argv[2]) : 8;
  const char *only = argc > 3 ? argv[3] : NULL;
  /* microseconds to idle between two updates, +- 50 %: the copy threads of the library spin ~200 us after a job and
   * then sleep, so gaps of that size put every post on the edge of a worker falling asleep (a stress for that handshake) */
  const long gap_us = argc > 4 ? atol(argv[4]) : 0;
  tgx_error err;
  memset(&err, 0, sizeof(err));
  tgx_options opts = {-1, 0, 0};
  CHECK_TGX(tgx_init(&opts, &err));

  /* the table: Int64 / Float64 columns alternating, every second one nullable at 5 %; column 0 = row ids */
  void **h_vals = calloc(n_cols, sizeof(void *)), **d_vals = calloc(n_cols, sizeof(void *));
  uint8_t **h_valid = calloc(n_cols, sizeof(uint8_t *)), **d_valid = calloc(n_cols, sizeof(uint8_t *));
  const size_t vbytes = (size_t)(n + 7) / 8 + 64;
  for (int c = 0; c < n_cols; c++) {
    h_vals[c] = malloc((size_t)n * 8);
    for (int64_t i = 0; i < n; i++) {
      const uint64_t r = mix64(((uint64_t)c << 40) ^ (uint64_t)i);
      if (c == 0)
        ((int64_t *)h_vals[c])[i] = i;
      else if (c & 1)
        ((double *)h_vals[c])[i] = (double)(r >> 11) * (1000.0 / 9007199254740992.0);
      else
        ((int64_t *)h_vals[c])[i] = (int64_t)(r >> 23) - ((int64_t)1 << 40);
    }
    CHECK_HIP(hipMalloc(&d_vals[c], (size_t)n * 8));
    CHECK_HIP(hipMemcpy(d_vals[c], h_vals[c], (size_t)n * 8, hipMemcpyHostToDevice));
    if (c & 1) {
      h_valid[c] = calloc(vbytes, 1);
      for (int64_t i = 0; i < n; i++)
        if (mix64(0x5555 ^ ((uint64_t)c << 44) ^ (uint64_t)i) >= (uint64_t)(0.05 * 18446744073709551616.0))
          h_valid[c][i >> 3] |= (uint8_t)(1u << (i & 7));
      CHECK_HIP(hipMalloc((void **)&d_valid[c], vbytes));
      CHECK_HIP(hipMemcpy(d_valid[c], h_valid[c], vbytes, hipMemcpyHostToDevice));
    }
  }

  for (int with_unique = 0; with_unique < 2; with_unique++) {
    /* completeness + min / max / mean on every column (+ uniqueness of the id column) */
    const int n_specs = 2 * n_cols + with_unique;
    tgx_check_spec *specs = calloc(n_specs, sizeof(tgx_check_spec));
    for (int c = 0; c < n_cols; c++) {
      specs[2 * c].kind = TGX_CHECK_COUNT;
      specs[2 * c].column = c;
      specs[2 * c].column2 = -1;
      specs[2 * c + 1].kind = TGX_CHECK_NUMERIC_STATS;
      specs[2 * c + 1].column = c;
      specs[2 * c + 1].column2 = -1;
    }
    if (with_unique) {
      specs[2 * n_cols].kind = TGX_CHECK_DISTINCT;
      specs[2 * n_cols].column = 0;
      specs[2 * n_cols].column2 = -1;
    }
    tgx_plan *plan = NULL;
    CHECK_TGX(tgx_plan_create(specs, n_specs, &plan, &err));
    tgx_result *res = calloc(n_specs, sizeof(tgx_result));
    tgx_column *cols = calloc(n_cols, sizeof(tgx_column));
    /* mem: 1 DEVICE buffers, 0 HOST buffers (read before tgx_update returns), 2 HOST buffers the caller keeps as they
     * are until tgx_finalize (TGX_MEM_HOST_RETAINED: their copies wait for the flush) */
    for (int mem = 2; mem >= 0; mem--) {
      const int64_t batch_sizes[3] = {n, 65536, 8192};
      for (int b = 0; b < 3; b++) {
        const int64_t rows = batch_sizes[b] < n ? batch_sizes[b] : n;
        char tag[64];
        snprintf(tag, sizeof(tag), "u%d-%s-%lld", with_unique, mem == 1 ? "device" : mem == 2 ? "hostkept" : "host", (long long)rows);
        if (only && strcmp(only, tag) != 0) continue;
        tgx_state *st = NULL;
        CHECK_TGX(tgx_state_create(plan, NULL, &st, &err));
        double best = 1e30;
        for (int rep = 0; rep < 4; rep++) {  /* the first pass allocates: best of the rest */
          CHECK_TGX(tgx_state_reset(plan, st, &err));
          CHECK_HIP(hipDeviceSynchronize());
          const double t0 = now();
          for (int64_t lo = 0; lo < n; lo += rows) {
            const int64_t len = lo + rows <= n ? rows : n - lo;
            for (int c = 0; c < n_cols; c++) {
              tgx_column *k = &cols[c];
              k->type = (c == 0 || !(c & 1)) ? TGX_INT64 : TGX_FLOAT64;
              k->mem = mem == 1 ? TGX_MEM_DEVICE : mem == 2 ? TGX_MEM_HOST_RETAINED : TGX_MEM_HOST;
              k->length = len;
              k->offset = lo; /* a slice of the table's buffers, as Arrow hands them out */
              k->null_count = -1;
              k->values = mem == 1 ? d_vals[c] : h_vals[c];
              k->validity = mem == 1 ? d_valid[c] : h_valid[c];
            }
            CHECK_TGX(tgx_update(plan, st, cols, (size_t)n_cols, &err));
            if (gap_us > 0) {
              const long us = gap_us / 2 + (long)(mix64((uint64_t)lo ^ (uint64_t)rep) % (uint64_t)gap_us);
              struct timespec ts = {0, us * 1000};
              nanosleep(&ts, NULL);
            }
          }
          CHECK_TGX(tgx_finalize(plan, st, res, (size_t)n_specs, &err));
          const double dt = now() - t0;
          if (rep > 0 && dt < best) best = dt;
        }
        int ok = res[1].total == n && res[1].min_i == 0 && res[1].max_i == n - 1 && res[1].sum_i == n * (n - 1) / 2;
        if (with_unique) ok = ok && res[2 * n_cols].distinct == n;
        const int64_t updates = (n + rows - 1) / rows;
        printf("{\"suite\": \"null+range x%d%s\", \"buffers\": \"%s\", \"rows\": %lld, \"batch_rows\": %lld, "
               "\"updates\": %lld, \"total_ms\": %.3f, \"us_per_update\": %.3f, \"rows_per_s\": %.4g, \"verified\": %s}\n",
               n_cols, with_unique ? " + unique x1" : "", mem == 1 ? "device" : mem == 2 ? "host, kept until finalize" : "host", (long long)n, (long long)rows,
               (long long)updates, best * 1e3, best * 1e6 / (double)updates, (double)n / best, ok ? "true" : "false");
        fflush(stdout);
        tgx_state_destroy(st);
      }
    }
    free(cols);
    free(res);
    tgx_plan_destroy(plan);
    free(specs);
  }
  return 0;
}
