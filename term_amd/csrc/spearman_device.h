// spearman_device.h -- SPEARMAN tasks of a plan/state (kernels/spearman.hip); see spearman_device.cpp.
#pragma once
#include <functional>
#include <vector>

#include "internal.h"

namespace tgx {
struct SpearmanTask {
  int col_x, col_y;
  bool exact_sums;
};
tgx_status spearman_plan_add(tgx_plan *plan, int spec_index, int *slot, tgx_error *err);
void spearman_plan_free(tgx_plan *plan);
size_t spearman_num_tasks(const tgx_plan *plan);
void spearman_mark_used(const tgx_plan *plan, std::vector<char> &used, std::vector<char> &reads_values);
void spearman_state_init(tgx_state *st);
void spearman_state_free(tgx_state *st);
void spearman_state_reset(tgx_state *st);
// `columns`: the batch as the caller handed it over; lendable: its DEVICE buffers stay as they are until the next
// tgx_finalize / tgx_state_sync (include/tgx.h) -- not a flush's gathered columns, not a producer that recycles buffers
tgx_status spearman_update(tgx_state *st, const tgx_column *dev_columns, const tgx_column *columns, bool lendable,
                           tgx_error *err);
// a batch the state only holds a view of becomes pairs of its own (the caller may release the batch after this)
tgx_status spearman_resolve_all(tgx_state *st, tgx_error *err);
tgx_status spearman_fill_result(tgx_state *st, int slot, tgx_result *r, tgx_error *err);
// rank-based states are not mergeable (TG/analyzers/advanced/correlation.rs:103-109): TGX_UNSUPPORTED when non-empty
tgx_status spearman_check_mergeable(tgx_state *st, tgx_error *err);

// ---- across ranks (tgx_allreduce) ----
// RANK() over the union of the ranks' pairs is a distributed sort, not a merge of states: every rank sorts its keys,
// the ranks agree on world-1 splitters from regular samples, every key travels to the rank that owns its value range
// (equal keys meet on one rank), is ranked there (position of its tie run + the keys owned by lower ranks), and the
// rank travels back to the row it came from; the five rank sums are then added up.  allreduce.cpp binds the two
// collectives this needs to its communicator.
struct SpearmanExchange {
  int32_t rank = 0, world = 1;
  // HOST in, HOST out: every rank's `bytes` to every rank
  std::function<tgx_status(const void *h_send, void *h_recv, size_t bytes)> allgather_host;
  // DEVICE in, DEVICE out: counts in elements of `elem` bytes
  std::function<tgx_status(const void *d_send, const uint64_t *send_counts, void *d_recv, const uint64_t *recv_counts,
                           size_t elem)>
      alltoallv;
};
struct SpearmanResolved {  // what a task holds after the exchange: the sums over ALL ranks' pairs
  int64_t total_rows;
  uint64_t pairs;
  unsigned long long wrapped[5], exact_lo[5], exact_hi[5];
};
tgx_status spearman_allreduce(tgx_state *st, const SpearmanExchange &x, std::vector<SpearmanResolved> *out,
                              tgx_error *err);
// the state now answers with `res` (one per task) until it is reset; it takes no further batches
void spearman_install(tgx_state *st, const std::vector<SpearmanResolved> &res);
// while set, spearman_check_mergeable lets the state through (tgx_allreduce carries the tasks' results itself)
void spearman_set_reducing(tgx_state *st, bool on);
}  // namespace tgx
