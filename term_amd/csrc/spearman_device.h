// spearman_device.h -- SPEARMAN tasks of a plan/state (kernels/spearman.hip); see spearman_device.cpp.
#pragma once
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
tgx_status spearman_update(tgx_state *st, const tgx_column *dev_columns, tgx_error *err);
tgx_status spearman_fill_result(tgx_state *st, int slot, tgx_result *r, tgx_error *err);
// rank-based states are not mergeable (TG/analyzers/advanced/correlation.rs:103-109): TGX_UNSUPPORTED when non-empty
tgx_status spearman_check_mergeable(tgx_state *st, tgx_error *err);
}  // namespace tgx
