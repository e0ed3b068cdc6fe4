// regex_device.h -- pattern-match tasks of a plan/state; see regex_device.cpp.
#pragma once
#include <vector>

#include "internal.h"

namespace tgx {
tgx_status regex_plan_add(tgx_plan *plan, int spec_index, int *slot, tgx_error *err);
// after the last regex_plan_add: groups the patterns of a column whose product automaton fits the LDS table
void regex_plan_finish(tgx_plan *plan);
void regex_plan_free(tgx_plan *plan);
size_t regex_num_tasks(const tgx_plan *plan);
void regex_mark_used(const tgx_plan *plan, std::vector<char> &used);
void regex_state_init(tgx_state *st);
void regex_state_free(tgx_state *st);
void regex_state_reset(tgx_state *st);
// fuse: dictionary columns whose per-row gather the caller will run together with the column's DISTINCT pass
tgx_status regex_update(tgx_state *st, const tgx_column *dev_columns, tgx_error *err, DictFuse *fuse = nullptr);
// bracket a loop of regex_fill_result calls: one readback of all counters instead of one per task
tgx_status regex_fetch_begin(tgx_state *st, tgx_error *err);
void regex_fetch_end(tgx_state *st);
tgx_status regex_fill_result(tgx_state *st, int slot, tgx_result *r, tgx_error *err);
tgx_status regex_merge_states(tgx_state *dst, tgx_state *src, tgx_error *err);
tgx_status regex_serialize(tgx_state *st, size_t *len, uint8_t *buf, size_t cap, tgx_error *err);
tgx_status regex_deserialize(tgx_state *st, const uint8_t *buf, size_t len, size_t *pos, tgx_error *err);
}  // namespace tgx
