// kll_device.h -- KLL sketch tasks of a state (device sketching + host merge); see kll_device.cpp.
#pragma once
#include "internal.h"

namespace tgx {
void kll_state_init(tgx_state *st);
void kll_state_free(tgx_state *st);
void kll_state_reset(tgx_state *st);
tgx_status kll_update(tgx_state *st, size_t slot, const tgx_column &col, tgx_error *err);
// the fused path (kernels/scan.hip): the task's sampler rides on the numeric scan of its column.  Batches big enough
// to be sampled are eligible; _prepare sizes the buffers the scan writes (n_waves waves, each at most
// max_rows_per_wave rows) and fills the descriptor the scan kernel gets; _finish, called once the scan has been
// queued, sketches the picks and leftovers it leaves.
bool kll_scan_eligible(int64_t rows);
tgx_status kll_scan_prepare(tgx_state *st, size_t slot, int64_t rows, int n_waves, int64_t max_rows_per_wave,
                            ScanKll *out, tgx_error *err);
tgx_status kll_scan_finish(tgx_state *st, tgx_error *err);  // every prepared task of the batch
// folds the device-side sketch of every KLL task into st->h_kll (leaves the device side empty)
tgx_status kll_flush(tgx_state *st, tgx_error *err);
tgx_status kll_fill_result(tgx_state *st, int slot, tgx_result *r, tgx_error *err);
tgx_status kll_merge_states(tgx_state *dst, tgx_state *src, tgx_error *err);
tgx_status kll_serialize(tgx_state *st, size_t *len, uint8_t *buf, size_t cap, tgx_error *err);
tgx_status kll_deserialize(tgx_state *st, const uint8_t *buf, size_t len, size_t *pos, tgx_error *err);
}  // namespace tgx
