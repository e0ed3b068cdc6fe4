// kll_device.h -- KLL sketch tasks of a state (device sketching + host merge); see kll_device.cpp.
#pragma once
#include "internal.h"

namespace tgx {
void kll_state_init(tgx_state *st);
void kll_state_free(tgx_state *st);
void kll_state_reset(tgx_state *st);
tgx_status kll_update(tgx_state *st, size_t slot, const tgx_column &col, tgx_error *err);
// folds the device-side sketch of every KLL task into st->h_kll (leaves the device side empty)
tgx_status kll_flush(tgx_state *st, tgx_error *err);
tgx_status kll_fill_result(tgx_state *st, int slot, tgx_result *r, tgx_error *err);
tgx_status kll_merge_states(tgx_state *dst, tgx_state *src, tgx_error *err);
tgx_status kll_serialize(tgx_state *st, size_t *len, uint8_t *buf, size_t cap, tgx_error *err);
tgx_status kll_deserialize(tgx_state *st, const uint8_t *buf, size_t len, size_t *pos, tgx_error *err);
}  // namespace tgx
