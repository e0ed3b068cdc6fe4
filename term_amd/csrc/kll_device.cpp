// kll_device.cpp -- placeholder until kernels/kll.hip lands: KLL specs are refused at update time.
#include "kll_device.h"

#include <stdio.h>
#include <string.h>

namespace tgx {
static tgx_status kfail(tgx_error *err, tgx_status code, const char *msg) {
  if (err) {
    err->code = code;
    snprintf(err->msg, sizeof(err->msg), "%s", msg);
  }
  return code;
}
void kll_state_init(tgx_state *) {}
void kll_state_free(tgx_state *) {}
void kll_state_reset(tgx_state *) {}
tgx_status kll_update(tgx_state *, size_t, const tgx_column &, tgx_error *err) {
  return kfail(err, TGX_UNSUPPORTED, "KLL is not implemented yet");
}
tgx_status kll_flush(tgx_state *, tgx_error *) { return TGX_OK; }
tgx_status kll_fill_result(tgx_state *st, int slot, tgx_result *r, tgx_error *) {
  r->kll_n = st->h_kll[slot].n;
  return TGX_OK;
}
tgx_status kll_merge_states(tgx_state *, tgx_state *, tgx_error *) { return TGX_OK; }
tgx_status kll_serialize(tgx_state *, size_t *, uint8_t *, size_t, tgx_error *) { return TGX_OK; }
tgx_status kll_deserialize(tgx_state *, const uint8_t *, size_t, size_t *, tgx_error *) { return TGX_OK; }
}  // namespace tgx

extern "C" tgx_status tgx_kll_quantile(const tgx_plan *, tgx_state *, size_t, double, double *, tgx_error *err) {
  return tgx::kfail(err, TGX_UNSUPPORTED, "KLL is not implemented yet");
}
extern "C" tgx_status tgx_kll_summary(const tgx_plan *, tgx_state *, size_t, uint64_t *, double *, double *,
                                      uint64_t *, uint64_t *, tgx_error *err) {
  return tgx::kfail(err, TGX_UNSUPPORTED, "KLL is not implemented yet");
}
extern "C" tgx_status tgx_kll_level_items(const tgx_plan *, tgx_state *, size_t, uint64_t, double *, uint64_t,
                                          uint64_t *, tgx_error *err) {
  return tgx::kfail(err, TGX_UNSUPPORTED, "KLL is not implemented yet");
}
extern "C" double tgx_kll_relative_error_bound(uint32_t k) { return 1.65 / __builtin_sqrt((double)k); }
