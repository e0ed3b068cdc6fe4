// regex_device.cpp -- placeholder until the regex engine lands: REGEX specs are refused at plan time.
#include "regex_device.h"

#include <stdio.h>

namespace tgx {
static tgx_status rfail(tgx_error *err, tgx_status code, const char *msg) {
  if (err) {
    err->code = code;
    snprintf(err->msg, sizeof(err->msg), "%s", msg);
  }
  return code;
}
tgx_status regex_plan_add(tgx_plan *, int, int *, tgx_error *err) {
  return rfail(err, TGX_UNSUPPORTED, "REGEX_MATCH is not implemented yet");
}
void regex_plan_free(tgx_plan *) {}
size_t regex_num_tasks(const tgx_plan *) { return 0; }
void regex_mark_used(const tgx_plan *, std::vector<char> &) {}
void regex_state_init(tgx_state *) {}
void regex_state_free(tgx_state *) {}
void regex_state_reset(tgx_state *) {}
tgx_status regex_update(tgx_state *, const tgx_column *, tgx_error *) { return TGX_OK; }
tgx_status regex_fill_result(tgx_state *, int, tgx_result *, tgx_error *) { return TGX_OK; }
tgx_status regex_merge_states(tgx_state *, tgx_state *, tgx_error *) { return TGX_OK; }
tgx_status regex_serialize(tgx_state *, size_t *, uint8_t *, size_t, tgx_error *) { return TGX_OK; }
tgx_status regex_deserialize(tgx_state *, const uint8_t *, size_t, size_t *, tgx_error *) { return TGX_OK; }
}  // namespace tgx

extern "C" tgx_status tgx_regex_validate(const char *, size_t, uint32_t, tgx_error *err) {
  return tgx::rfail(err, TGX_UNSUPPORTED, "REGEX_MATCH is not implemented yet");
}
extern "C" tgx_status tgx_regex_is_match(const char *, size_t, uint32_t, const uint8_t *, size_t, int32_t *,
                                         tgx_error *err) {
  return tgx::rfail(err, TGX_UNSUPPORTED, "REGEX_MATCH is not implemented yet");
}
