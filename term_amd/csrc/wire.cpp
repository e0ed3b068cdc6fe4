// wire.cpp -- tgx_state_serialize / tgx_state_deserialize (blobs v3; term_amd/wire.py documents the layout).
// Split off tgx_api.cpp in round 4.
#include "api_internal.h"

// ------------------------------------------------------------------------------------------------
// wire form
namespace {
struct Writer {
  uint8_t *buf;
  size_t cap, len = 0;
  void put(const void *p, size_t n) {
    if (buf && len + n <= cap) memcpy(buf + len, p, n);
    len += n;
  }
  template <class T>
  void pod(const T &v) { put(&v, sizeof(T)); }
};
struct Reader {
  const uint8_t *buf;
  size_t len, pos = 0;
  bool ok = true;
  void get(void *p, size_t n) {
    if (pos + n > len) {
      ok = false;
      memset(p, 0, n);
      return;
    }
    memcpy(p, buf + pos, n);
    pos += n;
  }
  template <class T>
  T pod() {
    T v;
    get(&v, sizeof(T));
    return v;
  }
};
}  // namespace

extern "C" tgx_status tgx_state_serialize(const tgx_plan *plan, tgx_state *st, uint8_t *buf, size_t cap,
                                          size_t *len, tgx_error *err) try {
  bind_thread();
  if (!plan || !st || st->plan != plan || !len) return fail(err, TGX_INVALID_ARGUMENT, "bad arguments");
  TGX_TRY(spearman_check_mergeable(st, err));
  Gathered g;
  TGX_TRY(gather(st, &g, err));
  Writer w{buf, cap};
  w.pod(kWireMagic);
  w.pod(kWireVersion);
  w.pod((uint32_t)g.scan.size());
  w.pod((uint32_t)g.count.size());
  w.pod((uint32_t)g.como.size());
  w.pod((uint32_t)g.distinct.size());
  w.pod((uint32_t)plan->kll.size());
  w.pod((uint32_t)regex_num_tasks(plan));
  w.pod((uint32_t)plan->hll.size());
  // the fingerprint key, when the blob holds fingerprints (a string / tuple key set that travels with its records)
  uint32_t keyed = 0;
  for (size_t k = 0; k < g.distinct.size(); k++) {
    const DistinctState &ds = st->distinct[k];
    const bool has_set = ds.mode == DistinctMode::kBitmap || ds.mode == DistinctMode::kHash;
    if (has_set && !ds.partitioned && ds.wide) keyed = 1;
  }
  w.pod(keyed);
  uint32_t key_words[4] = {0, 0, 0, 0};
  if (keyed) memcpy(key_words, plan->fp_key.k, 16);
  w.put(key_words, 16);
  for (auto &a : g.scan) w.pod(a);
  for (auto &a : g.count) w.pod(a);
  for (auto &a : g.como) w.pod(a);
  for (size_t k = 0; k < g.distinct.size(); k++) {
    DistinctState &ds = st->distinct[k];
    const DistinctTotals &t = g.distinct[k];
    const bool has_set = ds.mode == DistinctMode::kBitmap || ds.mode == DistinctMode::kHash;
    uint32_t partitioned = (ds.partitioned || !has_set) ? 1 : 0;
    w.pod(partitioned);
    w.pod((uint32_t)(ds.wide ? 1 : 0));
    w.pod(t);
    uint64_t n_records = 0;
    if (!partitioned) {
      // non-partitioned sets travel with their keys so the receiver can take an exact union
      const void *recs = nullptr;
      TGX_TRY(distinct_export_impl(st, k, 1, &recs, &n_records, err));
      w.pod(n_records);
      size_t bytes = (size_t)n_records * (ds.wide ? sizeof(KeyRecord128) : sizeof(KeyRecord));
      if (w.buf && w.len + bytes <= w.cap)
        HIP_TRY(hipMemcpy(w.buf + w.len, recs, bytes, hipMemcpyDeviceToHost));
      w.len += bytes;
    } else {
      w.pod(n_records);
    }
  }
  TGX_TRY(kll_serialize(st, &w.len, w.buf, w.cap, err));
  TGX_TRY(regex_serialize(st, &w.len, w.buf, w.cap, err));
  for (size_t k = 0; k < plan->hll.size(); k++) {  // { u32 mode, u32 has_registers; registers }
    w.pod((uint32_t)st->hll_mode[k]);
    w.pod((uint32_t)(g.hll[k].empty() ? 0 : 1));
    if (!g.hll[k].empty()) w.put(g.hll[k].data(), kHllRegisters);
  }
  *len = w.len;
  if (buf && w.len > cap) return fail(err, TGX_INVALID_ARGUMENT, "buffer too small: need %zu bytes", w.len);
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

extern "C" tgx_status tgx_blob_fingerprint_key(const uint8_t *buf, size_t len, uint8_t key_out[16], int32_t *keyed) {
  if (!buf || !key_out || !keyed) return TGX_INVALID_ARGUMENT;
  Reader r{buf, len};
  if (r.pod<uint32_t>() != kWireMagic || r.pod<uint32_t>() != kWireVersion) return TGX_INVALID_ARGUMENT;
  for (int i = 0; i < 7; i++) (void)r.pod<uint32_t>();
  const uint32_t k = r.pod<uint32_t>();
  r.get(key_out, 16);
  if (!r.ok || k > 1) return TGX_INVALID_ARGUMENT;
  *keyed = (int32_t)k;
  return TGX_OK;
}

extern "C" tgx_status tgx_state_deserialize(const tgx_plan *plan, const uint8_t *buf, size_t len,
                                            tgx_state **out, tgx_error *err) try {
  bind_thread();
  if (!plan || !buf || !out) return fail(err, TGX_INVALID_ARGUMENT, "bad arguments");
  *out = nullptr;
  Reader r{buf, len};
  if (r.pod<uint32_t>() != kWireMagic) return fail(err, TGX_INVALID_ARGUMENT, "not a tgx state blob");
  if (r.pod<uint32_t>() != kWireVersion) return fail(err, TGX_INVALID_ARGUMENT, "state blob version mismatch");
  uint32_t n_scan = r.pod<uint32_t>(), n_count = r.pod<uint32_t>(), n_como = r.pod<uint32_t>(),
           n_dist = r.pod<uint32_t>(), n_kll = r.pod<uint32_t>(), n_regex = r.pod<uint32_t>(), n_hll = r.pod<uint32_t>();
  if (n_scan != plan->scan.size() || n_count != plan->count.size() || n_como != plan->como.size() ||
      n_dist != plan->distinct.size() || n_kll != plan->kll.size() || n_regex != regex_num_tasks(plan) ||
      n_hll != plan->hll.size())
    return fail(err, TGX_INVALID_ARGUMENT, "state blob was produced by a different plan");
  {
    const uint32_t keyed = r.pod<uint32_t>();
    uint8_t key[16];
    r.get(key, 16);
    if (!r.ok) return fail(err, TGX_INVALID_ARGUMENT, "truncated state blob");
    if (keyed > 1) return fail(err, TGX_INVALID_ARGUMENT, "malformed state blob (key field)");
    if (keyed && memcmp(key, plan->fp_key.k, 16) != 0) {
      char hex[33];
      for (int i = 0; i < 16; i++) snprintf(hex + 2 * i, 3, "%02x", key[i]);
      return fail(err, TGX_INVALID_ARGUMENT,
                  "the blob's string keys were made under fingerprint key %s and the plan holds another: give the plan "
                  "that key (tgx_plan_set_fingerprint_key) before its first state", hex);
    }
  }
  std::unique_ptr<tgx_state, void (*)(tgx_state *)> st(new tgx_state(), tgx_state_destroy);
  state_init_host(st.get(), plan);
  for (auto &a : st->h_scan) a = r.pod<ScanAcc>();
  for (auto &a : st->h_count) a = r.pod<CountAcc>();
  for (auto &a : st->h_como) a = r.pod<ComomentAcc>();
  for (size_t k = 0; k < n_dist; k++) {
    DistinctState &ds = st->distinct[k];
    uint32_t partitioned = r.pod<uint32_t>();
    const bool wide = r.pod<uint32_t>() != 0;
    DistinctTotals t = r.pod<DistinctTotals>();
    uint64_t n_records = r.pod<uint64_t>();
    if (!r.ok) break;
    if (partitioned) {
      ds.partitioned = true;
      ds.h_total = t.total;
      ds.h_non_null = t.non_null;
      ds.h_distinct = t.distinct - (t.empty_rows > 0 ? 1 : 0);
      ds.h_twice = t.twice - (t.empty_rows > 1 ? 1 : 0);
      ds.h_empty_rows = t.empty_rows;
    } else {
      // rebuild the key set on the device from the records.  n_records comes from the blob: bound it by the bytes
      // that are really there BEFORE multiplying (a crafted count would wrap the product past the check)
      const size_t rec_bytes = wide ? sizeof(KeyRecord128) : sizeof(KeyRecord);
      if (n_records > (r.len - r.pos) / rec_bytes) {
        r.ok = false;
        break;
      }
      size_t bytes = (size_t)n_records * rec_bytes;
      tgx_status s = need_device(err);
      if (s != TGX_OK) return s;
      s = state_init_device(st.get(), err);
      if (s != TGX_OK) return s;
      DevBuf tmp;
      HIP_TRY(tmp.reserve(std::max<size_t>(bytes, 16)));
      HIP_TRY(hipMemcpy(tmp.p, r.buf + r.pos, bytes, hipMemcpyHostToDevice));
      r.pos += bytes;
      s = distinct_import_records(st.get(), k, tmp.p, n_records, wide, err);
      if (s != TGX_OK) return s;
      HIP_TRY(hipStreamSynchronize(st->stream));
      ds.h_total = t.total;
      ds.h_non_null = t.non_null;
    }
  }
  if (r.ok) {
    tgx_status s = kll_deserialize(st.get(), r.buf, r.len, &r.pos, err);
    if (s != TGX_OK) return s;
    s = regex_deserialize(st.get(), r.buf, r.len, &r.pos, err);
    if (s != TGX_OK) return s;
    for (size_t k = 0; k < plan->hll.size() && r.ok; k++) {
      const uint32_t mode = r.pod<uint32_t>(), has = r.pod<uint32_t>();
      if (mode > 2 || has > 1) return fail(err, TGX_INVALID_ARGUMENT, "malformed state blob (APPROX_DISTINCT task)");
      st->hll_mode[k] = (int)mode;
      if (has) {
        st->h_hll[k].resize(kHllRegisters);
        r.get(st->h_hll[k].data(), kHllRegisters);
      }
    }
  }
  if (!r.ok) return fail(err, TGX_INVALID_ARGUMENT, "truncated state blob");
  *out = st.release();
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

