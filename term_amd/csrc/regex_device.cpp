// regex_device.cpp -- REGEX_MATCH tasks: pattern validation + DFA compilation at plan time, the match
// kernel at update time (kernels/regex.hip), counts merged like any other additive state.
#include "regex_device.h"

#include <vector>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "kernels/regex_types.h"
#include "regex/regex_compile.h"

namespace tgx {

void launch_length_filter(const RegexColDesc &d, const LengthBounds &lb, unsigned long long *d_counters, int n_cu,
                          hipStream_t stream);
void launch_length(const RegexColDesc &d, const LengthBounds &lb, unsigned long long *d_counters, int n_cu,
                   hipStream_t stream);
void launch_regex(const RegexColDesc &d, const DfaView &dfa, unsigned long long *d_counters, int n_cu,
                  hipStream_t stream);

namespace {
struct RegexTask {
  int column;
  uint32_t flags;
  std::string pattern;
  rx::Dfa dfa;
  bool is_length = false;  // TGX_CHECK_LENGTH: no automaton, character-count bounds instead
  uint64_t len_min = 0, len_max = 0;
};
// patterns of one column (same TRIM flag) whose product automaton fits the LDS table: decided in ONE walk
struct RegexGroup {
  std::vector<int> members;  // task indices, <= kMaxRegexGroup
  rx::ProductDfa dfa;
};
struct RegexPlan {
  std::vector<RegexTask> tasks;
  std::vector<RegexGroup> groups;
  std::vector<int> group_of;  // per task: its group or -1
};
struct RegexTaskState {
  DevBuf table, byte_class, accept_end, counters;
  bool direct = false;  // the uploaded table is byte-indexed (256 columns)
  DevBuf dict_hits;  // Dictionary columns: one hit byte per dictionary entry of the current batch
  uint64_t h_total = 0, h_matches = 0;  // merged-in / deserialized contributions
  uint64_t total = 0;                   // rows handled on this device
};
struct RegexGroupState {
  DevBuf table, byte_class, accept_end;
  bool direct = false, uploaded = false;
};
struct RegexState {
  std::vector<RegexTaskState> tasks;
  std::vector<RegexGroupState> groups;
  DevBuf counter_pool;  // [task][2]: every RegexTaskState::counters is a slice, so results come back in ONE copy
  std::vector<unsigned long long> fetched;  // host copy of the pool, valid while `fetched_ok` (inside one API call)
  bool fetched_ok = false;
};

tgx_status rfail(tgx_error *err, tgx_status code, const char *fmt, ...) {
  if (err) {
    err->code = code;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err->msg, sizeof(err->msg), fmt, ap);
    va_end(ap);
  }
  return code;
}
#define RHIP(expr)                                                                                         \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess)                                                                                  \
      return rfail(err, e_ == hipErrorOutOfMemory ? TGX_OUT_OF_MEMORY : TGX_DEVICE_ERROR, "%s failed: %s", \
                   #expr, hipGetErrorString(e_));                                                          \
  } while (0)

#define TGX_TRY_R(expr)          \
  do {                           \
    tgx_status s_ = (expr);      \
    if (s_ != TGX_OK) return s_; \
  } while (0)

const RegexPlan *rplan(const tgx_plan *p) { return (const RegexPlan *)p->regex; }
RegexState *rstate(tgx_state *s) { return (RegexState *)s->regex; }

tgx_status compile_checked(const char *pattern, size_t len, uint32_t flags, rx::Dfa *dfa, tgx_error *err) {
  std::string msg;
  rx::CompileStatus rs = rx::validate_pattern_rules(pattern, len, &msg);
  if (rs != rx::kOk) return rfail(err, TGX_INVALID_ARGUMENT, "%s", msg.c_str());
  // a pattern is compiled by validate, by plan_create and by every is_match / match_group of a host-side check: the
  // last few automata of the thread are kept (a pattern with Unicode word boundaries takes 15-100 ms to build)
  struct Cached {
    std::string pattern;
    bool fold;
    rx::Dfa dfa;
  };
  static thread_local std::vector<Cached> cache;
  const bool fold = (flags & TGX_FLAG_CASE_INSENSITIVE) != 0;
  for (size_t i = 0; i < cache.size(); i++)
    if (cache[i].fold == fold && cache[i].pattern.size() == len && memcmp(cache[i].pattern.data(), pattern, len) == 0) {
      *dfa = cache[i].dfa;
      return TGX_OK;
    }
  rs = rx::compile(pattern, len, fold, dfa, &msg);
  if (rs == rx::kInvalid) return rfail(err, TGX_INVALID_ARGUMENT, "%s", msg.c_str());
  if (rs != rx::kOk) return rfail(err, TGX_UNSUPPORTED, "%s", msg.c_str());
  if (cache.size() >= 8) cache.erase(cache.begin());
  cache.push_back({std::string(pattern, len), fold, *dfa});
  return TGX_OK;
}
}  // namespace

tgx_status regex_plan_add(tgx_plan *plan, int spec_index, int *slot, tgx_error *err) {
  if (!plan->regex) plan->regex = new RegexPlan();
  RegexPlan *rp = (RegexPlan *)plan->regex;
  const tgx_check_spec &sp = plan->specs[spec_index];
  if (sp.kind == TGX_CHECK_LENGTH) {
    if (sp.length_min > sp.length_max) return rfail(err, TGX_INVALID_ARGUMENT, "LENGTH: length_min > length_max");
    for (size_t i = 0; i < rp->tasks.size(); i++)
      if (rp->tasks[i].is_length && rp->tasks[i].column == sp.column && rp->tasks[i].len_min == sp.length_min &&
          rp->tasks[i].len_max == sp.length_max) {
        *slot = (int)i;
        return TGX_OK;
      }
    RegexTask t;
    t.column = sp.column;
    t.flags = TGX_FLAG_NULL_IS_VALID;  // "OR col IS NULL" is part of the check (length.rs:169)
    t.is_length = true;
    t.len_min = sp.length_min;
    t.len_max = sp.length_max;
    rp->tasks.push_back(std::move(t));
    *slot = (int)rp->tasks.size() - 1;
    return TGX_OK;
  }
  const std::string &pat = plan->patterns[spec_index];
  const uint32_t flags = sp.flags & (TGX_FLAG_TRIM | TGX_FLAG_CASE_INSENSITIVE | TGX_FLAG_NULL_IS_VALID);
  for (size_t i = 0; i < rp->tasks.size(); i++)
    if (!rp->tasks[i].is_length && rp->tasks[i].column == sp.column && rp->tasks[i].flags == flags &&
        rp->tasks[i].pattern == pat) {
      *slot = (int)i;
      return TGX_OK;
    }
  RegexTask t;
  t.column = sp.column;
  t.flags = flags;
  t.pattern = pat;
  tgx_status s = compile_checked(pat.data(), pat.size(), flags, &t.dfa, err);
  if (s != TGX_OK) return s;
  rp->tasks.push_back(std::move(t));
  *slot = (int)rp->tasks.size() - 1;
  return TGX_OK;
}

// Called once every spec has been added: patterns of the same column and TRIM flag are grouped while the product of
// their automata still fits the LDS transition table (k format checks on a column then cost one walk over its bytes;
// TG/constraints/format.rs:750-776 issues one query per constraint).  Greedy, in task order.
void regex_plan_finish(tgx_plan *plan) {
  if (!plan->regex) return;
  RegexPlan *rp = (RegexPlan *)plan->regex;
  rp->groups.clear();
  rp->group_of.assign(rp->tasks.size(), -1);
  for (size_t i = 0; i < rp->tasks.size(); i++) {
    if (rp->tasks[i].is_length || rp->group_of[i] >= 0) continue;
    RegexGroup g;
    g.members.push_back((int)i);
    std::vector<const rx::Dfa *> parts = {&rp->tasks[i].dfa};
    for (size_t j = i + 1; j < rp->tasks.size() && g.members.size() < (size_t)kMaxRegexGroup; j++) {
      const RegexTask &a = rp->tasks[i], &b = rp->tasks[j];
      if (b.is_length || rp->group_of[j] >= 0 || b.column != a.column ||
          ((a.flags ^ b.flags) & TGX_FLAG_TRIM) != 0)
        continue;
      parts.push_back(&b.dfa);
      rx::ProductDfa prod;
      if (rx::dfa_product(parts, kRegexLdsEntries, &prod)) {
        g.dfa = std::move(prod);
        g.members.push_back((int)j);
      } else {
        parts.pop_back();
      }
    }
    if (g.members.size() < 2) continue;
    for (int m : g.members) rp->group_of[m] = (int)rp->groups.size();
    rp->groups.push_back(std::move(g));
  }
}

void regex_plan_free(tgx_plan *plan) {
  delete (RegexPlan *)plan->regex;
  plan->regex = nullptr;
}

size_t regex_num_tasks(const tgx_plan *plan) { return plan->regex ? rplan(plan)->tasks.size() : 0; }

void regex_mark_used(const tgx_plan *plan, std::vector<char> &used) {
  if (!plan->regex) return;
  for (auto &t : rplan(plan)->tasks) used[t.column] = 1;
}

void regex_state_init(tgx_state *st) {
  if (st->regex) return;
  RegexState *rs = new RegexState();
  rs->tasks.resize(regex_num_tasks(st->plan));
  rs->groups.resize(st->plan->regex ? rplan(st->plan)->groups.size() : 0);
  st->regex = rs;
}

void regex_state_free(tgx_state *st) {
  delete rstate(st);
  st->regex = nullptr;
}

void regex_state_reset(tgx_state *st) {
  RegexState *rs = rstate(st);
  if (!rs) return;
  for (auto &t : rs->tasks) {
    t.h_total = t.h_matches = t.total = 0;
  }
  if (rs->counter_pool.p) (void)hipMemsetAsync(rs->counter_pool.p, 0, rs->tasks.size() * 16, st->stream);
  rs->fetched_ok = false;
}

// the string layout one launch walks: the column itself, or the dictionary of a dictionary column (matched once per
// ENTRY; the rows then only gather the per-entry verdicts)
static void fill_col_desc(const tgx_column &c, uint32_t flags, RegexColDesc *d) {
  const bool is_dict = c.type == TGX_DICT32_UTF8, is_view = c.type == TGX_UTF8_VIEW;
  const tgx_column &sc = is_dict ? *c.dictionary : c;
  memset(d, 0, sizeof(*d));
  d->offsets = sc.offsets;
  d->data = sc.data;
  d->validity = sc.validity;
  d->offset = sc.offset;
  d->length = sc.length;
  d->large_offsets = sc.type == TGX_LARGE_UTF8;
  d->views = is_view ? c.values : nullptr;
  d->buffers = is_view ? c.variadic : nullptr;
  d->trim = (flags & TGX_FLAG_TRIM) != 0;
  d->null_is_valid = (flags & TGX_FLAG_NULL_IS_VALID) != 0;
}

// uploads an automaton (byte-indexed when tiny: one LDS lookup per input byte instead of class + transition -- only
// while the table stays small (8 KiB): the kernel hides its load latency with resident workgroups (7 per CU at
// <= 22 KiB of LDS each), and a byte-indexed 32 KiB table ran 1.6x SLOWER)
static tgx_status upload_dfa(uint32_t n_states, uint32_t n_classes, const uint8_t *byte_class,
                             const std::vector<uint16_t> &src, const std::vector<uint8_t> &accept, DevBuf &d_table,
                             DevBuf &d_class, DevBuf &d_accept, bool *direct, tgx_error *err) {
  *direct = (uint64_t)n_states * 256 <= 4096;
  std::vector<uint16_t> table = src;
  uint8_t cls[256];
  memcpy(cls, byte_class, 256);
  if (*direct) {
    table.assign((size_t)n_states * 256, 0);
    for (uint32_t s2 = 0; s2 < n_states; s2++)
      for (uint32_t b = 0; b < 256; b++) table[(size_t)s2 * 256 + b] = src[(size_t)s2 * n_classes + byte_class[b]];
    for (uint32_t b = 0; b < 256; b++) cls[b] = (uint8_t)b;
  }
  RHIP(d_table.reserve(table.size() * sizeof(uint16_t) + 16));
  RHIP(d_class.reserve(256));
  RHIP(d_accept.reserve(accept.size() + 16));
  RHIP(hipMemcpy(d_table.p, table.data(), table.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  RHIP(hipMemcpy(d_class.p, cls, 256, hipMemcpyHostToDevice));
  RHIP(hipMemcpy(d_accept.p, accept.data(), accept.size(), hipMemcpyHostToDevice));
  return TGX_OK;
}

tgx_status regex_update(tgx_state *st, const tgx_column *dev, tgx_error *err, DictFuse *fuse) {
  if (!st->plan->regex) return TGX_OK;
  const RegexPlan *rp = rplan(st->plan);
  RegexState *rs = rstate(st);
  rs->fetched_ok = false;
  const int n_cu = tgx_num_cus();
  struct Timer {  // one "regex" profile entry per launch
    tgx_state *st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    explicit Timer(tgx_state *s) : st(s) {
      if (st->profiling && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess)
        (void)hipEventRecord(e0, st->stream);
    }
    ~Timer() {
      if (st->profiling && e0 && e1) {
        (void)hipEventRecord(e1, st->stream);
        ProfileEntry &pe = st->profile["regex"];
        pe.pending.emplace_back(e0, e1);
        pe.pending_bytes.push_back(0);  // value bytes are data dependent; bench.py prices them itself
      }
    }
  };
  for (size_t i = 0; i < rp->tasks.size(); i++) {
    const RegexTask &t = rp->tasks[i];
    const tgx_column &c = dev[t.column];
    if (c.type != TGX_UTF8 && c.type != TGX_LARGE_UTF8 && c.type != TGX_DICT32_UTF8 && c.type != TGX_UTF8_VIEW)
      return rfail(err, TGX_UNSUPPORTED, "%s needs a Utf8 column (column %d has type %d)",
                   t.is_length ? "LENGTH" : "REGEX_MATCH", t.column, c.type);
  }
  if (!rs->counter_pool.p && !rs->tasks.empty()) {
    RHIP(rs->counter_pool.reserve(rs->tasks.size() * 16));
    // zero-filled on the state's (non-blocking) stream: ordered with the kernels that add to the counters
    RHIP(hipMemsetAsync(rs->counter_pool.p, 0, rs->tasks.size() * 16, st->stream));
    for (size_t k = 0; k < rs->tasks.size(); k++)
      rs->tasks[k].counters.borrow((char *)rs->counter_pool.p + 16 * k, 16);
  }
  std::vector<char> walked(rp->tasks.size(), 0);
  // ---- groups: several patterns of a column in one walk (product automaton)
  for (size_t g = 0; g < rp->groups.size(); g++) {
    const RegexGroup &grp = rp->groups[g];
    RegexGroupState &gs = rs->groups[g];
    const tgx_column &c = dev[rp->tasks[grp.members[0]].column];
    const bool is_dict = c.type == TGX_DICT32_UTF8;
    const tgx_column &sc = is_dict ? *c.dictionary : c;
    if (c.length == 0 || sc.length == 0) continue;
    if (!gs.uploaded) {
      TGX_TRY_R(upload_dfa(grp.dfa.n_states, grp.dfa.n_classes, grp.dfa.byte_class, grp.dfa.table, grp.dfa.accept_mask,
                           gs.table, gs.byte_class, gs.accept_end, &gs.direct, err));
      gs.uploaded = true;
    }
    RegexColDesc d;
    fill_col_desc(c, rp->tasks[grp.members[0]].flags, &d);
    d.n_pat = (int32_t)grp.members.size();
    for (size_t k = 0; k < grp.members.size(); k++) {
      const RegexTask &t = rp->tasks[grp.members[k]];
      RegexTaskState &ts = rs->tasks[grp.members[k]];
      if (t.flags & TGX_FLAG_NULL_IS_VALID) d.null_mask |= 1u << k;
      if (is_dict) RHIP(ts.dict_hits.reserve((size_t)sc.length + 32));
      d.hits_k[k] = is_dict ? ts.dict_hits.as<uint8_t>() : nullptr;
      // dictionary columns: counters[1] soaks up the per-entry match count, counters[0] receives the per-row one
      d.counters_k[k] = ts.counters.as<unsigned long long>() + (is_dict ? 1 : 0);
      walked[grp.members[k]] = 1;
    }
    DfaView v;
    v.table = gs.table.as<uint16_t>();
    v.byte_class = gs.byte_class.as<uint8_t>();
    v.accept_end = gs.accept_end.as<uint8_t>();
    v.n_states = grp.dfa.n_states;
    v.n_classes = gs.direct ? 256 : grp.dfa.n_classes;
    v.start = grp.dfa.start;
    v.direct = gs.direct ? 1 : 0;
    v.n_final = grp.dfa.n_final;
    Timer timer(st);
    launch_regex(d, v, nullptr, n_cu, st->stream);
  }
  // ---- single patterns, LENGTH checks, and the per-row gathers of dictionary columns
  for (size_t i = 0; i < rp->tasks.size(); i++) {
    const RegexTask &t = rp->tasks[i];
    RegexTaskState &ts = rs->tasks[i];
    const tgx_column &c = dev[t.column];
    const bool is_dict = c.type == TGX_DICT32_UTF8;
    if (!t.is_length && !ts.table.p && !walked[i])
      TGX_TRY_R(upload_dfa(t.dfa.n_states, t.dfa.n_classes, t.dfa.byte_class, t.dfa.table, t.dfa.accept_at_end, ts.table,
                           ts.byte_class, ts.accept_end, &ts.direct, err));
    ts.total += (uint64_t)c.length;
    if (c.length == 0) continue;
    const tgx_column &sc = is_dict ? *c.dictionary : c;
    if (is_dict) RHIP(ts.dict_hits.reserve((size_t)sc.length + 32));
    RegexColDesc d;
    fill_col_desc(c, t.flags, &d);
    d.hits = is_dict ? ts.dict_hits.as<uint8_t>() : nullptr;
    DfaView v;
    v.table = ts.table.as<uint16_t>();
    v.byte_class = ts.byte_class.as<uint8_t>();
    v.accept_end = ts.accept_end.as<uint8_t>();
    v.n_states = t.dfa.n_states;
    v.n_classes = ts.direct ? 256 : t.dfa.n_classes;
    v.start = t.dfa.start;
    v.direct = ts.direct ? 1 : 0;
    v.n_final = 2;
    Timer timer(st);
    const LengthBounds lb{t.len_min, t.len_max};
    // an automaton with a character count (`^C{m,n}$`, regex_compile.h): the walk leaves a byte per row, a second pass
    // takes the rows whose length is outside the bounds out of the matches and counts
    const bool bounded = !t.is_length && t.dfa.len_max >= 0;
    const LengthBounds lbb{(uint64_t)std::max<int64_t>(t.dfa.len_min, 0), (uint64_t)std::max<int64_t>(t.dfa.len_max, 0)};
    if (!is_dict) {
      if (t.is_length) {
        launch_length(d, lb, ts.counters.as<unsigned long long>(), n_cu, st->stream);
      } else if (!walked[i] && bounded) {
        RHIP(ts.dict_hits.reserve((size_t)c.length + 32));
        d.hits = ts.dict_hits.as<uint8_t>();
        launch_regex(d, v, ts.counters.as<unsigned long long>() + 1, n_cu, st->stream);  // (counters[1]: scratch)
        launch_length_filter(d, lbb, ts.counters.as<unsigned long long>(), n_cu, st->stream);
      } else if (!walked[i]) {
        launch_regex(d, v, ts.counters.as<unsigned long long>(), n_cu, st->stream);
      }
    } else {
      // counters[1] soaks up the per-entry match count, counters[0] receives the per-row one
      if (sc.length > 0) {
        if (t.is_length)
          launch_length(d, lb, ts.counters.as<unsigned long long>() + 1, n_cu, st->stream);
        else if (!walked[i])
          launch_regex(d, v, ts.counters.as<unsigned long long>() + 1, n_cu, st->stream);
        if (!t.is_length && !walked[i] && bounded)  // (the entries' verdicts, before the rows gather them)
          launch_length_filter(d, lbb, ts.counters.as<unsigned long long>() + 1, n_cu, st->stream);
      }
      bool fused = false;
      if (fuse) {
        auto cap = fuse->capacity.find(t.column);
        if (cap != fuse->capacity.end() && (int)fuse->by_column[t.column].size() < cap->second) {
          // the caller reads the indices once for this column: DISTINCT usage + this gather in one pass
          fuse->by_column[t.column].push_back({ts.dict_hits.as<uint8_t>(), ts.counters.as<unsigned long long>(),
                                               (t.flags & TGX_FLAG_NULL_IS_VALID) != 0});
          fused = true;
        }
      }
      if (!fused)
        launch_dict_count_hits((const int32_t *)c.values, c.validity, c.offset, c.length, sc.length,
                               sc.validity != nullptr, ts.dict_hits.as<uint8_t>(),
                               (t.flags & TGX_FLAG_NULL_IS_VALID) != 0, ts.counters.as<unsigned long long>(), n_cu,
                               st->stream);
    }
  }
  return TGX_OK;
}

// One copy of every task's counters; regex_totals() then answers from the host copy until regex_fetch_end().
// tgx_finalize / serialize / merge bracket their per-task loops with these (33 synchronous 8-byte copies were
// 1 ms of a 31 ms step).
tgx_status regex_fetch_begin(tgx_state *st, tgx_error *err) {
  RegexState *rs = rstate(st);
  if (!rs || !rs->counter_pool.p) return TGX_OK;
  rs->fetched.resize(rs->tasks.size() * 2);
  RHIP(hipMemcpyAsync(rs->fetched.data(), rs->counter_pool.p, rs->tasks.size() * 16, hipMemcpyDeviceToHost,
                      st->stream));  // ordered on the state's stream: no wait for the rest of the device
  RHIP(hipStreamSynchronize(st->stream));
  rs->fetched_ok = true;
  return TGX_OK;
}

void regex_fetch_end(tgx_state *st) {
  if (rstate(st)) rstate(st)->fetched_ok = false;
}

static tgx_status regex_totals(tgx_state *st, size_t i, uint64_t *total, uint64_t *matches, tgx_error *err) {
  RegexState *rs = rstate(st);
  RegexTaskState &ts = rs->tasks[i];
  unsigned long long dev_matches = 0;
  if (rs->fetched_ok) {
    dev_matches = rs->fetched[2 * i];
  } else if (ts.counters.p) {
    RHIP(hipMemcpyAsync(&dev_matches, ts.counters.p, 8, hipMemcpyDeviceToHost, st->stream));
    RHIP(hipStreamSynchronize(st->stream));
  }
  *total = ts.total + ts.h_total;
  *matches = dev_matches + ts.h_matches;
  return TGX_OK;
}

tgx_status regex_fill_result(tgx_state *st, int slot, tgx_result *r, tgx_error *err) {
  uint64_t total = 0, matches = 0;
  tgx_status s = regex_totals(st, (size_t)slot, &total, &matches, err);
  if (s != TGX_OK) return s;
  r->total = (int64_t)total;
  r->matches = (int64_t)matches;
  return TGX_OK;
}

tgx_status regex_merge_states(tgx_state *dst, tgx_state *src, tgx_error *err) {
  if (!dst->regex) return TGX_OK;
  tgx_status fs = regex_fetch_begin(src, err);
  if (fs != TGX_OK) return fs;
  for (size_t i = 0; i < rstate(dst)->tasks.size(); i++) {
    uint64_t total = 0, matches = 0;
    tgx_status s = regex_totals(src, i, &total, &matches, err);
    if (s != TGX_OK) {
      regex_fetch_end(src);
      return s;
    }
    rstate(dst)->tasks[i].h_total += total;
    rstate(dst)->tasks[i].h_matches += matches;
  }
  regex_fetch_end(src);
  return TGX_OK;
}

tgx_status regex_serialize(tgx_state *st, size_t *len, uint8_t *buf, size_t cap, tgx_error *err) {
  if (!st->regex) return TGX_OK;
  tgx_status fs = regex_fetch_begin(st, err);
  if (fs != TGX_OK) return fs;
  for (size_t i = 0; i < rstate(st)->tasks.size(); i++) {
    uint64_t v[2] = {0, 0};
    tgx_status s = regex_totals(st, i, &v[0], &v[1], err);
    if (s != TGX_OK) {
      regex_fetch_end(st);
      return s;
    }
    if (buf && *len + 16 <= cap) memcpy(buf + *len, v, 16);
    *len += 16;
  }
  regex_fetch_end(st);
  return TGX_OK;
}

tgx_status regex_deserialize(tgx_state *st, const uint8_t *buf, size_t len, size_t *pos, tgx_error *err) {
  if (!st->regex) return TGX_OK;
  for (auto &t : rstate(st)->tasks) {
    if (*pos + 16 > len) return rfail(err, TGX_INVALID_ARGUMENT, "truncated state blob (regex)");
    uint64_t v[2];
    memcpy(v, buf + *pos, 16);
    *pos += 16;
    t.h_total = v[0];
    t.h_matches = v[1];
  }
  return TGX_OK;
}

}  // namespace tgx

using namespace tgx;

extern "C" tgx_status tgx_regex_validate(const char *pattern, size_t len, uint32_t flags, tgx_error *err) try {
  if (!pattern && len) return rfail(err, TGX_INVALID_ARGUMENT, "pattern is NULL");
  rx::Dfa dfa;
  return compile_checked(pattern ? pattern : "", len, flags, &dfa, err);
} catch (...) {
  return tgx::abi_exception(err);
}

// Host-side walk of the compiled automaton for ONE value: a compile check for the caller (does this pattern
// mean what I think), never used by tgx_update.
extern "C" tgx_status tgx_regex_is_match(const char *pattern, size_t plen, uint32_t flags, const uint8_t *value,
                                         size_t vlen, int32_t *matched, tgx_error *err) try {
  if ((!pattern && plen) || (!value && vlen) || !matched) return rfail(err, TGX_INVALID_ARGUMENT, "NULL argument");
  rx::Dfa dfa;
  tgx_status s = compile_checked(pattern ? pattern : "", plen, flags, &dfa, err);
  if (s != TGX_OK) return s;
  size_t b = 0, e = vlen;
  if (flags & TGX_FLAG_TRIM) {
    while (b < e && value[b] == 0x20) b++;
    while (e > b && value[e - 1] == 0x20) e--;
  }
  *matched = rx::dfa_is_match(dfa, value + b, e - b) ? 1 : 0;
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}

extern "C" tgx_status tgx_regex_match_group(const char *const *patterns, const size_t *pattern_lens,
                                            const uint32_t *flags, size_t n_patterns, const uint8_t *value, size_t vlen,
                                            uint32_t *mask, int32_t *grouped, tgx_error *err) try {
  if (!patterns || !pattern_lens || !mask || (!value && vlen) || n_patterns == 0 || n_patterns > (size_t)kMaxRegexGroup)
    return rfail(err, TGX_INVALID_ARGUMENT, "bad arguments (1..%d patterns)", kMaxRegexGroup);
  std::vector<rx::Dfa> dfas(n_patterns);
  std::vector<const rx::Dfa *> parts;
  bool same_trim = true;
  for (size_t k = 0; k < n_patterns; k++) {
    const uint32_t f = flags ? flags[k] : 0;
    tgx_status s = compile_checked(patterns[k] ? patterns[k] : "", pattern_lens[k], f, &dfas[k], err);
    if (s != TGX_OK) return s;
    parts.push_back(&dfas[k]);
    same_trim &= ((f ^ (flags ? flags[0] : 0)) & TGX_FLAG_TRIM) == 0;
  }
  auto trimmed = [&](uint32_t f, size_t *b, size_t *e) {
    *b = 0;
    *e = vlen;
    if (f & TGX_FLAG_TRIM) {
      while (*b < *e && value[*b] == 0x20) (*b)++;
      while (*e > *b && value[*e - 1] == 0x20) (*e)--;
    }
  };
  rx::ProductDfa prod;
  const bool ok = same_trim && rx::dfa_product(parts, kRegexLdsEntries, &prod);
  if (grouped) *grouped = ok ? 1 : 0;
  size_t b, e;
  if (ok) {
    trimmed(flags ? flags[0] : 0, &b, &e);
    *mask = rx::product_match_mask(prod, value + b, e - b);
    return TGX_OK;
  }
  *mask = 0;
  for (size_t k = 0; k < n_patterns; k++) {
    trimmed(flags ? flags[k] : 0, &b, &e);
    if (rx::dfa_is_match(dfas[k], value + b, e - b)) *mask |= 1u << k;
  }
  return TGX_OK;
} catch (...) {
  return tgx::abi_exception(err);
}
