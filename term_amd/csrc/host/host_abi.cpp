// host_abi.cpp -- the C bridge of include/tgx_host.h over host/term_guard.{h,cpp}.
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../../include/tgx_host.h"
#include "json.h"
#include "analyzers.h"
#include "term_guard.h"

using namespace term_guard;

namespace term_guard {
void add_constraint_from_json(Check::Builder &b, const json::Value &c);
}

static tgx_status hfail(tgx_error *err, tgx_status code, const std::string &msg) {
  if (err) {
    err->code = code;
    snprintf(err->msg, sizeof(err->msg), "%s", msg.c_str());
  }
  return code;
}

static char *dup_string(const std::string &s) {
  char *p = (char *)malloc(s.size() + 1);
  if (p) memcpy(p, s.c_str(), s.size() + 1);
  return p;
}

extern "C" void tgx_host_free(char *s) { free(s); }

extern "C" tgx_status tgx_host_run_suite_json(const char *suite_json, const char *const *column_names,
                                              size_t n_columns, const tgx_column *columns, size_t n_batches,
                                              char **out_json, tgx_error *err) {
  if (!suite_json || !out_json || (n_columns && (!column_names || (n_batches && !columns))))
    return hfail(err, TGX_INVALID_ARGUMENT, "NULL argument");
  *out_json = nullptr;
  try {
    ValidationSuite suite = suite_from_json(suite_json);
    Table t;
    for (size_t i = 0; i < n_columns; i++) t.column_names.push_back(column_names[i]);
    for (size_t b = 0; b < n_batches; b++) {
      Batch batch;
      batch.columns.assign(columns + b * n_columns, columns + (b + 1) * n_columns);
      t.batches.push_back(std::move(batch));
    }
    Context ctx;
    if (n_columns > 0) ctx.register_table(suite.table_name(), std::move(t));
    ValidationResult r = suite.run(ctx);
    *out_json = dup_string(r.to_json());
    return TGX_OK;
  } catch (const TermError &e) {
    return hfail(err, TGX_INVALID_ARGUMENT, e.display());
  } catch (const std::bad_alloc &) {
    return hfail(err, TGX_OUT_OF_MEMORY, "host allocation failed (std::bad_alloc)");
  } catch (const std::exception &e) {
    return hfail(err, TGX_INTERNAL, e.what());
  } catch (...) {
    return hfail(err, TGX_INTERNAL, "unknown C++ exception");
  }
}

static std::shared_ptr<Constraint> constraint_from_json_text(const char *text) {
  json::Value v;
  std::string perr;
  if (!json::parse(text, &v, &perr)) throw TermError{TermError::Internal, "constraint JSON: " + perr};
  Check::Builder b = Check::builder("c");
  add_constraint_from_json(b, v);
  Check c = b.build();
  return c.constraints().at(0);
}

extern "C" tgx_status tgx_host_constraint_plan_json(const char *constraint_json, char **out_json, tgx_error *err) {
  if (!constraint_json || !out_json) return hfail(err, TGX_INVALID_ARGUMENT, "NULL argument");
  *out_json = nullptr;
  try {
    auto c = constraint_from_json_text(constraint_json);
    std::string o = "{\"name\": " + json::quote(c->name()) + ", \"requests\": [";
    auto reqs = c->plan();
    for (size_t i = 0; i < reqs.size(); i++) {
      const SpecRequest &r = reqs[i];
      if (i) o += ", ";
      o += "{\"kind\": " + std::to_string(r.kind) + ", \"column\": " + json::quote(r.column) + ", \"column2\": " +
           json::quote(r.column2) + ", \"flags\": " + std::to_string(r.flags) + ", \"pattern\": " + json::quote(r.pattern) +
           ", \"kll_k\": " + std::to_string(r.kll_k) + "}";
    }
    o += "]}";
    *out_json = dup_string(o);
    return TGX_OK;
  } catch (const TermError &e) {
    return hfail(err, TGX_INVALID_ARGUMENT, e.display());
  } catch (const std::bad_alloc &) {
    return hfail(err, TGX_OUT_OF_MEMORY, "host allocation failed (std::bad_alloc)");
  } catch (const std::exception &e) {
    return hfail(err, TGX_INTERNAL, e.what());
  } catch (...) {
    return hfail(err, TGX_INTERNAL, "unknown C++ exception");
  }
}

namespace {
struct FakeQuantiles {
  std::vector<std::vector<std::pair<double, double>>> per_request;
};
double fake_quantile(const void *ctx, size_t idx, double phi) {
  const FakeQuantiles *f = (const FakeQuantiles *)ctx;
  for (auto &kv : f->per_request.at(idx))
    if (kv.first == phi) return kv.second;
  throw TermError{TermError::Internal, "results_json has no quantile for phi=" + rust_f64(phi)};
}
}  // namespace

extern "C" tgx_status tgx_host_constraint_verdict_json(const char *constraint_json, const char *results_json,
                                                       char **out_json, tgx_error *err) {
  if (!constraint_json || !results_json || !out_json) return hfail(err, TGX_INVALID_ARGUMENT, "NULL argument");
  *out_json = nullptr;
  try {
    auto c = constraint_from_json_text(constraint_json);
    json::Value rv;
    std::string perr;
    if (!json::parse(results_json, &rv, &perr) || !rv.is(json::Value::Array))
      return hfail(err, TGX_INVALID_ARGUMENT, "results JSON must be an array: " + perr);
    std::vector<tgx_result> results(rv.arr.size());
    FakeQuantiles fq;
    fq.per_request.resize(rv.arr.size());
    for (size_t i = 0; i < rv.arr.size(); i++) {
      const json::Value &o = rv.arr[i];
      tgx_result &r = results[i];
      memset(&r, 0, sizeof(r));
      r.is_float = (int32_t)o.get_i64("is_float");
      r.total = o.get_i64("total");
      r.non_null = o.get_i64("non_null");
      r.has_value = (int32_t)o.get_i64("has_value");
      r.has_variance = (int32_t)o.get_i64("has_variance");
      r.min_i = o.get_i64("min_i");
      r.max_i = o.get_i64("max_i");
      r.min_f = o.get_num("min_f");
      r.max_f = o.get_num("max_f");
      r.sum_i = o.get_i64("sum_i");
      r.sum_f = o.get_num("sum_f");
      r.mean = o.get_num("mean");
      r.var_samp = o.get_num("var_samp");
      r.stddev_samp = o.get_num("stddev_samp");
      r.distinct = o.get_i64("distinct");
      r.groups_once = o.get_i64("groups_once");
      r.matches = o.get_i64("matches");
      r.sum_x = o.get_num("sum_x");
      r.sum_y = o.get_num("sum_y");
      r.sum_x2 = o.get_num("sum_x2");
      r.sum_y2 = o.get_num("sum_y2");
      r.sum_xy = o.get_num("sum_xy");
      if (o.get("co_m2_x")) {
        r.co_mean_x = o.get_num("co_mean_x");
        r.co_mean_y = o.get_num("co_mean_y");
        r.co_m2_x = o.get_num("co_m2_x");
        r.co_m2_y = o.get_num("co_m2_y");
        r.co_c_xy = o.get_num("co_c_xy");
      } else if (r.non_null > 0) {  // results that carry the raw sums only: centred from them
        const double n = (double)r.non_null;
        r.co_mean_x = r.sum_x / n;
        r.co_mean_y = r.sum_y / n;
        r.co_m2_x = std::max(0.0, r.sum_x2 - r.sum_x * r.sum_x / n);
        r.co_m2_y = std::max(0.0, r.sum_y2 - r.sum_y * r.sum_y / n);
        r.co_c_xy = r.sum_xy - r.sum_x * r.sum_y / n;
      }
      r.kll_n = o.get_u64("kll_n");
      if (const json::Value *q = o.get("quantiles"))
        for (auto &kv : q->obj) fq.per_request[i].emplace_back(atof(kv.first.c_str()), kv.second.num);
    }
    const size_t want = c->plan().size();
    if (results.size() != want)
      return hfail(err, TGX_INVALID_ARGUMENT,
                   "constraint plans " + std::to_string(want) + " aggregates, got " + std::to_string(results.size()));
    Constraint::Inputs in;
    for (auto &r : results) in.results.push_back(&r);
    in.ctx = &fq;
    in.quantile = fake_quantile;
    {  // the reference's result-type rule (term_guard.h reference_extracts): "column_type" = the column's Arrow DataType
      json::Value cv;
      std::string cerr;
      if (json::parse(constraint_json, &cv, &cerr) && cv.is(json::Value::Object)) {
        if (cv.get("column_type")) in.arrow_types.assign(results.size(), cv.get_str("column_type"));
        in.strict_reference_types = cv.get_bool("strict_reference_types", true);
      }
    }
    ConstraintResult cr = c->evaluate(in);
    const char *st = cr.status == ConstraintStatus::Success ? "success" : cr.status == ConstraintStatus::Failure ? "failure" : "skipped";
    std::string o = std::string("{\"status\": \"") + st + "\", \"metric\": ";
    if (cr.metric && !std::isnan(*cr.metric) && !std::isinf(*cr.metric)) {
      char buf[64];
      snprintf(buf, sizeof(buf), "%.17g", *cr.metric);
      o += buf;
    } else {
      o += "null";
    }
    o += ", \"message\": " + (cr.message ? json::quote(*cr.message) : std::string("null")) + ", \"name\": " +
         json::quote(c->name()) + "}";
    *out_json = dup_string(o);
    return TGX_OK;
  } catch (const TermError &e) {
    return hfail(err, TGX_INVALID_ARGUMENT, e.display());
  } catch (const std::bad_alloc &) {
    return hfail(err, TGX_OUT_OF_MEMORY, "host allocation failed (std::bad_alloc)");
  } catch (const std::exception &e) {
    return hfail(err, TGX_INTERNAL, e.what());
  } catch (...) {
    return hfail(err, TGX_INTERNAL, "unknown C++ exception");
  }
}

extern "C" tgx_status tgx_host_validate_identifier(const char *identifier, tgx_error *err) try {
  if (!identifier) return hfail(err, TGX_INVALID_ARGUMENT, "NULL argument");
  auto e = validate_identifier(identifier);
  if (e) return hfail(err, TGX_INVALID_ARGUMENT, e->display());
  return TGX_OK;
} catch (...) {
  return hfail(err, TGX_INTERNAL, "C++ exception while validating the identifier");
}

namespace term_guard {
ValidationSuite suite_from_json(const std::string &text);
}

extern "C" tgx_status tgx_host_assertion_json(const char *assertion_json, double value, int32_t *holds,
                                              char **description, tgx_error *err) {
  if (!assertion_json) return hfail(err, TGX_INVALID_ARGUMENT, "NULL argument");
  try {
    // reuse the constraint parser: wrap the assertion into a size constraint
    std::string wrapped = std::string("{\"type\": \"size\", \"assertion\": ") + assertion_json + "}";
    auto c = constraint_from_json_text(wrapped.c_str());
    tgx_result r;
    memset(&r, 0, sizeof(r));
    // evaluate through Size: metric == total; only exact for integral values, so parse the assertion directly too
    json::Value v;
    std::string perr;
    json::parse(assertion_json, &v, &perr);
    Assertion a = Assertion::equals(0);
    const std::string k = v.get_str("kind");
    const json::Value *args = v.get("args");
    auto arg = [&](size_t i) { return args && args->arr.size() > i ? args->arr[i].num : 0.0; };
    if (k == "equals") a = Assertion::equals(arg(0));
    else if (k == "not_equals") a = Assertion::not_equals(arg(0));
    else if (k == "greater_than") a = Assertion::greater_than(arg(0));
    else if (k == "greater_than_or_equal") a = Assertion::greater_than_or_equal(arg(0));
    else if (k == "less_than") a = Assertion::less_than(arg(0));
    else if (k == "less_than_or_equal") a = Assertion::less_than_or_equal(arg(0));
    else if (k == "between") a = Assertion::between(arg(0), arg(1));
    else if (k == "not_between") a = Assertion::not_between(arg(0), arg(1));
    if (holds) *holds = a.evaluate(value) ? 1 : 0;
    if (description) *description = dup_string(a.description());
    return TGX_OK;
  } catch (const TermError &e) {
    return hfail(err, TGX_INVALID_ARGUMENT, e.display());
  } catch (const std::bad_alloc &) {
    return hfail(err, TGX_OUT_OF_MEMORY, "host allocation failed (std::bad_alloc)");
  } catch (const std::exception &e) {
    return hfail(err, TGX_INTERNAL, e.what());
  } catch (...) {
    return hfail(err, TGX_INTERNAL, "unknown C++ exception");
  }
}

// ---- analyzers
static json::Value parse_json_text(const char *text, const char *what) {
  json::Value v;
  std::string perr;
  if (!json::parse(text, &v, &perr)) throw TermError{TermError::Internal, std::string(what) + " JSON: " + perr};
  return v;
}

extern "C" tgx_status tgx_host_run_analysis_json(const char *analysis_json, const char *const *column_names,
                                                 size_t n_columns, const tgx_column *columns, size_t n_batches,
                                                 char **out_json, tgx_error *err) {
  if (!analysis_json || !out_json || (n_columns && (!column_names || (n_batches && !columns))))
    return hfail(err, TGX_INVALID_ARGUMENT, "NULL argument");
  *out_json = nullptr;
  try {
    json::Value v = parse_json_text(analysis_json, "analysis");
    AnalysisRunner runner;
    runner.table_name(v.get_str("table_name", "data")).continue_on_error(v.get_bool("continue_on_error", true));
    if (const json::Value *as = v.get("analyzers"))
      for (const json::Value &a : as->arr) runner.add(analyzer_from_json(a));
    Table t;
    for (size_t i = 0; i < n_columns; i++) t.column_names.push_back(column_names[i]);
    for (size_t b = 0; b < n_batches; b++) {
      Batch batch;
      batch.columns.assign(columns + b * n_columns, columns + (b + 1) * n_columns);
      t.batches.push_back(std::move(batch));
    }
    Context ctx;
    if (n_columns > 0) ctx.register_table(v.get_str("table_name", "data"), std::move(t));
    *out_json = dup_string(runner.run(ctx).to_json());
    return TGX_OK;
  } catch (const AnalyzerError &e) {
    return hfail(err, TGX_INVALID_ARGUMENT, e.text);
  } catch (const TermError &e) {
    return hfail(err, TGX_INVALID_ARGUMENT, e.display());
  } catch (const std::bad_alloc &) {
    return hfail(err, TGX_OUT_OF_MEMORY, "host allocation failed (std::bad_alloc)");
  } catch (const std::exception &e) {
    return hfail(err, TGX_INTERNAL, e.what());
  } catch (...) {
    return hfail(err, TGX_INTERNAL, "unknown C++ exception");
  }
}

extern "C" tgx_status tgx_host_merge_states_json(const char *analyzer_json, const char *states_json,
                                                 char **out_state_json, tgx_error *err) {
  if (!analyzer_json || !states_json || !out_state_json) return hfail(err, TGX_INVALID_ARGUMENT, "NULL argument");
  *out_state_json = nullptr;
  try {
    auto a = analyzer_from_json(parse_json_text(analyzer_json, "analyzer"));
    json::Value states = parse_json_text(states_json, "states");
    if (states.type != json::Value::Array) return hfail(err, TGX_INVALID_ARGUMENT, "states must be a JSON array");
    *out_state_json = dup_string(json_dump(a->merge_states(states.arr)));
    return TGX_OK;
  } catch (const AnalyzerError &e) {
    return hfail(err, TGX_INVALID_ARGUMENT, e.text);
  } catch (const TermError &e) {
    return hfail(err, TGX_INVALID_ARGUMENT, e.display());
  } catch (const std::bad_alloc &) {
    return hfail(err, TGX_OUT_OF_MEMORY, "host allocation failed (std::bad_alloc)");
  } catch (const std::exception &e) {
    return hfail(err, TGX_INTERNAL, e.what());
  } catch (...) {
    return hfail(err, TGX_INTERNAL, "unknown C++ exception");
  }
}

extern "C" tgx_status tgx_host_metric_from_state_json(const char *analyzer_json, const char *state_json,
                                                      char **out_metric_json, tgx_error *err) {
  if (!analyzer_json || !state_json || !out_metric_json) return hfail(err, TGX_INVALID_ARGUMENT, "NULL argument");
  *out_metric_json = nullptr;
  try {
    auto a = analyzer_from_json(parse_json_text(analyzer_json, "analyzer"));
    *out_metric_json = dup_string(a->metric_from_state(parse_json_text(state_json, "state")).to_json());
    return TGX_OK;
  } catch (const AnalyzerError &e) {
    return hfail(err, TGX_INVALID_ARGUMENT, e.text);
  } catch (const TermError &e) {
    return hfail(err, TGX_INVALID_ARGUMENT, e.display());
  } catch (const std::bad_alloc &) {
    return hfail(err, TGX_OUT_OF_MEMORY, "host allocation failed (std::bad_alloc)");
  } catch (const std::exception &e) {
    return hfail(err, TGX_INTERNAL, e.what());
  } catch (...) {
    return hfail(err, TGX_INTERNAL, "unknown C++ exception");
  }
}
