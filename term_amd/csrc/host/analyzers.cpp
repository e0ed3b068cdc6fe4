// analyzers.cpp -- see analyzers.h.  Reference: TG/analyzers/{traits,runner,context,types,errors}.rs,
// TG/analyzers/basic/{size,completeness,distinctness,mean,min_max,sum}.rs,
// TG/analyzers/advanced/{standard_deviation,correlation}.rs.
#include "analyzers.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <deque>

namespace term_guard {

// ---------------------------------------------------------------- JSON helpers
static json::Value jnum(double v) {
  json::Value x;
  x.type = json::Value::Number;
  x.num = v;
  return x;
}
static json::Value jbool(bool v) {
  json::Value x;
  x.type = json::Value::Bool;
  x.b = v;
  return x;
}
static json::Value jnull() { return json::Value(); }
static json::Value jstr(const std::string &s) {
  json::Value x;
  x.type = json::Value::String;
  x.str = s;
  return x;
}
static json::Value jobj(std::vector<std::pair<std::string, json::Value>> kv) {
  json::Value x;
  x.type = json::Value::Object;
  x.obj = std::move(kv);
  return x;
}
static json::Value jcount(uint64_t v) { return json::Value::of_u64(v); }  // a `u64` field of a reference state struct
// An f64 field: the shortest text that reads back to the same double, and always with a fraction or an exponent so the
// token is a float to every reader (serde_json prints 30.0, 1e16, 0.1 the same way).
static std::string num_text(double v) {
  if (isnan(v) || isinf(v)) return "null";  // serde_json writes non-finite f64 as null
  char buf[40];
  if (v == floor(v) && fabs(v) < 9.0e15) {
    snprintf(buf, sizeof(buf), "%.1f", v);
    return buf;
  }
  for (int prec = 15; prec <= 17; prec++) {
    snprintf(buf, sizeof(buf), "%.*g", prec, v);
    if (strtod(buf, nullptr) == v) break;
  }
  if (!strpbrk(buf, ".eE")) strcat(buf, ".0");
  return buf;
}
static std::string value_text(const json::Value &v) {
  if (v.int_kind == json::Value::Unsigned) return std::to_string((unsigned long long)v.u);
  if (v.int_kind == json::Value::Signed) return std::to_string((long long)v.i);
  return num_text(v.num);
}
std::string json_dump(const json::Value &v) {
  switch (v.type) {
    case json::Value::Null: return "null";
    case json::Value::Bool: return v.b ? "true" : "false";
    case json::Value::Number: return value_text(v);
    case json::Value::String: return json::quote(v.str);
    case json::Value::Array: {
      std::string o = "[";
      for (size_t i = 0; i < v.arr.size(); i++) o += (i ? ", " : "") + json_dump(v.arr[i]);
      return o + "]";
    }
    case json::Value::Object: {
      std::string o = "{";
      for (size_t i = 0; i < v.obj.size(); i++)
        o += (i ? ", " : "") + json::quote(v.obj[i].first) + ": " + json_dump(v.obj[i].second);
      return o + "}";
    }
  }
  return "null";
}
static double f(const json::Value &s, const char *k) { return s.get_num(k, 0.0); }
static uint64_t u(const json::Value &s, const char *k) { return s.get_u64(k, 0); }
static std::optional<double> opt(const json::Value &s, const char *k) {
  const json::Value *v = s.get(k);
  if (!v || v->type != json::Value::Number) return std::nullopt;
  return v->num;
}

std::string MetricValue::to_json() const {
  switch (kind) {
    case Double: return "{\"type\": \"Double\", \"value\": " + num_text(d) + "}";
    case Long: return "{\"type\": \"Long\", \"value\": " + std::to_string(l) + "}";
    case Map: {
      std::string o = "{\"type\": \"Map\", \"value\": {";
      for (size_t i = 0; i < map.size(); i++) o += (i ? ", " : "") + json::quote(map[i].first) + ": " + map[i].second.to_json();
      return o + "}}";
    }
  }
  return "null";
}

std::string AnalyzerContext::to_json() const {
  std::string o = "{\"metrics\": {";
  for (size_t i = 0; i < metrics.size(); i++)
    o += (i ? ", " : "") + json::quote(metrics[i].first) + ": " + metrics[i].second.to_json();
  o += "}, \"states\": {";
  for (size_t i = 0; i < states.size(); i++)
    o += (i ? ", " : "") + json::quote(states[i].first) + ": " + json_dump(states[i].second);
  o += "}, \"errors\": [";
  for (size_t i = 0; i < errors.size(); i++)
    o += std::string(i ? ", " : "") + "{\"analyzer_name\": " + json::quote(errors[i].first) + ", \"error\": " +
         json::quote(errors[i].second) + "}";
  return o + "]}";
}

// ---------------------------------------------------------------- analyzers
namespace {

SpecRequest req(int kind, const std::string &col, uint32_t flags = 0) {
  SpecRequest r;
  r.kind = kind;
  r.column = col;
  r.flags = flags;
  return r;
}

class SizeAnalyzer : public Analyzer {  // basic/size.rs
 public:
  std::string name() const override { return "size"; }
  std::vector<SpecRequest> plan() const override { return {req(TGX_CHECK_COUNT, "")}; }  // COUNT(*): any column
  json::Value state_from_results(const std::vector<const tgx_result *> &r, const std::vector<int> &) const override {
    return jobj({{"count", jcount(r[0]->total)}});
  }
  json::Value merge_states(const std::vector<json::Value> &states) const override {  // size.rs:60-63
    uint64_t c = 0;
    for (auto &s : states) c += u(s, "count");
    return jobj({{"count", jcount(c)}});
  }
  MetricValue metric_from_state(const json::Value &s) const override { return MetricValue::of_long((int64_t)u(s, "count")); }
};

class ColumnAnalyzer : public Analyzer {
 public:
  explicit ColumnAnalyzer(std::string c) : column_(std::move(c)) {}
  std::string metric_key() const override { return name() + "." + column_; }
  std::vector<std::string> columns() const override { return {column_}; }

 protected:
  std::string column_;
};

class CompletenessAnalyzer : public ColumnAnalyzer {  // basic/completeness.rs
 public:
  using ColumnAnalyzer::ColumnAnalyzer;
  std::string name() const override { return "completeness"; }
  std::vector<SpecRequest> plan() const override { return {req(TGX_CHECK_COUNT, column_)}; }
  json::Value state_from_results(const std::vector<const tgx_result *> &r, const std::vector<int> &) const override {
    return jobj({{"total_count", jcount(r[0]->total)}, {"non_null_count", jcount(r[0]->non_null)}});
  }
  json::Value merge_states(const std::vector<json::Value> &states) const override {  // :76-84
    uint64_t t = 0, n = 0;
    for (auto &s : states) {
      t += u(s, "total_count");
      n += u(s, "non_null_count");
    }
    return jobj({{"total_count", jcount(t)}, {"non_null_count", jcount(n)}});
  }
  MetricValue metric_from_state(const json::Value &s) const override {  // :62-68: empty dataset is complete
    const uint64_t t = u(s, "total_count");
    return MetricValue::of_double(t == 0 ? 1.0 : (double)u(s, "non_null_count") / (double)t);
  }
};

class DistinctnessAnalyzer : public ColumnAnalyzer {  // basic/distinctness.rs
 public:
  using ColumnAnalyzer::ColumnAnalyzer;
  std::string name() const override { return "distinctness"; }
  std::vector<SpecRequest> plan() const override { return {req(TGX_CHECK_DISTINCT, column_)}; }
  json::Value state_from_results(const std::vector<const tgx_result *> &r, const std::vector<int> &) const override {
    // COUNT(col) is the denominator, not COUNT(*) (:113-116)
    return jobj({{"total_count", jcount(r[0]->non_null)}, {"distinct_count", jcount(r[0]->distinct)}});
  }
  json::Value merge_states(const std::vector<json::Value> &states) const override {  // :77-92: clamped sum, an upper bound
    uint64_t t = 0, d = 0;
    for (auto &s : states) {
      t += u(s, "total_count");
      d += u(s, "distinct_count");
    }
    return jobj({{"total_count", jcount(t)}, {"distinct_count", jcount(std::min(d, t))}});
  }
  MetricValue metric_from_state(const json::Value &s) const override {  // :35-41
    const uint64_t t = u(s, "total_count");
    return MetricValue::of_double(t == 0 ? 1.0 : (double)u(s, "distinct_count") / (double)t);
  }
};

// advanced/approx_count_distinct.rs: APPROX_DISTINCT(col) (DataFusion's HyperLogLog, a third-party estimate whose
// value is unpinned by the reference's tests) and COUNT(col).  The device path keeps a HyperLogLog of the same shape on
// the column's scan (TGX_CHECK_APPROX_DISTINCT; the exact count on string columns); state fields, the max-merge and the
// metric type are the reference's.
class ApproxCountDistinctAnalyzer : public ColumnAnalyzer {
 public:
  using ColumnAnalyzer::ColumnAnalyzer;
  std::string name() const override { return "approx_count_distinct"; }
  std::vector<SpecRequest> plan() const override { return {req(TGX_CHECK_APPROX_DISTINCT, column_)}; }
  json::Value state_from_results(const std::vector<const tgx_result *> &r, const std::vector<int> &) const override {
    return jobj({{"approx_distinct_count", jcount(r[0]->distinct)}, {"total_count", jcount(r[0]->non_null)}});
  }
  json::Value merge_states(const std::vector<json::Value> &states) const override {  // :45-61: max of the counts
    uint64_t d = 0, t = 0;
    for (auto &s : states) {
      d = std::max(d, u(s, "approx_distinct_count"));
      t += u(s, "total_count");
    }
    return jobj({{"approx_distinct_count", jcount(d)}, {"total_count", jcount(t)}});
  }
  MetricValue metric_from_state(const json::Value &s) const override {  // :126-128
    return MetricValue::of_long((int64_t)u(s, "approx_distinct_count"));
  }
};

// SUM(col) comes back as Float64 for Float64 columns and Int64 (wrapping) for Int64 columns
double sql_sum(const tgx_result *r) { return r->is_float ? r->sum_f : (double)r->sum_i; }

class MeanAnalyzer : public ColumnAnalyzer {  // basic/mean.rs
 public:
  using ColumnAnalyzer::ColumnAnalyzer;
  std::string name() const override { return "mean"; }
  std::vector<SpecRequest> plan() const override { return {req(TGX_CHECK_NUMERIC_STATS, column_)}; }
  json::Value state_from_results(const std::vector<const tgx_result *> &r, const std::vector<int> &) const override {
    // the sum is read as Float64Array only (:117-126): an Int64 column's SUM is Int64 -> InvalidData, unless the
    // sum is NULL (no non-null value), which reads as 0.0
    if (r[0]->non_null > 0 && !r[0]->is_float) throw AnalyzerError::invalid_data("Expected Float64 array for sum");
    return jobj({{"sum", jnum(r[0]->non_null > 0 ? r[0]->sum_f : 0.0)}, {"count", jcount(r[0]->non_null)}});
  }
  json::Value merge_states(const std::vector<json::Value> &states) const override {
    double sum = 0;
    uint64_t c = 0;
    for (auto &s : states) {
      sum += f(s, "sum");
      c += u(s, "count");
    }
    return jobj({{"sum", jnum(sum)}, {"count", jcount(c)}});
  }
  MetricValue metric_from_state(const json::Value &s) const override {  // :147-152
    const uint64_t c = u(s, "count");
    if (c == 0) throw AnalyzerError::no_data();
    return MetricValue::of_double(f(s, "sum") / (double)c);
  }
};

class MinMaxAnalyzer : public ColumnAnalyzer {  // basic/min_max.rs (MinAnalyzer / MaxAnalyzer share MinMaxState)
 public:
  MinMaxAnalyzer(std::string c, bool is_max) : ColumnAnalyzer(std::move(c)), is_max_(is_max) {}
  std::string name() const override { return is_max_ ? "max" : "min"; }
  std::vector<SpecRequest> plan() const override { return {req(TGX_CHECK_NUMERIC_STATS, column_)}; }
  json::Value state_from_results(const std::vector<const tgx_result *> &r, const std::vector<int> &) const override {
    if (!r[0]->has_value) return jobj({{"min", jnull()}, {"max", jnull()}});
    const double mn = r[0]->is_float ? r[0]->min_f : (double)r[0]->min_i;  // Int64 -> `as f64` (:118-123)
    const double mx = r[0]->is_float ? r[0]->max_f : (double)r[0]->max_i;
    return jobj({{"min", jnum(mn)}, {"max", jnum(mx)}});
  }
  json::Value merge_states(const std::vector<json::Value> &states) const override {  // :13-29
    std::optional<double> mn, mx;
    for (auto &s : states) {
      if (auto v = opt(s, "min")) mn = mn ? std::min(*mn, *v) : *v;
      if (auto v = opt(s, "max")) mx = mx ? std::max(*mx, *v) : *v;
    }
    return jobj({{"min", mn ? jnum(*mn) : jnull()}, {"max", mx ? jnum(*mx) : jnull()}});
  }
  MetricValue metric_from_state(const json::Value &s) const override {  // :167-172 / :316-321
    auto v = opt(s, is_max_ ? "max" : "min");
    if (!v) throw AnalyzerError::no_data();
    return MetricValue::of_double(*v);
  }

 private:
  bool is_max_;
};

class SumAnalyzer : public ColumnAnalyzer {  // basic/sum.rs
 public:
  using ColumnAnalyzer::ColumnAnalyzer;
  std::string name() const override { return "sum"; }
  std::vector<SpecRequest> plan() const override { return {req(TGX_CHECK_NUMERIC_STATS, column_)}; }
  json::Value state_from_results(const std::vector<const tgx_result *> &r, const std::vector<int> &) const override {
    return jobj({{"sum", jnum(r[0]->non_null > 0 ? sql_sum(r[0]) : 0.0)}, {"has_values", jbool(r[0]->non_null > 0)}});
  }
  json::Value merge_states(const std::vector<json::Value> &states) const override {  // :35-40
    double sum = 0;
    bool any = false;
    for (auto &s : states) {
      sum += f(s, "sum");
      any = any || s.get_bool("has_values");
    }
    return jobj({{"sum", jnum(sum)}, {"has_values", jbool(any)}});
  }
  MetricValue metric_from_state(const json::Value &s) const override {  // :145-151
    if (!s.get_bool("has_values")) throw AnalyzerError::no_data();
    return MetricValue::of_double(f(s, "sum"));
  }
};

class StandardDeviationAnalyzer : public ColumnAnalyzer {  // advanced/standard_deviation.rs
 public:
  using ColumnAnalyzer::ColumnAnalyzer;
  std::string name() const override { return "standard_deviation"; }
  std::string metric_key() const override { return name(); }  // the reference does not add the column (:281-291)
  std::vector<SpecRequest> plan() const override { return {req(TGX_CHECK_NUMERIC_STATS, column_, TGX_FLAG_VARIANCE)}; }
  json::Value state_from_results(const std::vector<const tgx_result *> &r, const std::vector<int> &) const override {
    const double n = (double)r[0]->non_null;
    if (r[0]->non_null == 0) return state(0, 0.0, 0.0, 0.0);  // :192-193
    // COUNT, AVG, SUM, SUM(x*x) WHERE x IS NOT NULL (:171-180); SUM of an Int64 column is Int64: "Expected Float64"
    if (!r[0]->is_float) throw AnalyzerError::invalid_data("Expected Float64 for sum");
    const double sum = r[0]->sum_f;
    // SUM(x*x) from the shifted moments the scan keeps: M2 + sum^2 / n  (M2 = var_samp * (n - 1))
    const double m2 = r[0]->has_variance ? r[0]->var_samp * (n - 1.0) : 0.0;
    return state(r[0]->non_null, sum, m2 + sum * sum / n, r[0]->mean);
  }
  json::Value merge_states(const std::vector<json::Value> &states) const override {  // :133-150
    if (states.empty()) throw AnalyzerError::state_merge("No states to merge");
    uint64_t c = 0;
    double sum = 0, sq = 0;
    for (auto &s : states) {
      c += u(s, "count");
      sum += f(s, "sum");
      sq += f(s, "sum_squared");
    }
    return state(c, sum, sq, c > 0 ? sum / (double)c : 0.0);
  }
  MetricValue metric_from_state(const json::Value &s) const override {  // :239-279
    const uint64_t c = u(s, "count");
    const double sum = f(s, "sum"), sq = f(s, "sum_squared"), mean = f(s, "mean");
    MetricValue m;
    m.kind = MetricValue::Map;
    m.map.push_back({"count", MetricValue::of_long((int64_t)c)});
    m.map.push_back({"mean", MetricValue::of_double(mean)});
    std::optional<double> pop_var, samp_var;
    if (c > 0) pop_var = std::max(sq / (double)c - mean * mean, 0.0);                          // :78-88
    if (c > 1) samp_var = std::max((sq - sum * sum / (double)c) / (double)(c - 1), 0.0);       // :96-106
    if (pop_var) m.map.push_back({"std_dev", MetricValue::of_double(sqrt(*pop_var))});
    if (samp_var) m.map.push_back({"sample_std_dev", MetricValue::of_double(sqrt(*samp_var))});
    if (pop_var) m.map.push_back({"variance", MetricValue::of_double(*pop_var)});
    if (samp_var) m.map.push_back({"sample_variance", MetricValue::of_double(*samp_var)});
    if (pop_var && fabs(mean) >= 2.220446049250313e-16)                                         // :122-129
      m.map.push_back({"coefficient_of_variation", MetricValue::of_double(sqrt(*pop_var) / fabs(mean))});
    return m;
  }

 private:
  static json::Value state(uint64_t c, double sum, double sq, double mean) {
    return jobj({{"count", jcount(c)}, {"sum", jnum(sum)}, {"sum_squared", jnum(sq)}, {"mean", jnum(mean)}});
  }
};

class CorrelationAnalyzer : public Analyzer {  // advanced/correlation.rs
 public:
  enum Type { Pearson, Spearman, Covariance };
  CorrelationAnalyzer(std::string a, std::string b, Type t) : a_(std::move(a)), b_(std::move(b)), t_(t) {}
  std::string name() const override { return "correlation"; }
  std::string metric_key() const override {  // :445-452
    return std::string("correlation_") + (t_ == Pearson ? "pearson" : t_ == Spearman ? "spearman" : "covariance") + "_" +
           a_ + "_" + b_;
  }
  std::vector<std::string> columns() const override { return {a_, b_}; }
  std::vector<SpecRequest> plan() const override {
    SpecRequest r = req(t_ == Spearman ? TGX_CHECK_SPEARMAN : TGX_CHECK_COMOMENTS, a_);
    r.column2 = b_;
    return {r};
  }
  json::Value state_from_results(const std::vector<const tgx_result *> &r, const std::vector<int> &) const override {
    return state((uint64_t)r[0]->non_null, r[0]->sum_x, r[0]->sum_y, r[0]->sum_x2, r[0]->sum_y2, r[0]->sum_xy);
  }
  json::Value merge_states(const std::vector<json::Value> &states) const override {  // :65-110
    if (states.empty()) throw AnalyzerError::state_merge("Cannot merge empty states");
    if (t_ == Spearman) throw AnalyzerError::state_merge("Cannot merge rank-based correlation states");
    uint64_t n = 0;
    double v[5] = {0, 0, 0, 0, 0};
    static const char *k[5] = {"sum_x", "sum_y", "sum_x2", "sum_y2", "sum_xy"};
    for (auto &s : states) {
      n += u(s, "n");
      for (int i = 0; i < 5; i++) v[i] += f(s, k[i]);
    }
    return state(n, v[0], v[1], v[2], v[3], v[4]);
  }
  MetricValue metric_from_state(const json::Value &s) const override {  // :407-435
    const uint64_t nn = u(s, "n");
    if (nn < 2) return MetricValue::of_double(NAN);
    const double n = (double)nn, sx = f(s, "sum_x"), sy = f(s, "sum_y"), sx2 = f(s, "sum_x2"), sy2 = f(s, "sum_y2"),
                 sxy = f(s, "sum_xy");
    if (t_ == Covariance) return MetricValue::of_double((sxy - (sx * sy) / n) / (n - 1.0));
    const double num = n * sxy - sx * sy;
    const double den = sqrt((n * sx2 - sx * sx) * (n * sy2 - sy * sy));
    return MetricValue::of_double(den == 0.0 ? 0.0 : num / den);
  }

 private:
  json::Value state(uint64_t n, double sx, double sy, double sx2, double sy2, double sxy) const {
    return jobj({{"n", jcount(n)}, {"sum_x", jnum(sx)}, {"sum_y", jnum(sy)}, {"sum_x2", jnum(sx2)},
                 {"sum_y2", jnum(sy2)}, {"sum_xy", jnum(sxy)}, {"x_ranks", jnull()}, {"y_ranks", jnull()},
                 {"correlation_type", jstr(t_ == Pearson ? "Pearson" : t_ == Spearman ? "Spearman" : "Covariance")}});
  }
  std::string a_, b_;
  Type t_;
};

struct Handles {
  tgx_plan *plan = nullptr;
  tgx_state *state = nullptr;
  Handles() = default;
  Handles(const Handles &) = delete;
  Handles &operator=(const Handles &) = delete;
  void reset() {
    if (state) tgx_state_destroy(state);
    if (plan) tgx_plan_destroy(plan);
    state = nullptr;
    plan = nullptr;
  }
  ~Handles() { reset(); }
};

}  // namespace

std::shared_ptr<Analyzer> analyzer_from_json(const json::Value &v) {
  const std::string t = v.get_str("type");
  auto col = [&]() {
    const std::string c = v.get_str("column");
    if (c.empty()) throw TermError{TermError::Internal, "analyzer '" + t + "' needs a column"};
    return c;
  };
  if (t == "size") return std::make_shared<SizeAnalyzer>();
  if (t == "completeness") return std::make_shared<CompletenessAnalyzer>(col());
  if (t == "distinctness") return std::make_shared<DistinctnessAnalyzer>(col());
  if (t == "approx_count_distinct") return std::make_shared<ApproxCountDistinctAnalyzer>(col());
  if (t == "mean") return std::make_shared<MeanAnalyzer>(col());
  if (t == "min") return std::make_shared<MinMaxAnalyzer>(col(), false);
  if (t == "max") return std::make_shared<MinMaxAnalyzer>(col(), true);
  if (t == "sum") return std::make_shared<SumAnalyzer>(col());
  if (t == "standard_deviation") return std::make_shared<StandardDeviationAnalyzer>(col());
  if (t == "correlation") {
    const std::string m = v.get_str("method", "pearson");
    CorrelationAnalyzer::Type ct = m == "pearson"      ? CorrelationAnalyzer::Pearson
                                   : m == "spearman"   ? CorrelationAnalyzer::Spearman
                                   : m == "covariance" ? CorrelationAnalyzer::Covariance
                                                       : throw TermError{TermError::Internal, "unknown correlation method '" + m + "'"};
    if (v.get_str("column1").empty() || v.get_str("column2").empty())
      throw TermError{TermError::Internal, "correlation needs column1 and column2"};
    return std::make_shared<CorrelationAnalyzer>(v.get_str("column1"), v.get_str("column2"), ct);
  }
  throw TermError{TermError::Internal, "unknown analyzer type '" + t + "'"};
}

// ---------------------------------------------------------------- runner
AnalyzerContext AnalysisRunner::run(const Context &ctx) const {
  AnalyzerContext out;
  const Table *table = ctx.table(table_name_);
  auto column_index = [&](const std::string &name) -> int {
    if (!table) return -1;
    for (size_t i = 0; i < table->column_names.size(); i++)
      if (table->column_names[i] == name) return (int)i;
    return -1;
  };
  struct Planned {
    std::vector<SpecRequest> reqs;
    std::vector<size_t> spec_index;
    std::vector<int> column_types;
    std::optional<std::string> error;
  };
  std::vector<Planned> planned(analyzers_.size());
  std::vector<SpecRequest> requests;
  for (size_t a = 0; a < analyzers_.size(); a++) {
    Planned &p = planned[a];
    if (!table) {
      p.error = AnalyzerError::query("Error during planning: table 'datafusion.public." + table_name_ + "' not found").text;
      continue;
    }
    for (SpecRequest r : analyzers_[a]->plan()) {
      if (r.column.empty() && !table->column_names.empty()) r.column = table->column_names[0];  // COUNT(*)
      for (const std::string *c : {&r.column, &r.column2}) {
        if (c == &r.column2 && r.column2.empty()) continue;
        if (column_index(*c) < 0 && !p.error)
          p.error = AnalyzerError::query("Schema error: No field named " + *c + ".").text;
      }
      if (p.error) break;
      p.reqs.push_back(r);
    }
    for (const std::string &c : analyzers_[a]->columns()) {
      int ci = column_index(c);
      int type = 0;
      if (ci >= 0)
        for (const Batch &b : table->batches)
          if (type == 0) type = b.columns[ci].type;
      p.column_types.push_back(type);
    }
  }
  // one pass over the table for every analyzer that is still in the run: their requests fused and de-duplicated
  std::vector<tgx_check_spec> specs;
  auto fuse = [&]() {
    requests.clear();
    specs.clear();
    for (Planned &p : planned) {
      p.spec_index.clear();
      if (p.error) continue;
      for (const SpecRequest &r : p.reqs) {
        size_t found = requests.size();
        for (size_t i = 0; i < requests.size(); i++)
          if (requests[i].kind == r.kind && requests[i].column == r.column && requests[i].column2 == r.column2 &&
              requests[i].flags == r.flags)
            found = i;
        if (found == requests.size()) requests.push_back(r);
        p.spec_index.push_back(found);
      }
    }
    for (const SpecRequest &r : requests) {
      tgx_check_spec s;
      memset(&s, 0, sizeof(s));
      s.kind = r.kind;
      s.column = column_index(r.column);
      s.column2 = r.column2.empty() ? -1 : column_index(r.column2);
      s.flags = r.flags;
      if (r.kind == TGX_CHECK_DISTINCT) s.flags |= TGX_FLAG_EXACT_KEYS;  // DistinctnessAnalyzer counts by value
      specs.push_back(s);
    }
  };
  Handles h;
  std::vector<tgx_result> results;
  std::optional<std::string> run_error;
  auto pass = [&](Handles &hh, size_t max_rows, tgx_status *status) -> std::optional<std::string> {
    tgx_error err;
    memset(&err, 0, sizeof(err));
    tgx_status s = tgx_init(nullptr, &err);
    if (s == TGX_OK) s = tgx_plan_create(specs.data(), specs.size(), &hh.plan, &err);
    if (s == TGX_OK) s = tgx_state_create(hh.plan, nullptr, &hh.state, &err);
    std::vector<tgx_column> cut;
    for (size_t b = 0; s == TGX_OK && b < table->batches.size(); b++) {
      const std::vector<tgx_column> &cols = table->batches[b].columns;
      if (max_rows == SIZE_MAX) {
        s = tgx_update(hh.plan, hh.state, cols.data(), cols.size(), &err);
      } else {  // (a probe: the first rows of the first batch)
        int64_t rows = 0;
        for (const tgx_column &c : cols) rows = std::max(rows, c.length);
        if (rows == 0 && b + 1 < table->batches.size()) continue;  // (an empty record batch says nothing)
        cut = cols;
        for (tgx_column &c : cut) c.length = std::min<int64_t>(c.length, (int64_t)max_rows);
        s = tgx_update(hh.plan, hh.state, cut.data(), cut.size(), &err);
        break;
      }
    }
    if (s == TGX_OK) {
      results.assign(specs.size(), tgx_result());
      s = tgx_finalize(hh.plan, hh.state, results.data(), results.size(), &err);
    }
    *status = s;
    if (s != TGX_OK) return AnalyzerError::query(std::string(tgx_status_name(s)) + ": " + err.msg).text;
    return std::nullopt;
  };
  fuse();
  if (!specs.empty()) {
    tgx_status status = TGX_OK;
    run_error = pass(h, SIZE_MAX, &status);
    if (run_error && (status == TGX_UNSUPPORTED || status == TGX_INVALID_ARGUMENT)) {
      // one analyzer the library refuses (a column type outside the path) keeps the refusal as ITS error -- the
      // reference runs every analyzer as a query of its own (runner.rs:141-201) -- and the others run again as one pass:
      // every analyzer's requests are tried alone on the table's first row (term_guard.cpp, ValidationSuite::run)
      std::vector<std::optional<std::string>> saved;
      for (const Planned &p : planned) saved.push_back(p.error);
      size_t refused = 0;
      for (size_t k = 0; k < planned.size(); k++) {
        if (saved[k]) continue;
        for (size_t j = 0; j < planned.size(); j++)
          if (j != k && !planned[j].error) planned[j].error = std::string();  // (not part of this probe)
        fuse();
        Handles probe;
        tgx_status ps = TGX_OK;
        const std::optional<std::string> pe = specs.empty() ? std::nullopt : pass(probe, 1, &ps);
        for (size_t j = 0; j < planned.size(); j++) planned[j].error = saved[j];
        if (pe && (ps == TGX_UNSUPPORTED || ps == TGX_INVALID_ARGUMENT)) {
          planned[k].error = saved[k] = pe;
          refused++;
        }
      }
      fuse();
      if (refused) {
        h.reset();
        run_error.reset();
        if (!specs.empty()) run_error = pass(h, SIZE_MAX, &status);
      }
    }
  }
  for (size_t a = 0; a < analyzers_.size(); a++) {
    const Analyzer &an = *analyzers_[a];
    std::optional<std::string> error = planned[a].error ? planned[a].error : run_error;
    if (!error) {
      try {
        std::vector<const tgx_result *> r;
        for (size_t si : planned[a].spec_index) r.push_back(&results[si]);
        json::Value st = an.state_from_results(r, planned[a].column_types);
        MetricValue m = an.metric_from_state(st);
        out.states.push_back({an.metric_key(), st});
        // context.rs:71-73: a later analyzer with the same key replaces the earlier metric
        bool replaced = false;
        for (auto &kv : out.metrics)
          if (kv.first == an.metric_key()) {
            kv.second = m;
            replaced = true;
          }
        if (!replaced) out.metrics.push_back({an.metric_key(), m});
      } catch (const AnalyzerError &e) {
        error = e.text;
      }
    }
    if (error) {
      out.errors.push_back({an.name(), *error});  // runner.rs:172-175
      if (!continue_on_error_) throw AnalyzerError::custom("Analyzer " + an.name() + " failed");  // :177-181
    }
  }
  return out;
}

}  // namespace term_guard
