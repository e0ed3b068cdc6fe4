// term_guard.cpp -- see term_guard.h.  Verdict rules and message texts follow the reference files cited
// next to each constraint; the aggregates come from libtgx (one fused plan per suite run).
#include "term_guard.h"

#include <deque>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <chrono>

#include "json.h"

namespace term_guard {

// ------------------------------------------------------------------------------------------------ basics
const char *level_str(Level l) {
  switch (l) {
    case Level::Info: return "info";
    case Level::Warning: return "warning";
    default: return "error";
  }
}

std::string TermError::display() const {
  switch (kind) {
    case SecurityError: return "Security error: " + message;
    case DataFusion: return "DataFusion error: " + message;
    case NotSupported: return "Operation not supported: " + message;
    case Configuration: return "Configuration error: " + message;
    case TypeMismatch: return "Type mismatch: " + message;  // error.rs:81 ("expected {expected}, found {found}")
    default: return "Internal error: " + message;
  }
}

std::string rust_f64(double v) {
  if (isnan(v)) return "NaN";
  if (isinf(v)) return v > 0 ? "inf" : "-inf";
  if (v == 0) return signbit(v) ? "-0" : "0";
  char buf[64];
  int prec = 1;
  for (; prec <= 17; prec++) {
    snprintf(buf, sizeof(buf), "%.*e", prec - 1, v);
    if (strtod(buf, nullptr) == v) break;
  }
  // buf = [-]d[.ddd]e[+-]XX
  std::string s(buf);
  bool neg = s[0] == '-';
  if (neg) s.erase(0, 1);
  size_t epos = s.find('e');
  int exp10 = atoi(s.c_str() + epos + 1);
  std::string digits;
  for (size_t i = 0; i < epos; i++)
    if (s[i] != '.') digits.push_back(s[i]);
  while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
  std::string out;
  int nd = (int)digits.size();
  if (exp10 >= nd - 1) {
    out = digits + std::string((size_t)(exp10 - (nd - 1)), '0');
  } else if (exp10 >= 0) {
    out = digits.substr(0, (size_t)exp10 + 1) + "." + digits.substr((size_t)exp10 + 1);
  } else {
    out = "0." + std::string((size_t)(-exp10 - 1), '0') + digits;
  }
  return neg ? "-" + out : out;
}

// Rust's `{:?}` for f64 (core::fmt::float::float_to_general_debug): shortest round-trip digits with at least one
// fractional digit, scientific notation from 1e16 up and below 1e-4
static std::string rust_f64_debug(double v) {
  if (isnan(v) || isinf(v)) return rust_f64(v);
  const double abs = fabs(v);
  if (abs >= 1e16 || (abs != 0.0 && abs < 1e-4)) {
    char buf[64];
    for (int prec = 1; prec <= 17; prec++) {
      snprintf(buf, sizeof(buf), "%.*e", prec - 1, v);
      if (strtod(buf, nullptr) == v) break;
    }
    std::string t(buf);
    const size_t epos = t.find('e');
    std::string mant = t.substr(0, epos);
    if (mant.find('.') != std::string::npos) {
      while (mant.back() == '0') mant.pop_back();
      if (mant.back() == '.') mant.pop_back();
    }
    return mant + "e" + std::to_string(atoi(t.c_str() + epos + 1));
  }
  std::string d = rust_f64(v);
  if (d.find('.') == std::string::npos) d += ".0";
  return d;
}

static std::string fixed(double v, int prec) {
  char buf[64];
  snprintf(buf, sizeof(buf), "%.*f", prec, v);
  return buf;
}

// constraints/assertion.rs:48-76
bool Assertion::evaluate(double value) const {
  const double EPSILON = 1e-10;
  switch (kind) {
    case Equals: return fabs(value - a) < EPSILON;
    case NotEquals: return fabs(value - a) >= EPSILON;
    case GreaterThan: return value > a;
    case GreaterThanOrEqual: return value >= a;
    case LessThan: return value < a;
    case LessThanOrEqual: return value <= a;
    case Between: return value >= a && value <= b;
    case NotBetween: return value < a || value > b;
  }
  return false;
}

std::string Assertion::description() const {
  switch (kind) {
    case Equals: return "equals " + rust_f64(a);
    case NotEquals: return "not equals " + rust_f64(a);
    case GreaterThan: return "greater than " + rust_f64(a);
    case GreaterThanOrEqual: return "greater than or equal to " + rust_f64(a);
    case LessThan: return "less than " + rust_f64(a);
    case LessThanOrEqual: return "less than or equal to " + rust_f64(a);
    case Between: return "between " + rust_f64(a) + " and " + rust_f64(b);
    case NotBetween: return "not between " + rust_f64(a) + " and " + rust_f64(b);
  }
  return "";
}

// core/logical.rs:69-100
bool LogicalOperator::evaluate(const std::vector<bool> &results) const {
  if (results.empty()) {
    switch (kind) {
      case All: return true;
      case Any: return false;
      case Exactly: return n == 0;
      case AtLeast: return n == 0;
      case AtMost: return true;
    }
  }
  size_t t = 0;
  for (bool b : results) t += b ? 1 : 0;
  switch (kind) {
    case All: return t == results.size();
    case Any: return t > 0;
    case Exactly: return t == n;
    case AtLeast: return t >= n;
    case AtMost: return t <= n;
  }
  return false;
}

std::string LogicalOperator::description() const {
  switch (kind) {
    case All: return "all";
    case Any: return "any";
    case Exactly: return "exactly " + std::to_string(n);
    case AtLeast: return "at least " + std::to_string(n);
    case AtMost: return "at most " + std::to_string(n);
  }
  return "";
}

// security.rs:103-146, 212-255
std::optional<TermError> validate_identifier(const std::string &id) {
  auto sec = [](std::string m) { return TermError{TermError::SecurityError, std::move(m)}; };
  if (id.empty()) return sec("SQL identifier cannot be empty");
  if (id.size() > 128) return sec("SQL identifier too long (max 128 characters)");
  if (id.find('\0') != std::string::npos) return sec("SQL identifier cannot contain null bytes");
  // ^[a-zA-Z_"][a-zA-Z0-9_"]*(\.[a-zA-Z_"][a-zA-Z0-9_"]*)*$
  auto head = [](char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_' || c == '"'; };
  auto tail = [&](char c) { return head(c) || (c >= '0' && c <= '9'); };
  bool ok = true, at_start = true;
  for (char c : id) {
    if (at_start) {
      if (!head(c)) ok = false;
      at_start = false;
    } else if (c == '.') {
      at_start = true;
    } else if (!tail(c)) {
      ok = false;
    }
  }
  if (at_start) ok = false;  // trailing dot / empty segment
  if (!ok)
    return sec("Invalid SQL identifier format: '" + id +
               "'. Identifiers must start with a letter or underscore and contain only letters, numbers, "
               "underscores, and dots");
  std::string lower = id;
  for (auto &c : lower) c = (char)tolower((unsigned char)c);
  for (const char *p : {";", "--", "/*", "*/"})
    if (lower.find(p) != std::string::npos)
      return sec(std::string("SQL identifier contains dangerous character sequence: '") + p + "'");
  if (lower.rfind("xp_", 0) == 0 || lower.rfind("sp_", 0) == 0)
    return sec("SQL identifier looks like a system stored procedure");
  static const char *const inj[] = {"union ", "union_", "select ", "select_", "insert ", "insert_", "update ",
                                    "update_", "delete ", "delete_", "drop ", "drop_", "create ", "alter ", "exec ",
                                    "execute ", "declare ", "cursor ", "fetch ", "open ", "close "};
  for (const char *p : inj)
    if (lower.find(p) != std::string::npos) {
      std::string kw(p);
      while (!kw.empty() && kw.back() == '_') kw.pop_back();
      while (!kw.empty() && kw.back() == ' ') kw.pop_back();
      return sec("SQL identifier contains suspicious SQL keyword pattern: '" + kw + "'");
    }
  return {};
}

static void require_identifier(const std::string &id) {
  auto e = validate_identifier(id);
  if (e) throw *e;
}

// ------------------------------------------------------------------------------------------------ format types
std::string FormatType::get_pattern() const {
  switch (kind) {
    case Regex: return pattern;
    case Email:
      return R"(^[a-zA-Z0-9.!#$%&'*+/=?^_`{|}~-]+@[a-zA-Z0-9](?:[a-zA-Z0-9-]{0,61}[a-zA-Z0-9])?(?:\.[a-zA-Z0-9](?:[a-zA-Z0-9-]{0,61}[a-zA-Z0-9])?)*$)";
    case Url:
      return allow_localhost
                 ? R"(^https?://(?:localhost|(?:[a-zA-Z0-9.-]+\.?[a-zA-Z]{2,}|(?:\d{1,3}\.){3}\d{1,3}))(?::\d+)?(?:/[^\s]*)?$)"
                 : R"(^https?://[a-zA-Z0-9.-]+\.[a-zA-Z]{2,}(?::\d+)?(?:/[^\s]*)?$)";
    case CreditCard:
      return R"(^(?:4[0-9]{12}(?:[0-9]{3})?|5[1-5][0-9]{14}|3[47][0-9]{13}|3[0-9]{13}|6(?:011|5[0-9]{2})[0-9]{12})$|^(?:\d{4}[-\s]?){3}\d{4}$)";
    case Phone: {
      const std::string c = country.value_or("");
      if (c == "US" || c == "CA") return R"(^(\+?1[-.\s]?)?\(?([0-9]{3})\)?[-.\s]?([0-9]{3})[-.\s]?([0-9]{4})$)";
      if (c == "UK") return R"(^(\+44\s?)?(?:\(?0\d{4}\)?\s?\d{6}|\(?0\d{3}\)?\s?\d{7}|\(?0\d{2}\)?\s?\d{8})$)";
      if (c == "DE") return R"(^(\+49\s?)?(?:\(?0\d{2,5}\)?\s?\d{4,12})$)";
      if (c == "FR") return R"(^(\+33\s?)?(?:\(?0\d{1}\)?\s?\d{8})$)";
      return R"(^[\+]?[1-9][\d]{0,15}$)";
    }
    case PostalCode: {
      const std::string c = country.value_or("");
      if (c == "US") return R"(^\d{5}(-\d{4})?$)";
      if (c == "CA") return R"(^[A-Za-z]\d[A-Za-z][ -]?\d[A-Za-z]\d$)";
      if (c == "UK") return R"(^[A-Z]{1,2}\d[A-Z\d]?\s?\d[A-Z]{2}$)";
      if (c == "DE" || c == "FR") return R"(^\d{5}$)";
      if (c == "JP") return R"(^\d{3}-\d{4}$)";
      if (c == "AU") return R"(^\d{4}$)";
      return R"(^[A-Za-z0-9\s-]{3,10}$)";
    }
    case UUID:
      return R"(^[0-9a-fA-F]{8}-[0-9a-fA-F]{4}-[1-5][0-9a-fA-F]{3}-[89abAB][0-9a-fA-F]{3}-[0-9a-fA-F]{12}$)";
    case IPv4:
      return R"(^(?:(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.){3}(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)$)";
    case IPv6:
      return R"(^([0-9a-fA-F]{0,4}:){1,7}([0-9a-fA-F]{0,4})?$|^::$|^::1$|^([0-9a-fA-F]{1,4}:)*::([0-9a-fA-F]{1,4}:)*[0-9a-fA-F]{1,4}$)";
    case Json: return R"(^\s*[\{\[].*[\}\]]\s*$)";
    case Iso8601DateTime:
      return R"(^\d{4}-\d{2}-\d{2}T\d{2}:\d{2}:\d{2}(?:\.\d+)?(?:Z|[+-]\d{2}:\d{2})$)";
    case SocialSecurityNumber:
      return R"(^(00[1-9]|0[1-9][0-9]|[1-5][0-9]{2}|6[0-5][0-9]|66[0-5]|667|66[89]|6[7-9][0-9]|[7-8][0-9]{2})-?(0[1-9]|[1-9][0-9])-?(000[1-9]|00[1-9][0-9]|0[1-9][0-9]{2}|[1-9][0-9]{3})$)";
  }
  return "";
}

std::string FormatType::name() const {
  static const char *const n[] = {"regex", "email", "url", "credit_card", "phone", "postal_code", "uuid", "ipv4",
                                  "ipv6", "json", "iso8601_datetime", "social_security_number"};
  return n[(int)kind];
}

std::string FormatType::description() const {
  switch (kind) {
    case Regex: return "matches pattern '" + pattern + "'";
    case Email: return "are valid email addresses";
    case Url: return allow_localhost ? "are valid URLs (including localhost)" : "are valid URLs";
    case CreditCard: return detect_only ? "contain credit card number patterns" : "are valid credit card numbers";
    case Phone: return country ? "are valid " + *country + " phone numbers" : "are valid phone numbers";
    case PostalCode: return "are valid " + country.value_or("") + " postal codes";
    case UUID: return "are valid UUIDs";
    case IPv4: return "are valid IPv4 addresses";
    case IPv6: return "are valid IPv6 addresses";
    case Json: return "are valid JSON documents";
    case Iso8601DateTime: return "are valid ISO 8601 date-time strings";
    case SocialSecurityNumber: return "contain Social Security Number patterns";
  }
  return "";
}

std::string UniquenessType::name() const {
  switch (kind) {
    case FullUniqueness: return "full_uniqueness";
    case Distinctness: return "distinctness";
    case UniqueValueRatio: return "unique_value_ratio";
    case PrimaryKey: return "primary_key";
    case UniqueWithNulls: return "unique_with_nulls";
  }
  return "";
}

std::string StatisticType::name() const {
  switch (kind) {
    case Min: return "minimum";
    case Max: return "maximum";
    case Mean: return "mean";
    case Sum: return "sum";
    case StandardDeviation: return "standard deviation";
    case Variance: return "variance";
    case Median: return "median";
    case Percentile: return fabs(p - 0.5) < 2.220446049250313e-16 ? "median" : "percentile";
  }
  return "";
}

std::string StatisticType::constraint_name() const {
  static const char *const n[] = {"min", "max", "mean", "sum", "standard_deviation", "variance", "median", "percentile"};
  return n[(int)kind];
}

static std::string join(const std::vector<std::string> &v, const char *sep) {
  std::string o;
  for (size_t i = 0; i < v.size(); i++) {
    if (i) o += sep;
    o += v[i];
  }
  return o;
}

static void require_threshold(double t) {
  if (!(t >= 0.0 && t <= 1.0)) throw TermError{TermError::SecurityError, "Threshold must be between 0.0 and 1.0"};
}

// ------------------------------------------------------------------------------------------------ constraints
// ---- the reference's result-type rule (term_guard.h) ----
static bool is_one_of(const std::string &t, std::initializer_list<const char *> names) {
  for (const char *n : names)
    if (t == n) return true;
  return false;
}
bool reference_extracts(StatisticResultKind stat, const std::string &t) {
  if (t.empty() || t == "Int64" || t == "Float64") return true;
  const bool sint = is_one_of(t, {"Int8", "Int16", "Int32"});
  const bool uint = is_one_of(t, {"UInt8", "UInt16", "UInt32", "UInt64"});
  const bool flt = is_one_of(t, {"Float16", "Float32"});
  switch (stat) {
    case StatisticResultKind::Min:
    case StatisticResultKind::Max:
    case StatisticResultKind::Quantile:
      return false;  // the aggregate keeps the column's type: neither Int64Array nor Float64Array
    case StatisticResultKind::Sum:
      return sint || flt;  // Int64 / Float64; unsigned sums are UInt64
    case StatisticResultKind::Mean:
    case StatisticResultKind::StandardDeviation:
    case StatisticResultKind::Variance:
      return sint || uint || flt;  // Float64
  }
  return false;
}
bool reference_extracts_quantile(const std::string &t) {
  return t.empty() || t == "Int64" || t == "Float64" || t == "Int32";
}

namespace {

// constraints/size.rs:60-120
class SizeConstraint : public Constraint {
 public:
  explicit SizeConstraint(Assertion a) : a_(a) {}
  std::string name() const override { return "size"; }
  std::vector<SpecRequest> plan() const override {
    SpecRequest r;
    r.kind = TGX_CHECK_COUNT;  // COUNT(*) rides on any column; "" = the table's first column
    return {r};
  }
  ConstraintResult evaluate(const Inputs &in) const override {
    const double rows = (double)in.results[0]->total;
    if (a_.evaluate(rows)) return ConstraintResult::success_with_metric(rows);
    return ConstraintResult::failure_with_metric(rows, "Size " + rust_f64(rows) + " does not " + a_.description());
  }

 private:
  Assertion a_;
};

// constraints/completeness.rs:80-250 + core/unified.rs:41-123
class CompletenessConstraint : public Constraint {
 public:
  CompletenessConstraint(std::vector<std::string> cols, LogicalOperator op, double threshold)
      : cols_(std::move(cols)), op_(op), threshold_(threshold) {
    if (!(threshold >= 0.0 && threshold <= 1.0))
      throw TermError{TermError::Internal, "Threshold must be between 0.0 and 1.0"};  // completeness.rs:82-85 panics
  }
  std::string name() const override { return "completeness"; }
  std::optional<std::string> column() const override {
    return cols_.size() == 1 ? std::optional<std::string>(cols_[0]) : std::nullopt;
  }
  std::vector<SpecRequest> plan() const override {
    std::vector<SpecRequest> out;
    for (auto &c : cols_) {
      require_identifier(c);
      SpecRequest r;
      r.kind = TGX_CHECK_COUNT;
      r.column = c;
      out.push_back(r);
    }
    return out;
  }
  ConstraintResult column_result(const std::string &col, const tgx_result *r) const {
    if (r->total == 0) return ConstraintResult::skipped("No data to validate");
    const double completeness = (double)r->non_null / (double)r->total;
    if (completeness >= threshold_) return ConstraintResult::success_with_metric(completeness);
    return ConstraintResult::failure_with_metric(
        completeness, "Column '" + col + "' completeness " + fixed(completeness * 100.0, 2) +
                          "% is below threshold " + fixed(threshold_ * 100.0, 2) + "%");
  }
  ConstraintResult evaluate(const Inputs &in) const override {
    if (cols_.empty()) return ConstraintResult::skipped("No columns specified");
    if (cols_.size() == 1) return column_result(cols_[0], in.results[0]);
    std::vector<bool> oks;
    std::vector<double> metrics;
    for (size_t i = 0; i < cols_.size(); i++) {
      ConstraintResult r = column_result(cols_[i], in.results[i]);
      oks.push_back(r.status == ConstraintStatus::Success);
      if (r.metric) metrics.push_back(*r.metric);
    }
    const bool combined = op_.evaluate(oks);
    ConstraintResult out;
    if (!metrics.empty()) {
      double s = 0;
      for (double m : metrics) s += m;
      out.metric = s / (double)metrics.size();
    }
    if (combined) {
      out.status = ConstraintStatus::Success;
      if (op_.kind == LogicalOperator::All) {
        out.message = "All " + std::to_string(cols_.size()) + " columns satisfy the constraint";
      } else if (op_.kind == LogicalOperator::Any) {
        std::vector<std::string> passed;
        for (size_t i = 0; i < cols_.size(); i++)
          if (oks[i]) passed.push_back(cols_[i]);
        out.message = "Columns " + join(passed, ", ") + " satisfy the constraint";
      }
    } else {
      out.status = ConstraintStatus::Failure;
      std::vector<std::string> failed;
      for (size_t i = 0; i < cols_.size(); i++)
        if (!oks[i]) failed.push_back(cols_[i]);
      out.message = "Constraint failed for columns: " + join(failed, ", ") + ". Required: " + op_.description();
    }
    return out;
  }

 private:
  std::vector<std::string> cols_;
  LogicalOperator op_;
  double threshold_;
};

constexpr uint32_t kDefaultKllK = 200;

// the value one statistic reads out of its aggregate (statistics.rs:278-308: Float64, else Int64 cast to f64);
// false = SQL NULL.  `request` indexes the constraint's plan(): a NUMERIC_STATS request, or a KLL request for
// Median / Percentile (APPROX_PERCENTILE_CONT in the reference, the KLL sketch here)
static StatisticResultKind result_kind_of(const StatisticType &st) {
  switch (st.kind) {
    case StatisticType::Min: return StatisticResultKind::Min;
    case StatisticType::Max: return StatisticResultKind::Max;
    case StatisticType::Mean: return StatisticResultKind::Mean;
    case StatisticType::Sum: return StatisticResultKind::Sum;
    case StatisticType::StandardDeviation: return StatisticResultKind::StandardDeviation;
    case StatisticType::Variance: return StatisticResultKind::Variance;
    default: return StatisticResultKind::Quantile;
  }
}
static bool extractable(const StatisticType &st, const Constraint::Inputs &in, size_t request) {
  if (!in.strict_reference_types || request >= in.arrow_types.size()) return true;
  return reference_extracts(result_kind_of(st), in.arrow_types[request]);
}

static bool statistic_value(const StatisticType &st, const Constraint::Inputs &in, size_t request, double *value) {
  const tgx_result *r = in.results[request];
  switch (st.kind) {
    case StatisticType::Min: *value = r->min_f; return r->has_value != 0;
    case StatisticType::Max: *value = r->max_f; return r->has_value != 0;
    case StatisticType::Mean: *value = r->mean; return r->has_value != 0;
    case StatisticType::Sum: *value = r->is_float ? r->sum_f : (double)r->sum_i; return r->has_value != 0;
    case StatisticType::StandardDeviation: *value = r->stddev_samp; return r->has_variance != 0;
    case StatisticType::Variance: *value = r->var_samp; return r->has_variance != 0;
    case StatisticType::Median:
    case StatisticType::Percentile:
      if (r->kll_n == 0) return false;
      *value = in.quantile(in.ctx, request, st.kind == StatisticType::Median ? 0.5 : st.p);
      return true;
  }
  return false;
}

// constraints/statistics.rs:174-330
class StatisticalConstraint : public Constraint {
 public:
  StatisticalConstraint(std::string col, StatisticType st, Assertion a) : col_(std::move(col)), st_(st), a_(a) {
    require_identifier(col_);
    if (st_.kind == StatisticType::Percentile && !(st_.p >= 0.0 && st_.p <= 1.0))
      throw TermError{TermError::SecurityError, "Percentile must be between 0.0 and 1.0"};
  }
  std::string name() const override { return st_.constraint_name(); }
  std::optional<std::string> column() const override { return col_; }
  bool is_quantile() const { return st_.kind == StatisticType::Median || st_.kind == StatisticType::Percentile; }
  std::vector<SpecRequest> plan() const override {
    SpecRequest r;
    r.column = col_;
    if (is_quantile()) {
      // APPROX_PERCENTILE_CONT in the reference (t-digest); here the KLL sketch (DESIGN.md "Quantiles")
      r.kind = TGX_CHECK_KLL;
      r.kll_k = kDefaultKllK;
    } else {
      r.kind = TGX_CHECK_NUMERIC_STATS;
      if (st_.kind == StatisticType::StandardDeviation || st_.kind == StatisticType::Variance) r.flags = TGX_FLAG_VARIANCE;
    }
    return {r};
  }
  ConstraintResult evaluate(const Inputs &in) const override {
    // statistics.rs:277-308: a result column that is neither Int64Array nor Float64Array
    if (!extractable(st_, in, 0)) throw TermError{TermError::Internal, "Failed to extract statistic value"};
    double value = 0;
    const bool null = !statistic_value(st_, in, 0, &value);
    if (null) return ConstraintResult::failure(st_.name() + " is null (no non-null values)");  // statistics.rs:284-301
    if (a_.evaluate(value)) return ConstraintResult::success_with_metric(value);
    return ConstraintResult::failure_with_metric(value,
                                                 st_.name() + " " + rust_f64(value) + " does not " + a_.description());
  }

 private:
  std::string col_;
  StatisticType st_;
  Assertion a_;
};

// constraints/statistics.rs:376-531: several statistics of one column from ONE query; the metric of a success is the
// first statistic's value (:498-500), failures are joined by "; " and carry no metric (:501-503)
class MultiStatisticalConstraint : public Constraint {
 public:
  MultiStatisticalConstraint(std::string col, std::vector<std::pair<StatisticType, Assertion>> stats)
      : col_(std::move(col)), stats_(std::move(stats)) {
    require_identifier(col_);
    for (auto &sa : stats_)
      if (sa.first.kind == StatisticType::Percentile && !(sa.first.p >= 0.0 && sa.first.p <= 1.0))
        throw TermError{TermError::SecurityError, "Percentile must be between 0.0 and 1.0"};  // :399-407
    for (auto &sa : stats_) {
      const bool q = sa.first.kind == StatisticType::Median || sa.first.kind == StatisticType::Percentile;
      (q ? any_quantile_ : any_plain_) = true;
      if (sa.first.kind == StatisticType::StandardDeviation || sa.first.kind == StatisticType::Variance) variance_ = true;
    }
  }
  std::string name() const override { return "multi_statistical"; }
  std::optional<std::string> column() const override { return col_; }
  std::vector<SpecRequest> plan() const override {
    std::vector<SpecRequest> out;
    if (any_plain_) {
      SpecRequest r;
      r.kind = TGX_CHECK_NUMERIC_STATS;
      r.column = col_;
      r.flags = variance_ ? (uint32_t)TGX_FLAG_VARIANCE : 0u;
      out.push_back(r);
    }
    if (any_quantile_) {
      SpecRequest r;
      r.kind = TGX_CHECK_KLL;
      r.column = col_;
      r.kll_k = kDefaultKllK;
      out.push_back(r);
    }
    return out;
  }
  ConstraintResult evaluate(const Inputs &in) const override {
    std::vector<std::string> failures;
    std::optional<double> first_metric;
    for (auto &sa : stats_) {
      const bool q = sa.first.kind == StatisticType::Median || sa.first.kind == StatisticType::Percentile;
      double value = 0;
      if (!extractable(sa.first, in, q && any_plain_ ? 1 : 0)) {
        failures.push_back("Failed to compute " + sa.first.name());  // statistics.rs:478-482
        continue;
      }
      if (!statistic_value(sa.first, in, q && any_plain_ ? 1 : 0, &value)) {
        failures.push_back(sa.first.name() + " is null");  // :466-470
        continue;
      }
      if (!first_metric) first_metric = value;
      if (!sa.second.evaluate(value))
        failures.push_back(sa.first.name() + " is " + rust_f64(value) + " which does not " + sa.second.description());
    }
    if (failures.empty()) return ConstraintResult::success_with_metric(first_metric ? *first_metric : 0.0);
    return ConstraintResult::failure(join(failures, "; "));
  }

 private:
  std::string col_;
  std::vector<std::pair<StatisticType, Assertion>> stats_;
  bool any_plain_ = false, any_quantile_ = false, variance_ = false;
};

// constraints/uniqueness.rs:380-860 (single-column forms)
class UniquenessConstraint : public Constraint {
 public:
  UniquenessConstraint(std::vector<std::string> cols, UniquenessType t) : cols_(std::move(cols)), t_(t) {
    if (cols_.empty()) throw TermError{TermError::Internal, "At least one column must be specified"};
    for (auto &c : cols_) require_identifier(c);
    if (t_.kind == UniquenessType::FullUniqueness || t_.kind == UniquenessType::UniqueWithNulls) require_threshold(t_.threshold);
  }
  std::string name() const override { return t_.name(); }
  std::optional<std::string> column() const override {
    return cols_.size() == 1 ? std::optional<std::string>(cols_[0]) : std::nullopt;
  }
  std::vector<SpecRequest> plan() const override {
    if (cols_.size() > 8)
      throw TermError{TermError::NotSupported, "uniqueness over more than 8 columns is not on the GPU path"};
    SpecRequest r;
    r.kind = TGX_CHECK_DISTINCT;
    r.column = cols_[0];
    // COUNT(DISTINCT (a, b)) / GROUP BY a, b (uniqueness.rs:557-562, 687-699): the tuple is one value.  The
    // multi-column Distinctness SQL concatenates COALESCE(CAST(c AS VARCHAR), '<NULL>') with '|' (:643-647); the
    // tuple gives the same count unless values contain '|' or the literal '<NULL>' (where the concatenation
    // conflates different rows).
    if (cols_.size() >= 2) r.columns = cols_;
    if (t_.kind == UniquenessType::UniqueValueRatio) r.flags = TGX_FLAG_MULTIPLICITY;
    return {r};
  }
  ConstraintResult evaluate(const Inputs &in) const override {
    const tgx_result *r = in.results[0];
    const double total = (double)r->total;
    if (total == 0.0) return ConstraintResult::skipped("No data to validate");
    const double nulls = (double)(r->total - r->non_null);
    const std::string cols = join(cols_, ", ");
    switch (t_.kind) {
      case UniquenessType::FullUniqueness:
      case UniquenessType::UniqueWithNulls: {
        double unique = (double)r->distinct;
        if (t_.kind == UniquenessType::UniqueWithNulls && cols_.size() == 1) {  // multi-column: plain tuple count (:579-585, 600-607)
          if (t_.null_handling == NullHandling::Include) unique += nulls > 0 ? 1.0 : 0.0;  // COALESCE(c, '<NULL>')
          if (t_.null_handling == NullHandling::Distinct) unique += nulls;                 // :594-598
        }
        const double ratio = unique / total;
        if (ratio >= t_.threshold) return ConstraintResult::success_with_metric(ratio);
        return ConstraintResult::failure_with_metric(
            ratio, "Uniqueness ratio " + fixed(ratio, 3) + " is below threshold " + fixed(t_.threshold, 3) +
                       " for columns: " + cols);
      }
      case UniquenessType::Distinctness:
      case UniquenessType::UniqueValueRatio: {
        const double count = t_.kind == UniquenessType::Distinctness ? (double)r->distinct : (double)r->groups_once;
        const double ratio = count / total;
        if (t_.assertion->evaluate(ratio)) return ConstraintResult::success_with_metric(ratio);
        return ConstraintResult::failure_with_metric(
            ratio, t_.name() + " ratio " + fixed(ratio, 3) + " does not satisfy " + t_.assertion->description() +
                       " for columns: " + cols);
      }
      case UniquenessType::PrimaryKey: {
        const double unique = (double)r->distinct;
        if (nulls > 0.0)
          return ConstraintResult::failure_with_metric(
              nulls / total, "Primary key columns contain " + rust_f64(nulls) + " NULL values: " + cols);
        if (unique != total)
          return ConstraintResult::failure_with_metric(
              (total - unique) / total,
              "Primary key columns contain " + rust_f64(total - unique) + " duplicate values: " + cols);
        return ConstraintResult::success_with_metric(1.0);
      }
    }
    return ConstraintResult::success();
  }

 private:
  std::vector<std::string> cols_;
  UniquenessType t_;
};

// constraints/format.rs:490-845
class FormatConstraint : public Constraint {
 public:
  FormatConstraint(std::string col, FormatType f, double threshold, FormatOptions o)
      : col_(std::move(col)), f_(std::move(f)), threshold_(threshold), o_(o) {
    require_identifier(col_);
    require_threshold(threshold_);
    // format.get_pattern() validates FormatType::Regex through SqlSecurity::validate_regex_pattern
    const std::string pat = f_.get_pattern();
    tgx_error err;
    memset(&err, 0, sizeof(err));
    const uint32_t flags = o_.case_sensitive ? 0u : (uint32_t)TGX_FLAG_CASE_INSENSITIVE;
    tgx_status s = tgx_regex_validate(pat.data(), pat.size(), flags, &err);
    if (s == TGX_INVALID_ARGUMENT) throw TermError{TermError::SecurityError, err.msg};
    if (s != TGX_OK) throw TermError{TermError::NotSupported, err.msg};
  }
  std::string name() const override { return f_.name(); }
  std::optional<std::string> column() const override { return col_; }
  std::vector<SpecRequest> plan() const override {
    SpecRequest r;
    r.kind = TGX_CHECK_REGEX_MATCH;
    r.column = col_;
    r.pattern = f_.get_pattern();
    r.flags = (o_.case_sensitive ? 0u : (uint32_t)TGX_FLAG_CASE_INSENSITIVE) |
              (o_.trim_before_check ? (uint32_t)TGX_FLAG_TRIM : 0u) |
              (o_.null_is_valid ? (uint32_t)TGX_FLAG_NULL_IS_VALID : 0u);
    return {r};
  }
  ConstraintResult evaluate(const Inputs &in) const override {
    const tgx_result *r = in.results[0];
    const double total = (double)r->total;
    if (total == 0.0) return ConstraintResult::skipped("No data to validate");
    const double ratio = (double)r->matches / total;
    const bool detect = f_.kind == FormatType::CreditCard && f_.detect_only;
    const bool ok = detect ? ratio <= threshold_ : ratio >= threshold_;
    if (ok) return ConstraintResult::success_with_metric(ratio);
    if (detect)
      return ConstraintResult::failure_with_metric(
          ratio, "Credit card detection ratio " + fixed(ratio, 3) + " exceeds threshold " + fixed(threshold_, 3));
    return ConstraintResult::failure_with_metric(
        ratio, "Format validation ratio " + fixed(ratio, 3) + " is below threshold " + fixed(threshold_, 3) +
                   " - values that " + f_.description());
  }

 private:
  std::string col_;
  FormatType f_;
  double threshold_;
  FormatOptions o_;
};

// constraints/length.rs:19-231: COUNT(CASE WHEN <cond on LENGTH(c)> OR c IS NULL THEN 1 END) / COUNT(*) must be 1.0
class LengthConstraint : public Constraint {
 public:
  LengthConstraint(std::string col, std::string kind, uint64_t a, uint64_t b) : col_(std::move(col)), kind_(std::move(kind)), a_(a), b_(b) {
    require_identifier(col_);
    if (kind_ != "min" && kind_ != "max" && kind_ != "between" && kind_ != "exactly" && kind_ != "not_empty")
      throw TermError{TermError::Internal, "unknown length assertion '" + kind_ + "'"};
  }
  std::string name() const override {  // length.rs:48-56
    return kind_ == "min" ? "min_length" : kind_ == "max" ? "max_length" : kind_ == "between" ? "length_between"
           : kind_ == "exactly" ? "exact_length" : "not_empty";
  }
  std::string description() const {  // :59-67
    if (kind_ == "min") return "at least " + std::to_string(a_) + " characters";
    if (kind_ == "max") return "at most " + std::to_string(a_) + " characters";
    if (kind_ == "between") return "between " + std::to_string(a_) + " and " + std::to_string(b_) + " characters";
    if (kind_ == "exactly") return "exactly " + std::to_string(a_) + " characters";
    return "not empty";
  }
  std::optional<std::string> column() const override { return col_; }
  std::vector<SpecRequest> plan() const override {
    SpecRequest r;
    r.kind = TGX_CHECK_LENGTH;
    r.column = col_;
    r.length_min = kind_ == "min" || kind_ == "between" || kind_ == "exactly" ? a_ : kind_ == "not_empty" ? 1 : 0;
    r.length_max = kind_ == "max" || kind_ == "exactly" ? a_ : kind_ == "between" ? b_ : ~0ull;
    return {r};
  }
  ConstraintResult evaluate(const Inputs &in) const override {
    const tgx_result *r = in.results[0];
    if (r->total == 0) return ConstraintResult::skipped("No data to validate");  // NULLIF(COUNT(*), 0) (:167-193)
    const double ratio = (double)r->matches * 1.0 / (double)r->total;
    if (ratio >= 1.0) return ConstraintResult::success_with_metric(ratio);
    return ConstraintResult::failure_with_metric(
        ratio, "Length constraint failed: " + fixed(ratio * 100.0, 2) + "% of values are " + description());
  }

 private:
  std::string col_, kind_;
  uint64_t a_, b_;
};

// constraints/values.rs:200-330: COUNT(CASE WHEN c IN ('a', 'b', ..) THEN 1 END) / COUNT(*) WHERE c IS NOT NULL == 1.0.
// The IN-list is matched as the anchored alternation ^(?:a|b|..)$ of the escaped literals by the pattern kernel
// (Rust's `$` is the end of the text, so this is exact string equality).
class ContainmentConstraint : public Constraint {
 public:
  ContainmentConstraint(std::string col, std::vector<std::string> allowed) : col_(std::move(col)), allowed_(std::move(allowed)) {
    if (allowed_.empty()) throw TermError{TermError::Internal, "containment needs at least one allowed value"};
  }
  std::string name() const override { return "containment"; }
  std::optional<std::string> column() const override { return col_; }
  static std::string escape(const std::string &lit) {  // regex::escape
    std::string o;
    for (char ch : lit) {
      if (strchr("\\.+*?()|[]{}^$#&-~", ch) && ch != 0) o.push_back('\\');
      o.push_back(ch);
    }
    return o;
  }
  std::vector<SpecRequest> plan() const override {
    SpecRequest cnt;
    cnt.kind = TGX_CHECK_COUNT;
    cnt.column = col_;
    SpecRequest m;
    m.kind = TGX_CHECK_REGEX_MATCH;
    m.column = col_;
    m.pattern = "^(?:";
    for (size_t i = 0; i < allowed_.size(); i++) m.pattern += (i ? "|" : "") + escape(allowed_[i]);
    m.pattern += ")$";
    m.flags = 0;  // NULL rows are outside the WHERE clause
    return {cnt, m};
  }
  ConstraintResult evaluate(const Inputs &in) const override {
    const double total = (double)in.results[0]->non_null, valid = (double)in.results[1]->matches;
    if (total == 0.0) return ConstraintResult::skipped("No non-null data to validate");  // values.rs:274-276
    const double ratio = valid / total;
    if (ratio == 1.0) return ConstraintResult::success_with_metric(ratio);
    return ConstraintResult::failure_with_metric(ratio, rust_f64(total - valid) + " values are not in the allowed set");
  }

 private:
  std::string col_;
  std::vector<std::string> allowed_;
};

// constraints/approx_count_distinct.rs:53-120: `SELECT APPROX_DISTINCT(col)`.  DataFusion's sketch is a HyperLogLog of
// 2^14 registers over ahash values -- third-party arithmetic whose hash cannot be reproduced here; the library keeps a
// sketch of the same shape as one more lane of the column's scan (TGX_CHECK_APPROX_DISTINCT, include/tgx.h) with
// Ertl's estimator, the one DataFusion's `count()` uses.  The metric is that estimate (standard error 0.8 %; the
// reference's own tests only bound theirs, :186-347: within 3 %); on string / dictionary columns, and when the suite
// also asks for the exact count of the column, the library answers with the exact count.  An empty or all-NULL column
// gives 0, not Skipped (:299-325).
class ApproxCountDistinctConstraint : public Constraint {
 public:
  ApproxCountDistinctConstraint(std::string col, Assertion a) : col_(std::move(col)), a_(a) {}
  std::string name() const override { return "approx_count_distinct"; }
  std::optional<std::string> column() const override { return col_; }
  std::vector<SpecRequest> plan() const override {
    SpecRequest r;
    r.kind = TGX_CHECK_APPROX_DISTINCT;
    r.column = col_;
    return {r};
  }
  ConstraintResult evaluate(const Inputs &in) const override {
    const double count = (double)in.results[0]->distinct;
    if (a_.evaluate(count)) return ConstraintResult::success_with_metric(count);
    return ConstraintResult::failure_with_metric(count, "Approximate distinct count " + rust_f64(count) +
                                                            " does not satisfy assertion " + a_.description() +
                                                            " for column '" + col_ + "'");
  }

 private:
  std::string col_;
  Assertion a_;
};

// constraints/quantile.rs:144-497.  APPROX_PERCENTILE_CONT (DataFusion's t-digest) is the KLL sketch here: ONE
// sketch of the column answers every quantile of the constraint (the reference already asks for all of them in one
// query, :352-366).  Distribution / Custom have no evaluation in the reference either (:481-486).
class QuantileConstraint : public Constraint {
 public:
  QuantileConstraint(std::string col, QuantileValidation v) : col_(std::move(col)), v_(std::move(v)) {
    require_identifier(col_);
    auto in_range = [](double q) { return q >= 0.0 && q <= 1.0; };
    for (auto &c : v_.checks)
      if (!in_range(c.quantile)) throw TermError{TermError::Configuration, "Quantile must be between 0.0 and 1.0"};  // :47-52
    if (v_.kind == QuantileValidation::Multiple && v_.checks.empty())
      throw TermError{TermError::Configuration, "At least one quantile check is required"};  // :196-200
    if (v_.kind == QuantileValidation::Single && v_.checks.size() != 1)
      throw TermError{TermError::Internal, "a single quantile validation takes exactly one check"};
  }
  std::string name() const override { return "quantile"; }
  std::optional<std::string> column() const override { return col_; }
  std::vector<SpecRequest> plan() const override {
    SpecRequest r;
    r.kind = TGX_CHECK_KLL;
    r.column = col_;
    r.kll_k = kDefaultKllK;
    return {r};
  }
  ConstraintResult evaluate(const Inputs &in) const override {
    if (v_.kind == QuantileValidation::Distribution || v_.kind == QuantileValidation::Custom)
      return ConstraintResult::skipped("Validation type not yet implemented");  // :481-486
    if (in.results[0]->kll_n == 0) return ConstraintResult::skipped("No data to validate");
    // quantile.rs:308-324 (and :369-385, :433-449): APPROX_PERCENTILE_CONT keeps the column's type; what is not
    // Float64 / Int64 / Int32 is a TypeMismatch error whose `found` is the DataType's Debug form
    if (in.strict_reference_types && !in.arrow_types.empty() && !reference_extracts_quantile(in.arrow_types[0]))
      throw TermError{TermError::TypeMismatch, "expected Float64, Int64, or Int32, found " + in.arrow_types[0]};
    switch (v_.kind) {
      case QuantileValidation::Single: {  // :287-345
        const QuantileCheck &c = v_.checks[0];
        const double value = in.quantile(in.ctx, 0, c.quantile);
        if (c.assertion.evaluate(value)) return ConstraintResult::success_with_metric(value);
        return ConstraintResult::failure_with_metric(value, "Quantile " + rust_f64(c.quantile) + " is " + rust_f64(value) +
                                                                " which does not " + c.assertion.description());
      }
      case QuantileValidation::Multiple: {  // :346-420
        std::vector<std::string> failures;
        for (const QuantileCheck &c : v_.checks) {
          const double value = in.quantile(in.ctx, 0, c.quantile);
          if (!c.assertion.evaluate(value))
            failures.push_back("Q" + std::to_string((int32_t)(c.quantile * 100.0)) + " is " + rust_f64(value) +
                               " which does not " + c.assertion.description());
        }
        if (failures.empty()) return ConstraintResult::success();
        return ConstraintResult::failure(join(failures, "; "));
      }
      case QuantileValidation::Monotonic: {  // :421-480
        std::vector<double> values;
        for (double q : v_.quantiles) values.push_back(in.quantile(in.ctx, 0, q));
        bool monotonic = true;
        for (size_t i = 1; i < values.size() && monotonic; i++)
          monotonic = v_.strict ? values[i] > values[i - 1] : values[i] >= values[i - 1];
        if (monotonic) return ConstraintResult::success();
        std::string list = "[";
        for (size_t i = 0; i < values.size(); i++) list += (i ? ", " : "") + rust_f64_debug(values[i]);
        return ConstraintResult::failure(std::string("Quantiles are not ") + (v_.strict ? "strictly" : "") +
                                         " monotonic: " + list + "]");
      }
      default:
        break;
    }
    return ConstraintResult::skipped("Validation type not yet implemented");
  }

 private:
  std::string col_;
  QuantileValidation v_;
};

// constraints/correlation.rs:147-508.  Pearson (`CORR`), Covariance (`COVAR_SAMP`), Range (= Pairwise with Between,
// :376-396) and Independence (`ABS(CORR) <= max`, :397-439) all read the six co-moments of one COMOMENTS request.
// Spearman / Kendall / mutual information, MultiColumn and Stability are "not yet implemented" in the reference too
// (:336-341, :440-442).  A Custom SQL expression cannot run on this path: NotSupported, so the caller falls back to
// the stock SQL constraint (after the reference's own unsafe-content screen, :324-330).
class CorrelationConstraint : public Constraint {
 public:
  explicit CorrelationConstraint(CorrelationValidation v) : v_(std::move(v)) {
    if (v_.kind == CorrelationValidation::MultiColumn) {
      if (v_.columns.size() < 2)
        throw TermError{TermError::Configuration, "At least 2 columns required for correlation analysis"};  // :178-182
      for (auto &c : v_.columns) require_identifier(c);
    } else {
      require_identifier(v_.column1);
      require_identifier(v_.column2);
    }
    if (v_.kind == CorrelationValidation::Independence && !(v_.max_correlation >= 0.0 && v_.max_correlation <= 1.0))
      throw TermError{TermError::Configuration, "Max correlation must be between 0.0 and 1.0"};  // :253-257
  }
  std::string name() const override {  // :446-456
    switch (v_.kind) {
      case CorrelationValidation::Pairwise: return v_.type.constraint_name();
      case CorrelationValidation::Range: return "correlation_range";
      case CorrelationValidation::Independence: return "independence";
      case CorrelationValidation::MultiColumn: return "multi_correlation";
      default: return "correlation_stability";
    }
  }
  bool on_device() const {
    if (v_.kind == CorrelationValidation::Independence) return true;
    if (v_.kind != CorrelationValidation::Pairwise && v_.kind != CorrelationValidation::Range) return false;
    return v_.type.kind == CorrelationType::Pearson || v_.type.kind == CorrelationType::Covariance;
  }
  std::vector<SpecRequest> plan() const override {
    if ((v_.kind == CorrelationValidation::Pairwise || v_.kind == CorrelationValidation::Range) &&
        v_.type.kind == CorrelationType::Custom && !custom_is_unsafe())
      throw TermError{TermError::NotSupported, "custom correlation SQL expressions do not run on the GPU path"};
    if (!on_device()) return {};
    SpecRequest r;
    r.kind = TGX_CHECK_COMOMENTS;
    r.column = v_.column1;
    r.column2 = v_.column2;
    return {r};
  }
  ConstraintResult evaluate(const Inputs &in) const override {
    if (v_.kind == CorrelationValidation::MultiColumn || v_.kind == CorrelationValidation::Stability)
      return ConstraintResult::skipped("Validation type not yet implemented");
    if (v_.kind != CorrelationValidation::Independence) {
      if (v_.type.kind == CorrelationType::Custom)
        return ConstraintResult::failure("Custom SQL expression contains potentially unsafe content");
      if (!on_device()) return ConstraintResult::skipped("Correlation type not yet implemented");
    }
    const tgx_result *r = in.results[0];
    if (r->total == 0) return ConstraintResult::skipped("No data to validate");
    if (v_.kind == CorrelationValidation::Independence) {
      const double abs_corr = fabs(pearson(r));
      if (abs_corr <= v_.max_correlation) return ConstraintResult::success_with_metric(abs_corr);
      return ConstraintResult::failure_with_metric(
          abs_corr, "Columns " + v_.column1 + " and " + v_.column2 + " have correlation " + rust_f64(abs_corr) +
                        " exceeding independence threshold " + rust_f64(v_.max_correlation));
    }
    const Assertion a = v_.kind == CorrelationValidation::Range ? Assertion::between(v_.min, v_.max) : v_.assertion;
    const double value = v_.type.kind == CorrelationType::Covariance ? covar_samp(r) : pearson(r);
    if (a.evaluate(value)) return ConstraintResult::success_with_metric(value);
    return ConstraintResult::failure_with_metric(value, v_.type.name() + " between " + v_.column1 + " and " + v_.column2 +
                                                            " is " + rust_f64(value) + " which does not " + a.description());
  }

 private:
  bool custom_is_unsafe() const {  // :324-330
    std::string lower = v_.type.sql_expression;
    for (auto &c : lower) c = (char)tolower((unsigned char)c);
    return lower.find(';') != std::string::npos || lower.find("drop") != std::string::npos;
  }
  // DataFusion's CORR: population covariance over the population standard deviations, 0 when either deviation is
  // 0 (constraints/correlation.rs:260-275 emits `CORR(a, b)`; its accumulators are online -- Welford variances and
  // a co-moment -- so what they hold at the end are the CENTRED moments, which is what the library returns in
  // co_m2_x / co_m2_y / co_c_xy: on offset data the raw-sum form n Sxy - Sx Sy has lost its digits)
  static double pearson(const tgx_result *r) {
    const double n = (double)r->non_null;
    if (n < 1) return 0.0;
    const double cov = r->co_c_xy / n;
    const double sx = r->co_m2_x > 0 ? sqrt(r->co_m2_x / n) : 0.0, sy = r->co_m2_y > 0 ? sqrt(r->co_m2_y / n) : 0.0;
    return (sx == 0.0 || sy == 0.0) ? 0.0 : cov / sx / sy;
  }
  // COVAR_SAMP = Cxy / (n - 1); SQL NULL below two rows, which the reference reads as the raw slot value
  // (`value(0)` without a null check, :355-362 -- unpinned): 0.0 here
  static double covar_samp(const tgx_result *r) {
    const double n = (double)r->non_null;
    if (n < 2) return 0.0;
    return r->co_c_xy / (n - 1.0);
  }
  CorrelationValidation v_;
};

}  // namespace

// ------------------------------------------------------------------------------------------------ builders
Check::Builder Check::builder(std::string name) { return Builder(std::move(name)); }
ValidationSuite::Builder ValidationSuite::builder(std::string name) { return Builder(std::move(name)); }

Check::Builder &Check::Builder::has_size(Assertion a) { return constraint(std::make_shared<SizeConstraint>(a)); }
Check::Builder &Check::Builder::has_approx_count_distinct(std::string column, Assertion a) {
  return constraint(std::make_shared<ApproxCountDistinctConstraint>(std::move(column), a));
}
Check::Builder &Check::Builder::completeness(std::vector<std::string> columns, CompletenessOptions o) {
  return constraint(std::make_shared<CompletenessConstraint>(std::move(columns), o.op, o.threshold));
}
Check::Builder &Check::Builder::any_complete(std::vector<std::string> columns) {
  return constraint(std::make_shared<CompletenessConstraint>(std::move(columns), LogicalOperator{LogicalOperator::Any, 0}, 1.0));
}
Check::Builder &Check::Builder::at_least_complete(size_t n, std::vector<std::string> columns, double threshold) {
  return constraint(
      std::make_shared<CompletenessConstraint>(std::move(columns), LogicalOperator{LogicalOperator::AtLeast, n}, threshold));
}
Check::Builder &Check::Builder::exactly_complete(size_t n, std::vector<std::string> columns, double threshold) {
  return constraint(
      std::make_shared<CompletenessConstraint>(std::move(columns), LogicalOperator{LogicalOperator::Exactly, n}, threshold));
}
Check::Builder &Check::Builder::statistic(std::string column, StatisticType stat, Assertion a) {
  return constraint(std::make_shared<StatisticalConstraint>(std::move(column), stat, a));
}
Check::Builder &Check::Builder::uniqueness(std::vector<std::string> columns, UniquenessType type) {
  return constraint(std::make_shared<UniquenessConstraint>(std::move(columns), type));
}
Check::Builder &Check::Builder::validates_uniqueness(std::vector<std::string> columns, double threshold) {
  UniquenessType t;
  t.kind = UniquenessType::FullUniqueness;
  t.threshold = threshold;
  return uniqueness(std::move(columns), t);
}
Check::Builder &Check::Builder::validates_distinctness(std::vector<std::string> columns, Assertion a) {
  UniquenessType t;
  t.kind = UniquenessType::Distinctness;
  t.assertion = a;
  return uniqueness(std::move(columns), t);
}
Check::Builder &Check::Builder::validates_unique_value_ratio(std::vector<std::string> columns, Assertion a) {
  UniquenessType t;
  t.kind = UniquenessType::UniqueValueRatio;
  t.assertion = a;
  return uniqueness(std::move(columns), t);
}
Check::Builder &Check::Builder::validates_primary_key(std::vector<std::string> columns) {
  UniquenessType t;
  t.kind = UniquenessType::PrimaryKey;
  return uniqueness(std::move(columns), t);
}
Check::Builder &Check::Builder::validates_uniqueness_with_nulls(std::vector<std::string> columns, double threshold,
                                                                NullHandling h) {
  UniquenessType t;
  t.kind = UniquenessType::UniqueWithNulls;
  t.threshold = threshold;
  t.null_handling = h;
  return uniqueness(std::move(columns), t);
}
Check::Builder &Check::Builder::primary_key(std::vector<std::string> columns) {
  completeness(columns, CompletenessOptions::full());
  return validates_uniqueness(std::move(columns), 1.0);
}
Check::Builder &Check::Builder::is_contained_in(std::string column, std::vector<std::string> allowed) {
  return constraint(std::make_shared<ContainmentConstraint>(std::move(column), std::move(allowed)));
}
Check::Builder &Check::Builder::length(std::string column, std::string kind, uint64_t a, uint64_t b) {
  return constraint(std::make_shared<LengthConstraint>(std::move(column), std::move(kind), a, b));
}
Check::Builder &Check::Builder::has_format(std::string column, FormatType format, double threshold, FormatOptions o) {
  return constraint(std::make_shared<FormatConstraint>(std::move(column), std::move(format), threshold, o));
}
static FormatType ft(FormatType::Kind k) {
  FormatType f;
  f.kind = k;
  return f;
}
Check::Builder &Check::Builder::validates_regex(std::string column, std::string pattern, double threshold) {
  FormatType f = ft(FormatType::Regex);
  f.pattern = std::move(pattern);
  return has_format(std::move(column), f, threshold, FormatOptions());
}
Check::Builder &Check::Builder::validates_email(std::string column, double threshold) {
  return has_format(std::move(column), ft(FormatType::Email), threshold, FormatOptions());
}
Check::Builder &Check::Builder::validates_url(std::string column, double threshold, bool allow_localhost) {
  FormatType f = ft(FormatType::Url);
  f.allow_localhost = allow_localhost;
  return has_format(std::move(column), f, threshold, FormatOptions());
}
Check::Builder &Check::Builder::validates_credit_card(std::string column, double threshold, bool detect_only) {
  FormatType f = ft(FormatType::CreditCard);
  f.detect_only = detect_only;
  return has_format(std::move(column), f, threshold, FormatOptions());
}
static FormatOptions trimming() {
  FormatOptions o;
  o.trim_before_check = true;
  return o;
}
Check::Builder &Check::Builder::validates_phone(std::string column, double threshold, std::optional<std::string> country) {
  FormatType f = ft(FormatType::Phone);
  f.country = std::move(country);
  return has_format(std::move(column), f, threshold, trimming());  // format.rs:606-616
}
Check::Builder &Check::Builder::validates_postal_code(std::string column, double threshold, std::string country) {
  FormatType f = ft(FormatType::PostalCode);
  f.country = std::move(country);
  return has_format(std::move(column), f, threshold, trimming());  // format.rs:619-633
}
Check::Builder &Check::Builder::validates_uuid(std::string c, double t) { return has_format(std::move(c), ft(FormatType::UUID), t, {}); }
Check::Builder &Check::Builder::validates_ipv4(std::string c, double t) { return has_format(std::move(c), ft(FormatType::IPv4), t, {}); }
Check::Builder &Check::Builder::validates_ipv6(std::string c, double t) { return has_format(std::move(c), ft(FormatType::IPv6), t, {}); }
Check::Builder &Check::Builder::validates_json(std::string c, double t) { return has_format(std::move(c), ft(FormatType::Json), t, {}); }
Check::Builder &Check::Builder::validates_iso8601_datetime(std::string c, double t) {
  return has_format(std::move(c), ft(FormatType::Iso8601DateTime), t, {});
}
Check::Builder &Check::Builder::email(std::string column, double threshold) {
  FormatOptions o;
  o.trim_before_check = true;
  o.null_is_valid = false;  // builder_extensions.rs:309-318
  return has_format(std::move(column), ft(FormatType::Email), threshold, o);
}
Check::Builder &Check::Builder::contains_ssn(std::string column, double threshold) {
  return has_format(std::move(column), ft(FormatType::SocialSecurityNumber), threshold, trimming());
}
Check::Builder &Check::Builder::has_approx_quantile(std::string column, double quantile, Assertion a) {
  QuantileValidation v;
  v.kind = QuantileValidation::Single;
  v.checks.push_back({quantile, a});
  return quantile_validation(std::move(column), std::move(v));
}
Check::Builder &Check::Builder::has_correlation(std::string c1, std::string c2, Assertion a) {
  CorrelationValidation v;
  v.kind = CorrelationValidation::Pairwise;
  v.column1 = std::move(c1);
  v.column2 = std::move(c2);
  v.assertion = a;
  return correlation(std::move(v));
}
Check::Builder &Check::Builder::multi_statistic(std::string column, std::vector<std::pair<StatisticType, Assertion>> st) {
  return constraint(std::make_shared<MultiStatisticalConstraint>(std::move(column), std::move(st)));
}
Check::Builder &Check::Builder::quantile_validation(std::string column, QuantileValidation v) {
  return constraint(std::make_shared<QuantileConstraint>(std::move(column), std::move(v)));
}
Check::Builder &Check::Builder::correlation(CorrelationValidation v) {
  return constraint(std::make_shared<CorrelationConstraint>(std::move(v)));
}

std::string CorrelationType::name() const {
  switch (kind) {
    case Pearson: return "Pearson correlation";
    case Spearman: return "Spearman correlation";
    case KendallTau: return "Kendall's tau";
    case MutualInformation: return "mutual information";
    case Covariance: return "covariance";
    default: return "custom correlation";
  }
}
std::string CorrelationType::constraint_name() const {
  switch (kind) {
    case Pearson: return "correlation";
    case Spearman: return "spearman_correlation";
    case KendallTau: return "kendall_correlation";
    case MutualInformation: return "mutual_information";
    case Covariance: return "covariance";
    default: return "custom_correlation";
  }
}

bool ValidationReport::has_errors() const {
  for (auto &i : issues)
    if (i.level == Level::Error) return true;
  return false;
}
bool ValidationReport::has_warnings() const {
  for (auto &i : issues)
    if (i.level == Level::Warning) return true;
  return false;
}

// ------------------------------------------------------------------------------------------------ suite runner
namespace {
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
struct QuantileCtx {
  const tgx_plan *plan;
  tgx_state *state;
  std::vector<size_t> spec_of_request;
};
double quantile_cb(const void *ctx, size_t request_index, double phi) {
  const QuantileCtx *q = (const QuantileCtx *)ctx;
  double out = NAN;
  tgx_error err;
  if (tgx_kll_quantile(q->plan, q->state, q->spec_of_request[request_index], phi, &out, &err) != TGX_OK)
    throw TermError{TermError::Internal, err.msg};
  return out;
}
std::string now_rfc3339() {
  using namespace std::chrono;
  auto now = system_clock::now();
  time_t t = system_clock::to_time_t(now);
  struct tm tmv;
  gmtime_r(&t, &tmv);
  char buf[64];
  strftime(buf, sizeof(buf), "%Y-%m-%dT%H:%M:%S", &tmv);
  auto us = duration_cast<microseconds>(now.time_since_epoch()).count() % 1000000;
  char out[96];
  snprintf(out, sizeof(out), "%s.%06lld+00:00", buf, (long long)us);
  return out;
}
}  // namespace

ValidationResult ValidationSuite::run(const Context &ctx) const {
  const auto t0 = std::chrono::steady_clock::now();
  ValidationResult result;
  ValidationReport &report = result.report;
  report.suite_name = name_;
  report.timestamp = now_rfc3339();
  ValidationMetrics metrics;
  bool has_errors = false;

  const Table *table = ctx.table(table_name_);

  // ---- plan: every constraint's aggregate requests, fused and de-duplicated
  struct Planned {
    const Check *check;
    const Constraint *constraint;
    std::vector<SpecRequest> requests;
    std::vector<size_t> spec_index;
    std::optional<std::string> error;  // what the reference would return as Err(e) from evaluate()
  };
  std::vector<Planned> planned;
  std::vector<tgx_check_spec> specs;
  std::vector<SpecRequest> spec_requests;  // owns the pattern strings the specs point to
  std::deque<std::vector<int32_t>> tuple_columns;  // owns the column lists of tuple specs
  auto column_index = [&](const std::string &name) -> int {
    if (!table) return -1;
    for (size_t i = 0; i < table->column_names.size(); i++)
      if (table->column_names[i] == name) return (int)i;
    return -1;
  };
  // the Arrow DataType of a column: what the caller declared, else what its tgx_type says
  auto arrow_type_of = [&](const std::string &name) -> std::string {
    auto declared = declared_types_.find(name);
    if (declared != declared_types_.end()) return declared->second;
    const int ci = column_index(name);
    if (ci < 0) return std::string();
    if ((size_t)ci < table->arrow_types.size() && !table->arrow_types[ci].empty()) return table->arrow_types[ci];
    for (const Batch &b : table->batches)
      if ((size_t)ci < b.columns.size()) switch (b.columns[ci].type) {
          case TGX_INT64: return "Int64";
          case TGX_FLOAT64: return "Float64";
          case TGX_INT32: return "Int32";
          case TGX_FLOAT32: return "Float32";
          default: return std::string();
        }
    return std::string();
  };
  for (const Check &check : checks_) {
    for (const auto &c : check.constraints()) {
      Planned p;
      p.check = &check;
      p.constraint = c.get();
      if (!table) {
        p.error = TermError{TermError::DataFusion,
                            "Error during planning: table 'datafusion.public." + table_name_ + "' not found"}
                      .display();
      } else {
        try {
          p.requests = c->plan();
        } catch (const TermError &e) {
          p.error = e.display();
        }
      }
      if (!p.error) {
        for (SpecRequest &r : p.requests) {
          if (r.column.empty() && r.kind == TGX_CHECK_COUNT && !table->column_names.empty())
            r.column = table->column_names[0];
          std::vector<const std::string *> named = {&r.column};
          if (r.kind == TGX_CHECK_COMOMENTS) named.push_back(&r.column2);
          for (const std::string &c2 : r.columns) named.push_back(&c2);
          for (const std::string *col : named) {
            if (column_index(*col) < 0) {
              p.error = TermError{TermError::DataFusion, "Schema error: No field named " + *col + "."}.display();
              break;
            }
          }
          if (p.error) break;
        }
      }
      planned.push_back(std::move(p));
    }
  }
  // the specs of the constraints that are still in the run, fused and de-duplicated
  auto fuse = [&]() {
    specs.clear();
    spec_requests.clear();
    tuple_columns.clear();
    spec_requests.reserve(256);
    for (Planned &p : planned) {
      p.spec_index.clear();
      if (p.error) continue;
      for (const SpecRequest &r : p.requests) {
        size_t found = spec_requests.size();
        for (size_t i = 0; i < spec_requests.size(); i++) {
          const SpecRequest &q = spec_requests[i];
          if (q.kind == r.kind && q.column == r.column && q.column2 == r.column2 && q.columns == r.columns &&
              q.flags == r.flags && q.pattern == r.pattern && q.kll_k == r.kll_k && q.length_min == r.length_min &&
              q.length_max == r.length_max)
            found = i;
        }
        if (found == spec_requests.size()) spec_requests.push_back(r);
        p.spec_index.push_back(found);
      }
    }
    for (const SpecRequest &r : spec_requests) {
      tgx_check_spec s;
      memset(&s, 0, sizeof(s));
      s.kind = r.kind;
      s.column = column_index(r.column);
      s.column2 = r.kind == TGX_CHECK_COMOMENTS ? column_index(r.column2) : -1;
      s.flags = r.flags;
      // COUNT(DISTINCT) by VALUE, as DataFusion groups (hash + equality): string / tuple keys are kept with their bytes
      if (r.kind == TGX_CHECK_DISTINCT && exact_keys_) s.flags |= TGX_FLAG_EXACT_KEYS;
      s.pattern = r.pattern.empty() ? nullptr : r.pattern.data();
      s.pattern_len = r.pattern.size();
      s.kll_k = r.kll_k;
      s.length_min = r.length_min;
      s.length_max = r.length_max;
      if (r.columns.size() >= 2) {
        tuple_columns.emplace_back();
        for (const std::string &c2 : r.columns) tuple_columns.back().push_back(column_index(c2));
        s.columns = tuple_columns.back().data();
        s.n_columns = (uint32_t)tuple_columns.back().size();
      }
      specs.push_back(s);
    }
  };

  // ---- one pass over the table
  Handles h;
  std::vector<tgx_result> results;
  std::optional<std::string> run_error;
  auto pass = [&](Handles &hh, size_t max_rows, bool finalize, tgx_status *status) -> std::optional<std::string> {
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
      } else {  // (a probe: the first rows of the first batch say whether the library takes these checks on these types)
        int64_t rows = 0;
        for (const tgx_column &c : cols) rows = std::max(rows, c.length);
        if (rows == 0 && b + 1 < table->batches.size()) continue;  // (an empty record batch says nothing)
        cut = cols;
        for (tgx_column &c : cut) c.length = std::min<int64_t>(c.length, (int64_t)max_rows);
        s = tgx_update(hh.plan, hh.state, cut.data(), cut.size(), &err);
        break;
      }
    }
    if (s == TGX_OK && finalize) {
      results.assign(specs.size(), tgx_result());
      s = tgx_finalize(hh.plan, hh.state, results.data(), results.size(), &err);
    }
    *status = s;
    if (s != TGX_OK) return TermError{TermError::Internal, std::string(tgx_status_name(s)) + ": " + err.msg}.display();
    return std::nullopt;
  };
  fuse();
  if (!specs.empty()) {
    tgx_status status = TGX_OK;
    run_error = pass(h, SIZE_MAX, true, &status);
    if (run_error && (status == TGX_UNSUPPORTED || status == TGX_INVALID_ARGUMENT)) {
      // One binding the library does not take (a check on a column type outside the path, a pattern it cannot compile)
      // must not cost the others their verdicts -- in the reference every constraint is a query of its own.  Every
      // constraint's specs are tried ALONE on the first row of the table; the ones that are refused keep the refusal
      // as their error ("Error evaluating constraint: ..", as an Err from evaluate() reads), the rest run again as one
      // pass.  (Only after a refusal: a suite the library takes whole pays nothing for this.)
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
        // (finalized: a batch this small is only noted by tgx_update -- what it is refused for shows at the flush)
        const std::optional<std::string> pe = specs.empty() ? std::nullopt : pass(probe, 1, true, &ps);
        for (size_t j = 0; j < planned.size(); j++) planned[j].error = saved[j];
        if (pe && (ps == TGX_UNSUPPORTED || ps == TGX_INVALID_ARGUMENT)) {
          planned[k].error = saved[k] = pe;
          refused++;
        }
      }
      if (refused) {
        h.reset();
        fuse();
        run_error.reset();
        if (!specs.empty()) run_error = pass(h, SIZE_MAX, true, &status);
      } else {
        fuse();
      }
    }
  }

  // ---- verdicts + tally (core/suite.rs:84-257)
  for (Planned &p : planned) {
    metrics.total_checks += 1;
    ConstraintResult cr;
    std::optional<std::string> error = p.error ? p.error : run_error;
    if (!error) {
      Constraint::Inputs in;
      QuantileCtx qctx{h.plan, h.state, p.spec_index};
      for (size_t si : p.spec_index) in.results.push_back(&results[si]);
      for (const SpecRequest &r : p.requests) in.arrow_types.push_back(arrow_type_of(r.column));
      in.strict_reference_types = strict_types_;
      in.ctx = &qctx;
      in.quantile = quantile_cb;
      try {
        cr = p.constraint->evaluate(in);
      } catch (const TermError &e) {
        error = e.display();
      }
    }
    if (error) {
      metrics.failed_checks += 1;
      report.issues.push_back({p.check->name(), p.constraint->name(), p.check->level(),
                               "Error evaluating constraint: " + *error, {}});
      if (p.check->level() == Level::Error) has_errors = true;
      continue;
    }
    switch (cr.status) {
      case ConstraintStatus::Success:
        metrics.passed_checks += 1;
        break;
      case ConstraintStatus::Failure: {
        metrics.failed_checks += 1;
        std::string msg = cr.message ? *cr.message : "Constraint " + p.constraint->name() + " failed";
        report.issues.push_back({p.check->name(), p.constraint->name(), p.check->level(), msg, cr.metric});
        if (p.check->level() == Level::Error) has_errors = true;
        break;
      }
      case ConstraintStatus::Skipped:
        metrics.skipped_checks += 1;
        break;
    }
    if (cr.metric) metrics.custom_metrics[p.check->name() + "." + p.constraint->name()] = *cr.metric;
  }
  metrics.execution_time_ms =
      (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
  report.metrics = metrics;
  result.success = !has_errors;
  return result;
}

// ------------------------------------------------------------------------------------------------ JSON out
static std::string json_num(double v) {
  if (isnan(v) || isinf(v)) return "null";  // serde_json writes non-finite f64 as null
  // serde_json prints integral floats with a trailing ".0"
  std::string s = rust_f64(v);
  if (s.find('.') == std::string::npos && s.find('e') == std::string::npos) s += ".0";
  return s;
}

static void metrics_json(const ValidationMetrics &m, const std::string &ind, std::string &o) {
  o += "{\n";
  o += ind + "  \"total_checks\": " + std::to_string(m.total_checks) + ",\n";
  o += ind + "  \"passed_checks\": " + std::to_string(m.passed_checks) + ",\n";
  o += ind + "  \"failed_checks\": " + std::to_string(m.failed_checks) + ",\n";
  o += ind + "  \"skipped_checks\": " + std::to_string(m.skipped_checks) + ",\n";
  o += ind + "  \"execution_time_ms\": " + std::to_string(m.execution_time_ms);
  if (!m.custom_metrics.empty()) {  // skip_serializing_if = "HashMap::is_empty"
    o += ",\n" + ind + "  \"custom_metrics\": {\n";
    size_t k = 0;
    for (auto &kv : m.custom_metrics) {
      o += ind + "    " + json::quote(kv.first) + ": " + json_num(kv.second);
      o += ++k < m.custom_metrics.size() ? ",\n" : "\n";
    }
    o += ind + "  }";
  }
  o += "\n" + ind + "}";
}

std::string ValidationResult::to_json() const {
  std::string o = "{\n";
  o += std::string("  \"status\": \"") + (success ? "success" : "failure") + "\",\n";
  if (success) {
    o += "  \"metrics\": ";
    metrics_json(report.metrics, "  ", o);
    o += ",\n";
  }
  o += "  \"report\": {\n";
  o += "    \"suite_name\": " + json::quote(report.suite_name) + ",\n";
  o += "    \"timestamp\": " + json::quote(report.timestamp) + ",\n";
  o += "    \"metrics\": ";
  metrics_json(report.metrics, "    ", o);
  o += ",\n    \"issues\": [";
  for (size_t i = 0; i < report.issues.size(); i++) {
    const ValidationIssue &is = report.issues[i];
    o += i ? ",\n      {\n" : "\n      {\n";
    o += "        \"check_name\": " + json::quote(is.check_name) + ",\n";
    o += "        \"constraint_name\": " + json::quote(is.constraint_name) + ",\n";
    o += std::string("        \"level\": \"") + level_str(is.level) + "\",\n";
    o += "        \"message\": " + json::quote(is.message);
    if (is.metric) o += ",\n        \"metric\": " + json_num(*is.metric);
    o += "\n      }";
  }
  o += report.issues.empty() ? "]\n" : "\n    ]\n";
  o += "  }\n}";
  return o;
}

// ------------------------------------------------------------------------------------------------ JSON in
static Assertion assertion_from(const json::Value &v) {
  if (!v.is(json::Value::Object)) throw TermError{TermError::Internal, "assertion must be an object"};
  const std::string k = v.get_str("kind");
  const json::Value *args = v.get("args");
  auto arg = [&](size_t i) -> double {
    if (!args || !args->is(json::Value::Array) || args->arr.size() <= i || !args->arr[i].is(json::Value::Number))
      throw TermError{TermError::Internal, "assertion '" + k + "' is missing a numeric argument"};
    return args->arr[i].num;
  };
  if (k == "equals") return Assertion::equals(arg(0));
  if (k == "not_equals") return Assertion::not_equals(arg(0));
  if (k == "greater_than") return Assertion::greater_than(arg(0));
  if (k == "greater_than_or_equal") return Assertion::greater_than_or_equal(arg(0));
  if (k == "less_than") return Assertion::less_than(arg(0));
  if (k == "less_than_or_equal") return Assertion::less_than_or_equal(arg(0));
  if (k == "between") return Assertion::between(arg(0), arg(1));
  if (k == "not_between") return Assertion::not_between(arg(0), arg(1));
  throw TermError{TermError::Internal, "unknown assertion kind '" + k + "'"};
}

static std::vector<std::string> strings_from(const json::Value *v) {
  std::vector<std::string> out;
  if (!v) return out;
  if (v->is(json::Value::String)) {
    out.push_back(v->str);
  } else if (v->is(json::Value::Array)) {
    for (auto &e : v->arr)
      if (e.is(json::Value::String)) out.push_back(e.str);
  }
  return out;
}

static LogicalOperator operator_from(const json::Value *v) {
  LogicalOperator op;
  if (!v) return op;
  if (v->is(json::Value::String)) {
    if (v->str == "any") op.kind = LogicalOperator::Any;
    else if (v->str != "all") throw TermError{TermError::Internal, "unknown operator '" + v->str + "'"};
    return op;
  }
  if (v->is(json::Value::Object) && v->obj.size() == 1 && v->obj[0].second.is(json::Value::Number)) {
    const std::string &k = v->obj[0].first;
    op.n = (size_t)v->obj[0].second.num;
    if (k == "exactly") op.kind = LogicalOperator::Exactly;
    else if (k == "at_least") op.kind = LogicalOperator::AtLeast;
    else if (k == "at_most") op.kind = LogicalOperator::AtMost;
    else throw TermError{TermError::Internal, "unknown operator '" + k + "'"};
    return op;
  }
  throw TermError{TermError::Internal, "malformed operator"};
}

static StatisticType statistic_from(const std::string &s, double p) {
  static const char *const names[] = {"min", "max", "mean", "sum", "standard_deviation", "variance", "median", "percentile"};
  StatisticType st;
  int k = -1;
  for (int i = 0; i < 8; i++)
    if (s == names[i]) k = i;
  if (k < 0) throw TermError{TermError::Internal, "unknown statistic '" + s + "'"};
  st.kind = (StatisticType::Kind)k;
  st.p = p;
  return st;
}

static const json::Value *need(const json::Value &c, const char *key) {
  const json::Value *v = c.get(key);
  if (!v) throw TermError{TermError::Internal, std::string("constraint JSON is missing '") + key + "'"};
  return v;
}

void add_constraint_from_json(Check::Builder &b, const json::Value &c) {
  const std::string type = c.get_str("type");
  if (type == "size") {
    b.has_size(assertion_from(*c.get("assertion")));
  } else if (type == "approx_count_distinct") {
    b.has_approx_count_distinct(c.get_str("column"), assertion_from(*c.get("assertion")));
  } else if (type == "completeness") {
    CompletenessOptions o;
    o.op = operator_from(c.get("operator"));
    o.threshold = c.get_num("threshold", 1.0);
    b.completeness(strings_from(c.get("columns")), o);
  } else if (type == "statistic") {
    b.statistic(c.get_str("column"), statistic_from(c.get_str("statistic"), c.get_num("p", 0.5)),
                assertion_from(*need(c, "assertion")));
  } else if (type == "uniqueness") {
    UniquenessType t;
    const std::string k = c.get_str("kind", "full_uniqueness");
    if (k == "full_uniqueness") t.kind = UniquenessType::FullUniqueness;
    else if (k == "distinctness") t.kind = UniquenessType::Distinctness;
    else if (k == "unique_value_ratio") t.kind = UniquenessType::UniqueValueRatio;
    else if (k == "primary_key") t.kind = UniquenessType::PrimaryKey;
    else if (k == "unique_with_nulls") t.kind = UniquenessType::UniqueWithNulls;
    else throw TermError{TermError::Internal, "unknown uniqueness kind '" + k + "'"};
    t.threshold = c.get_num("threshold", 1.0);
    if (c.get("assertion")) t.assertion = assertion_from(*c.get("assertion"));
    if ((t.kind == UniquenessType::Distinctness || t.kind == UniquenessType::UniqueValueRatio) && !t.assertion)
      throw TermError{TermError::Internal, "uniqueness kind '" + k + "' needs an assertion"};
    const std::string nh = c.get_str("null_handling", "exclude");
    t.null_handling = nh == "include" ? NullHandling::Include : nh == "distinct" ? NullHandling::Distinct : NullHandling::Exclude;
    b.uniqueness(strings_from(c.get("columns")), t);
  } else if (type == "containment") {
    b.is_contained_in(c.get_str("column"), strings_from(c.get("allowed_values")));
  } else if (type == "length") {
    b.length(c.get_str("column"), c.get_str("kind"), (uint64_t)c.get_num("a", 0), (uint64_t)c.get_num("b", 0));
  } else if (type == "format") {
    static const char *const names[] = {"regex", "email", "url", "credit_card", "phone", "postal_code", "uuid", "ipv4",
                                        "ipv6", "json", "iso8601_datetime", "social_security_number"};
    FormatType f;
    const std::string s = c.get_str("format");
    int k = -1;
    for (int i = 0; i < 12; i++)
      if (s == names[i]) k = i;
    if (k < 0) throw TermError{TermError::Internal, "unknown format '" + s + "'"};
    f.kind = (FormatType::Kind)k;
    f.pattern = c.get_str("pattern");
    f.allow_localhost = c.get_bool("allow_localhost");
    f.detect_only = c.get_bool("detect_only");
    if (c.get("country") && c.get("country")->is(json::Value::String)) f.country = c.get_str("country");
    FormatOptions o;
    if (const json::Value *ov = c.get("options")) {
      o.case_sensitive = ov->get_bool("case_sensitive", true);
      o.trim_before_check = ov->get_bool("trim_before_check", false);
      o.null_is_valid = ov->get_bool("null_is_valid", true);
    }
    b.has_format(c.get_str("column"), f, c.get_num("threshold", 1.0), o);
  } else if (type == "multi_statistic") {
    std::vector<std::pair<StatisticType, Assertion>> stats;
    if (const json::Value *sv = c.get("statistics"))
      for (const json::Value &e : sv->arr)
        stats.emplace_back(statistic_from(e.get_str("statistic"), e.get_num("p", 0.5)), assertion_from(*need(e, "assertion")));
    b.multi_statistic(c.get_str("column"), std::move(stats));
  } else if (type == "quantile") {
    QuantileValidation v;
    const std::string k = c.get_str("validation", "single");
    if (k == "single") {
      v.kind = QuantileValidation::Single;
      v.checks.push_back({c.get_num("quantile", 0.5), assertion_from(*need(c, "assertion"))});
    } else if (k == "multiple") {
      v.kind = QuantileValidation::Multiple;
      if (const json::Value *cv = c.get("checks"))
        for (const json::Value &e : cv->arr) v.checks.push_back({e.get_num("quantile", 0.5), assertion_from(*need(e, "assertion"))});
    } else if (k == "monotonic") {
      v.kind = QuantileValidation::Monotonic;
      v.strict = c.get_bool("strict");
      if (const json::Value *qv = c.get("quantiles"))
        for (const json::Value &e : qv->arr)
          if (e.is(json::Value::Number)) v.quantiles.push_back(e.num);
    } else if (k == "distribution") {
      v.kind = QuantileValidation::Distribution;
    } else if (k == "custom") {
      v.kind = QuantileValidation::Custom;
    } else {
      throw TermError{TermError::Internal, "unknown quantile validation '" + k + "'"};
    }
    b.quantile_validation(c.get_str("column"), std::move(v));
  } else if (type == "correlation") {
    CorrelationValidation v;
    const std::string k = c.get_str("validation", "pairwise");
    if (k == "pairwise") v.kind = CorrelationValidation::Pairwise;
    else if (k == "range") v.kind = CorrelationValidation::Range;
    else if (k == "independence") v.kind = CorrelationValidation::Independence;
    else if (k == "multi_column") v.kind = CorrelationValidation::MultiColumn;
    else if (k == "stability") v.kind = CorrelationValidation::Stability;
    else throw TermError{TermError::Internal, "unknown correlation validation '" + k + "'"};
    static const char *const types[] = {"pearson", "spearman", "kendall_tau", "mutual_information", "covariance", "custom"};
    const std::string t = c.get_str("correlation_type", "pearson");
    int ti = -1;
    for (int i = 0; i < 6; i++)
      if (t == types[i]) ti = i;
    if (ti < 0) throw TermError{TermError::Internal, "unknown correlation type '" + t + "'"};
    v.type.kind = (CorrelationType::Kind)ti;
    v.type.sql_expression = c.get_str("sql_expression");
    v.column1 = c.get_str("column1");
    v.column2 = c.get_str("column2");
    v.columns = strings_from(c.get("columns"));
    if (v.kind == CorrelationValidation::Pairwise) v.assertion = assertion_from(*need(c, "assertion"));
    v.min = c.get_num("min", 0.0);
    v.max = c.get_num("max", 0.0);
    v.max_correlation = c.get_num("max_correlation", 0.0);
    b.correlation(std::move(v));
  } else {
    throw TermError{TermError::Internal, "unknown constraint type '" + type + "'"};
  }
}

ValidationSuite suite_from_json(const std::string &text) {
  json::Value root;
  std::string err;
  if (!json::parse(text, &root, &err) || !root.is(json::Value::Object))
    throw TermError{TermError::Internal, "suite JSON: " + (err.empty() ? std::string("not an object") : err)};
  ValidationSuite::Builder sb = ValidationSuite::builder(root.get_str("name", "suite"));
  if (root.get("table_name")) sb.table_name(root.get_str("table_name", "data"));
  if (root.get("strict_reference_types")) sb.strict_reference_types(root.get_bool("strict_reference_types"));
  if (root.get("exact_string_keys")) sb.exact_string_keys(root.get_bool("exact_string_keys"));
  if (const json::Value *ct = root.get("column_types"))
    if (ct->is(json::Value::Object))
      for (const auto &kv : ct->obj)
        if (kv.second.is(json::Value::String)) sb.column_type(kv.first, kv.second.str);
  if (root.get("description")) sb.description(root.get_str("description"));
  if (const json::Value *checks = root.get("checks")) {
    for (const json::Value &cv : checks->arr) {
      Check::Builder cb = Check::builder(cv.get_str("name", "check"));
      const std::string lvl = cv.get_str("level", "warning");
      cb.level(lvl == "error" ? Level::Error : lvl == "info" ? Level::Info : Level::Warning);
      if (cv.get("description")) cb.description(cv.get_str("description"));
      if (const json::Value *cs = cv.get("constraints"))
        for (const json::Value &c : cs->arr) add_constraint_from_json(cb, c);
      sb.check(cb.build());
    }
  }
  return sb.build();
}

}  // namespace term_guard
