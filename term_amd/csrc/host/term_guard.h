// term_guard.h -- host-side mirror of term-guard's ValidationSuite / Check / Constraint surface, driving
// libtgx instead of DataFusion.
//
// Same names, argument meaning, verdict rules and messages as the reference (cited per item, paths
// relative to /root/reference/term-guard/src):
//   ValidationSuite::builder(..).table_name(..).check(..).build().run(..)   core/suite.rs:351-600
//   Check::builder(..).level(..).completeness(..).has_min(..) ...           core/check.rs:172-2310
//   Constraint / ConstraintResult / ConstraintStatus                        core/constraint.rs:13-225
//   Assertion                                                               constraints/assertion.rs:27-76
//   LogicalOperator / ConstraintOptions                                     core/logical.rs:32-100, core/unified.rs
//   ValidationResult / Report / Metrics / Issue, Level                      core/result.rs, core/level.rs
// What differs by design: where the reference runs one SQL scan per constraint (core/suite.rs:67-100),
// run() plans every constraint's aggregates into ONE tgx_plan, feeds each batch once and then lets each
// constraint apply its own verdict to the shared results.  The DataFusion SessionContext is replaced by
// `Context`, a registry of named tables whose batches are Arrow-layout column views (host or device).
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "../../../include/tgx.h"

namespace term_guard {

// ---- core/level.rs:76-117
enum class Level { Info, Warning, Error };
const char *level_str(Level l);

// ---- core/constraint.rs:13-99
enum class ConstraintStatus { Success, Failure, Skipped };
struct ConstraintResult {
  ConstraintStatus status = ConstraintStatus::Success;
  std::optional<double> metric;
  std::optional<std::string> message;
  static ConstraintResult success() { return {ConstraintStatus::Success, {}, {}}; }
  static ConstraintResult success_with_metric(double m) { return {ConstraintStatus::Success, m, {}}; }
  static ConstraintResult failure(std::string msg) { return {ConstraintStatus::Failure, {}, std::move(msg)}; }
  static ConstraintResult failure_with_metric(double m, std::string msg) {
    return {ConstraintStatus::Failure, m, std::move(msg)};
  }
  static ConstraintResult skipped(std::string msg) { return {ConstraintStatus::Skipped, {}, std::move(msg)}; }
};

// TermError (error.rs:14-145): only the variants this path produces
struct TermError {
  enum Kind { Internal, SecurityError, DataFusion, NotSupported, Configuration, TypeMismatch } kind = Internal;
  std::string message;
  std::string display() const;  // thiserror Display strings, e.g. "Security error: ..."
};

// ---- constraints/assertion.rs:27-76
struct Assertion {
  enum Kind { Equals, NotEquals, GreaterThan, GreaterThanOrEqual, LessThan, LessThanOrEqual, Between, NotBetween };
  Kind kind;
  double a = 0, b = 0;
  static Assertion equals(double v) { return {Equals, v, 0}; }
  static Assertion not_equals(double v) { return {NotEquals, v, 0}; }
  static Assertion greater_than(double v) { return {GreaterThan, v, 0}; }
  static Assertion greater_than_or_equal(double v) { return {GreaterThanOrEqual, v, 0}; }
  static Assertion less_than(double v) { return {LessThan, v, 0}; }
  static Assertion less_than_or_equal(double v) { return {LessThanOrEqual, v, 0}; }
  static Assertion between(double lo, double hi) { return {Between, lo, hi}; }
  static Assertion not_between(double lo, double hi) { return {NotBetween, lo, hi}; }
  bool evaluate(double value) const;
  std::string description() const;
};

// Rust's `{}` for f64 (shortest round-trip digits, never scientific notation)
std::string rust_f64(double v);

// ---- core/logical.rs:32-100
struct LogicalOperator {
  enum Kind { All, Any, Exactly, AtLeast, AtMost } kind = All;
  size_t n = 0;
  bool evaluate(const std::vector<bool> &results) const;
  std::string description() const;
};

// ---- security.rs:89-255
std::optional<TermError> validate_identifier(const std::string &identifier);

// ---- constraints/format.rs:160-390
struct FormatType {
  enum Kind { Regex, Email, Url, CreditCard, Phone, PostalCode, UUID, IPv4, IPv6, Json, Iso8601DateTime,
              SocialSecurityNumber } kind = Regex;
  std::string pattern;        // Regex
  bool allow_localhost = false;
  bool detect_only = false;
  std::optional<std::string> country;  // Phone (optional) / PostalCode (required)
  std::string get_pattern() const;     // format.rs:217-307
  std::string name() const;            // :310-325
  std::string description() const;     // :328-360
};
struct FormatOptions {
  bool case_sensitive = true, trim_before_check = false, null_is_valid = true;  // format.rs:376-384
};

// ---- constraints/uniqueness.rs:56-140
enum class NullHandling { Exclude, Include, Distinct };
struct UniquenessType {
  enum Kind { FullUniqueness, Distinctness, UniqueValueRatio, PrimaryKey, UniqueWithNulls } kind = FullUniqueness;
  double threshold = 1.0;
  std::optional<Assertion> assertion;
  NullHandling null_handling = NullHandling::Exclude;
  std::string name() const;
};

// ---- constraints/statistics.rs:24-108
struct StatisticType {
  enum Kind { Min, Max, Mean, Sum, StandardDeviation, Variance, Median, Percentile } kind = Min;
  double p = 0.5;
  std::string name() const;             // "minimum", ...
  std::string constraint_name() const;  // "min", ...
};

// ---- constraints/quantile.rs:36-112
struct QuantileCheck {
  double quantile = 0.5;
  Assertion assertion = Assertion::equals(0);
};
struct QuantileValidation {
  enum Kind { Single, Multiple, Distribution, Monotonic, Custom } kind = Single;
  std::vector<QuantileCheck> checks;  // Single: one entry; Multiple: one per quantile
  std::vector<double> quantiles;      // Monotonic
  bool strict = false;                // Monotonic
};

// ---- constraints/correlation.rs:19-117
struct CorrelationType {
  enum Kind { Pearson, Spearman, KendallTau, MutualInformation, Covariance, Custom } kind = Pearson;
  std::string sql_expression;        // Custom
  std::string name() const;             // "Pearson correlation", ...  (:39-48)
  std::string constraint_name() const;  // "correlation", ...          (:51-60)
};
struct CorrelationValidation {
  enum Kind { Pairwise, Range, MultiColumn, Independence, Stability } kind = Pairwise;
  std::string column1, column2;
  CorrelationType type;                       // Pairwise / Range
  Assertion assertion = Assertion::equals(0);  // Pairwise
  double min = 0, max = 0;                    // Range
  double max_correlation = 0;                 // Independence
  std::vector<std::string> columns;           // MultiColumn (needs >= 2)
};

// One aggregate a constraint needs; the suite runner resolves column names and fuses all requests.
struct SpecRequest {
  int kind = 0;  // tgx_check_kind
  std::string column, column2;
  std::vector<std::string> columns;  // DISTINCT over a tuple of columns (then `column` is columns[0])
  uint32_t flags = 0;
  std::string pattern;
  uint32_t kll_k = 0;
  uint64_t length_min = 0, length_max = ~0ull;  // LENGTH: inclusive character-count bounds
};

// ---- core/constraint.rs:187-225.  `evaluate(&SessionContext)` is split in two so scans can be fused:
// plan() says which aggregates are needed, evaluate() turns them into the verdict.
class Constraint {
 public:
  virtual ~Constraint() {}
  virtual std::string name() const = 0;
  virtual std::optional<std::string> column() const { return {}; }
  // throws TermError for what the reference reports as Err(...) from evaluate()
  virtual std::vector<SpecRequest> plan() const = 0;
  // results[i] answers plan()[i]; `quantile` evaluates a KLL request (index into plan()) at phi
  struct Inputs {
    std::vector<const tgx_result *> results;
    const void *ctx = nullptr;
    double (*quantile)(const void *ctx, size_t request_index, double phi) = nullptr;
    // the Arrow DataType of the column request i reads, as the caller holds it ("Int64", "Int32", "Date32",
    // "Timestamp", "Float32", "UInt16", ...; empty = unknown, taken for Int64 / Float64), and whether the verdict
    // follows the reference's own extraction rule for it (reference_extracts below)
    std::vector<std::string> arrow_types;
    bool strict_reference_types = true;
  };
  virtual ConstraintResult evaluate(const Inputs &in) const = 0;
};

// What DataFusion 50's aggregate hands back for a column of Arrow type `arrow_type`, and whether the reference can
// read it: StatisticalConstraint::evaluate downcasts the result column to Int64Array, then Float64Array, else
// Err("Failed to extract statistic value") (constraints/statistics.rs:277-308; MultiStatisticalConstraint pushes
// "Failed to compute {name}", :466-483).  MIN / MAX / APPROX_PERCENTILE_CONT keep the input type, SUM widens signed
// integers to Int64 and floats to Float64 (unsigned: UInt64), AVG / STDDEV / VARIANCE are Float64.  So on an Int32,
// Date32, Float32, Timestamp or UInt column the reference's has_min is an ERROR, not a verdict -- and with
// `strict_reference_types` (the default) so is this library's, although the kernels could answer (the widening
// behaviour is opt-in and listed as a deviation in INTEGRATION.md).
enum class StatisticResultKind { Min, Max, Mean, Sum, StandardDeviation, Variance, Quantile };
bool reference_extracts(StatisticResultKind stat, const std::string &arrow_type);
// QuantileConstraint reads Float64, Int64 or Int32 (constraints/quantile.rs:308-324), else TypeMismatch
bool reference_extracts_quantile(const std::string &arrow_type);

// ---- core/check.rs
class Check {
 public:
  const std::string &name() const { return name_; }
  Level level() const { return level_; }
  const std::optional<std::string> &description() const { return description_; }
  const std::vector<std::shared_ptr<Constraint>> &constraints() const { return constraints_; }
  class Builder;
  static Builder builder(std::string name);

 private:
  friend class Builder;
  std::string name_;
  Level level_ = Level::Warning;  // Level::default() (core/level.rs:80-81)
  std::optional<std::string> description_;
  std::vector<std::shared_ptr<Constraint>> constraints_;
};

struct CompletenessOptions {  // core/builder_extensions.rs:14-80 (as ConstraintOptions)
  LogicalOperator op;
  double threshold = 1.0;
  static CompletenessOptions full() { return {}; }
  static CompletenessOptions with_threshold(double t) {
    CompletenessOptions o;
    o.threshold = t;
    return o;
  }
  static CompletenessOptions at_least(size_t n) {
    CompletenessOptions o;
    o.op = {LogicalOperator::AtLeast, n};
    return o;
  }
  static CompletenessOptions any() {
    CompletenessOptions o;
    o.op = {LogicalOperator::Any, 0};
    return o;
  }
};

class Check::Builder {
 public:
  explicit Builder(std::string name) { check_.name_ = std::move(name); }
  Builder &level(Level l) { check_.level_ = l; return *this; }
  Builder &description(std::string d) { check_.description_ = std::move(d); return *this; }
  Builder &constraint(std::shared_ptr<Constraint> c) { check_.constraints_.push_back(std::move(c)); return *this; }
  // check.rs:321 / :1743 / :2233-2300
  Builder &has_size(Assertion a);
  Builder &has_approx_count_distinct(std::string column, Assertion a);  // core/check.rs:379-390 (metric: exact count)
  Builder &completeness(std::vector<std::string> columns, CompletenessOptions options);
  Builder &completeness(std::string column, CompletenessOptions options) {
    return completeness(std::vector<std::string>{std::move(column)}, options);
  }
  Builder &any_complete(std::vector<std::string> columns);
  Builder &at_least_complete(size_t n, std::vector<std::string> columns, double threshold);
  Builder &exactly_complete(size_t n, std::vector<std::string> columns, double threshold);
  // check.rs:1812-1960
  Builder &statistic(std::string column, StatisticType stat, Assertion a);
  Builder &has_min(std::string c, Assertion a) { return statistic(std::move(c), {StatisticType::Min}, a); }
  Builder &has_max(std::string c, Assertion a) { return statistic(std::move(c), {StatisticType::Max}, a); }
  Builder &has_mean(std::string c, Assertion a) { return statistic(std::move(c), {StatisticType::Mean}, a); }
  Builder &has_sum(std::string c, Assertion a) { return statistic(std::move(c), {StatisticType::Sum}, a); }
  Builder &has_standard_deviation(std::string c, Assertion a) {
    return statistic(std::move(c), {StatisticType::StandardDeviation}, a);
  }
  Builder &has_variance(std::string c, Assertion a) { return statistic(std::move(c), {StatisticType::Variance}, a); }
  // check.rs:1480-1740
  Builder &uniqueness(std::vector<std::string> columns, UniquenessType type);
  Builder &validates_uniqueness(std::vector<std::string> columns, double threshold);
  Builder &validates_distinctness(std::vector<std::string> columns, Assertion a);
  Builder &validates_unique_value_ratio(std::vector<std::string> columns, Assertion a);
  Builder &validates_primary_key(std::vector<std::string> columns);
  Builder &validates_uniqueness_with_nulls(std::vector<std::string> columns, double threshold, NullHandling h);
  Builder &primary_key(std::vector<std::string> columns);  // builder_extensions.rs:276-295
  // check.rs:829-1260, builder_extensions.rs:309-420
  // check.rs:518-623, 1777-1785 + constraints/length.rs (kind: min | max | between | exactly | not_empty)
  Builder &is_contained_in(std::string column, std::vector<std::string> allowed_values);  // constraints/values.rs:200-218
  Builder &length(std::string column, std::string kind, uint64_t a, uint64_t b);
  Builder &has_min_length(std::string column, uint64_t n) { return length(std::move(column), "min", n, 0); }
  Builder &has_max_length(std::string column, uint64_t n) { return length(std::move(column), "max", n, 0); }
  Builder &has_length_between(std::string column, uint64_t lo, uint64_t hi) { return length(std::move(column), "between", lo, hi); }
  Builder &has_exact_length(std::string column, uint64_t n) { return length(std::move(column), "exactly", n, 0); }
  Builder &is_not_empty(std::string column) { return length(std::move(column), "not_empty", 0, 0); }
  Builder &has_format(std::string column, FormatType format, double threshold, FormatOptions options);
  Builder &validates_regex(std::string column, std::string pattern, double threshold);
  Builder &validates_email(std::string column, double threshold);
  Builder &validates_url(std::string column, double threshold, bool allow_localhost);
  Builder &validates_credit_card(std::string column, double threshold, bool detect_only);
  Builder &validates_phone(std::string column, double threshold, std::optional<std::string> country);
  Builder &validates_postal_code(std::string column, double threshold, std::string country);
  Builder &validates_uuid(std::string column, double threshold);
  Builder &validates_ipv4(std::string column, double threshold);
  Builder &validates_ipv6(std::string column, double threshold);
  Builder &validates_json(std::string column, double threshold);
  Builder &validates_iso8601_datetime(std::string column, double threshold);
  Builder &email(std::string column, double threshold);
  Builder &contains_ssn(std::string column, double threshold);
  // check.rs:414 / :478
  Builder &has_approx_quantile(std::string column, double quantile, Assertion a);
  Builder &has_correlation(std::string column1, std::string column2, Assertion a);
  // the unified constraints handed to CheckBuilder::constraint(..) in the reference (core/check.rs:263):
  // MultiStatisticalConstraint::new (constraints/statistics.rs:377-417), QuantileConstraint::new / ::multiple
  // (constraints/quantile.rs:159-216), CorrelationConstraint::new / ::independence (constraints/correlation.rs:156-263)
  Builder &multi_statistic(std::string column, std::vector<std::pair<StatisticType, Assertion>> statistics);
  Builder &quantile_validation(std::string column, QuantileValidation validation);
  Builder &correlation(CorrelationValidation validation);
  Check build() { return check_; }

 private:
  Check check_;
};

// ---- the SessionContext stand-in: named tables = named columns x batches
struct Batch {
  std::vector<tgx_column> columns;  // parallel to Table::column_names
};
struct Table {
  std::vector<std::string> column_names;
  std::vector<Batch> batches;
  // Arrow DataType names parallel to column_names (optional: a column without one is what its tgx_type says --
  // TGX_INT64 "Int64", TGX_INT32 "Int32", TGX_FLOAT32 "Float32", ...; a Timestamp / Date32 column handed over as
  // TGX_INT64 / TGX_INT32 needs its name here for the reference's result-type rule to apply)
  std::vector<std::string> arrow_types;
};
class Context {
 public:
  void register_table(const std::string &name, Table t) { tables_[name] = std::move(t); }
  const Table *table(const std::string &name) const {
    auto it = tables_.find(name);
    return it == tables_.end() ? nullptr : &it->second;
  }

 private:
  std::map<std::string, Table> tables_;
};

// ---- core/result.rs
struct ValidationMetrics {
  size_t total_checks = 0, passed_checks = 0, failed_checks = 0, skipped_checks = 0;
  uint64_t execution_time_ms = 0;
  std::map<std::string, double> custom_metrics;
  double success_rate() const { return total_checks == 0 ? 100.0 : 100.0 * passed_checks / total_checks; }
};
struct ValidationIssue {
  std::string check_name, constraint_name;
  Level level;
  std::string message;
  std::optional<double> metric;
};
struct ValidationReport {
  std::string suite_name, timestamp;
  ValidationMetrics metrics;
  std::vector<ValidationIssue> issues;
  bool has_errors() const;
  bool has_warnings() const;
};
struct ValidationResult {
  bool success = true;  // ValidationResult::Success{metrics, report} / Failure{report}
  ValidationReport report;
  bool is_success() const { return success; }
  bool is_failure() const { return !success; }
  std::string to_json() const;  // serde_json::to_string_pretty of the tagged enum (formatters.rs:222-245)
};

// ---- core/suite.rs:351-600
class ValidationSuite {
 public:
  class Builder;
  static Builder builder(std::string name);
  const std::string &name() const { return name_; }
  const std::string &table_name() const { return table_name_; }
  const std::vector<Check> &checks() const { return checks_; }
  // throws TermError only for library-level failures (no device, ...); constraint errors become issues
  ValidationResult run(const Context &ctx) const;

 private:
  friend class Builder;
  std::string name_;
  std::optional<std::string> description_;
  std::string table_name_ = "data";  // suite.rs:549
  std::vector<Check> checks_;
  bool strict_types_ = true;
  bool exact_keys_ = true;
  std::map<std::string, std::string> declared_types_;
};
class ValidationSuite::Builder {
 public:
  explicit Builder(std::string name) { suite_.name_ = std::move(name); }
  Builder &description(std::string d) { suite_.description_ = std::move(d); return *this; }
  Builder &table_name(std::string t) { suite_.table_name_ = std::move(t); return *this; }
  Builder &check(Check c) { suite_.checks_.push_back(std::move(c)); return *this; }
  Builder &with_optimizer(bool) { return *this; }  // suite.rs:457-469: the reference ignores it too
  // true (default): MIN / MAX / SUM / quantiles on columns whose aggregate the reference cannot read are errors, as
  // there (reference_extracts); false: the widened value answers (a deviation)
  Builder &strict_reference_types(bool on) { suite_.strict_types_ = on; return *this; }
  // true (default): uniqueness checks over string / binary / tuple keys count by VALUE -- equal fingerprints are
  // confirmed byte by byte (TGX_FLAG_EXACT_KEYS), `COUNT(DISTINCT c)` as the reference's DataFusion computes it
  // (constraints/uniqueness.rs:612-617); false: by keyed 128-bit fingerprint alone (faster on big batches; two distinct
  // values count once only if all 128 bits agree under a key the data's producer does not know: INTEGRATION.md)
  Builder &exact_string_keys(bool on) { suite_.exact_keys_ = on; return *this; }
  // the Arrow DataType (its Debug form: "Int32", "Date32", "Timestamp(Nanosecond, None)", "UInt8", ...) of a column of
  // the table the suite will run on; takes precedence over Table::arrow_types
  Builder &column_type(std::string column, std::string arrow_type) {
    suite_.declared_types_[std::move(column)] = std::move(arrow_type);
    return *this;
  }
  ValidationSuite build() { return suite_; }

 private:
  ValidationSuite suite_;
};

// JSON description of a suite (the bridge the Python binding uses) -> suite
ValidationSuite suite_from_json(const std::string &json);  // throws TermError on malformed input

}  // namespace term_guard
