// analyzers.h -- host-side mirror of term-guard's Analyzer / AnalyzerState / AnalysisRunner surface
// (TG/analyzers/traits.rs:65-179, TG/analyzers/runner.rs:64-202, TG/analyzers/context.rs:35-128).
//
// The reference runs every analyzer as its own SQL scan (runner.rs:141-165, "TODO: group compatible analyzers");
// here AnalysisRunner::run plans ALL analyzers into one tgx_plan, feeds the table once and lets each analyzer
// turn the shared aggregates into its state and metric.  States keep the reference's serde field names, so a
// state JSON produced here can be merged / turned into a metric exactly as AnalyzerState::merge and
// Analyzer::compute_metric_from_state do -- those two halves need no device.
#pragma once
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "json.h"
#include "term_guard.h"

namespace term_guard {

// TG/analyzers/types.rs:13-35 (serde: {"type": "Double", "value": 0.8})
struct MetricValue {
  enum Kind { Double, Long, Map } kind = Double;
  double d = 0;
  int64_t l = 0;
  std::vector<std::pair<std::string, MetricValue>> map;  // insertion ordered (the reference's HashMap is unordered)
  static MetricValue of_double(double v) {
    MetricValue m;
    m.kind = Double;
    m.d = v;
    return m;
  }
  static MetricValue of_long(int64_t v) {
    MetricValue m;
    m.kind = Long;
    m.l = v;
    return m;
  }
  std::string to_json() const;
};

// TG/analyzers/errors.rs:10-50 -- what(): the reference's Display text
struct AnalyzerError {
  std::string text;
  static AnalyzerError no_data() { return {"No data available for analysis"}; }
  static AnalyzerError invalid_data(const std::string &m) { return {"Invalid data: " + m}; }
  static AnalyzerError state_merge(const std::string &m) { return {"Failed to merge states: " + m}; }
  static AnalyzerError query(const std::string &m) { return {"Query execution failed: " + m}; }
  static AnalyzerError custom(const std::string &m) { return {m}; }
};

class Analyzer {
 public:
  virtual ~Analyzer() {}
  virtual std::string name() const = 0;
  virtual std::string metric_key() const { return name(); }  // traits.rs:133-135
  virtual std::vector<std::string> columns() const { return {}; }
  // aggregates this analyzer needs (its half of the fused plan)
  virtual std::vector<SpecRequest> plan() const = 0;
  // compute_state_from_data's second half: aggregates (answering plan() in order) -> state (serde field names).
  // column_types: tgx_type of columns() in order (the reference's downcasts depend on the SQL result type).
  virtual json::Value state_from_results(const std::vector<const tgx_result *> &r,
                                         const std::vector<int> &column_types) const = 0;
  virtual json::Value merge_states(const std::vector<json::Value> &states) const = 0;      // AnalyzerState::merge
  virtual MetricValue metric_from_state(const json::Value &state) const = 0;              // throws AnalyzerError
};

std::shared_ptr<Analyzer> analyzer_from_json(const json::Value &v);  // throws TermError on malformed input
std::string json_dump(const json::Value &v);

struct AnalyzerContext {  // context.rs:35-44
  std::vector<std::pair<std::string, MetricValue>> metrics;  // key -> value, in execution order
  std::vector<std::pair<std::string, json::Value>> states;   // key -> state (not in the reference's context; for merges)
  std::vector<std::pair<std::string, std::string>> errors;   // (analyzer_name, error text)
  std::string to_json() const;
};

class AnalysisRunner {  // runner.rs:64-202
 public:
  AnalysisRunner &add(std::shared_ptr<Analyzer> a) {
    analyzers_.push_back(std::move(a));
    return *this;
  }
  AnalysisRunner &continue_on_error(bool c) {
    continue_on_error_ = c;
    return *this;
  }
  AnalysisRunner &table_name(std::string t) {
    table_name_ = std::move(t);
    return *this;
  }
  size_t analyzer_count() const { return analyzers_.size(); }
  // throws AnalyzerError ("Analyzer {name} failed") when an analyzer fails and continue_on_error is off
  AnalyzerContext run(const Context &ctx) const;

 private:
  std::vector<std::shared_ptr<Analyzer>> analyzers_;
  bool continue_on_error_ = true;   // runner.rs:71
  std::string table_name_ = "data";  // core/validation_context.rs default
};

}  // namespace term_guard
