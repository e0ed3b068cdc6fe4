/*
 * tgx_host.h -- C entry points of the host-side mirror of term-guard's ValidationSuite / Check /
 * Constraint surface (term_amd/csrc/host/term_guard.h is the C++ API; this is the bridge other languages
 * bind).  Suites travel as JSON; results come back as the JSON term-guard's own JsonFormatter produces
 * (`serde_json::to_string_pretty(ValidationResult)`, TG/formatters.rs:222-245, TG/core/result.rs:123-136).
 *
 * Suite JSON:
 *   {"name": "...", "table_name": "data",
 *    "column_types": {"c": "Int32", "d": "Timestamp(Nanosecond, None)"},   (optional: Arrow DataTypes as the caller holds
 *                     them; a column without an entry is what its tgx_type says)
 *    "strict_reference_types": true,    (default.  StatisticalConstraint reads its aggregate as Int64Array or Float64Array
 *                     and fails with "Failed to extract statistic value" otherwise (TG/constraints/statistics.rs:277-308):
 *                     MIN / MAX / quantiles of an Int32, Date32, Float32, Timestamp or UInt column, SUM of a UInt column
 *                     are ERRORS there, and here.  false: the value of the widened column answers -- a deviation)
 *    "checks": [{"name": "...", "level": "error|warning|info",
 *     "constraints": [
 *       {"type": "size", "assertion": A},
 *       {"type": "approx_count_distinct", "column": "c", "assertion": A},   (metric: the exact distinct count)
 *       {"type": "completeness", "columns": ["c", ...], "operator": "all"|"any"|{"at_least": n}|{"exactly": n}|
 *                                 {"at_most": n}, "threshold": 1.0},
 *       {"type": "statistic", "column": "c", "statistic": "min|max|mean|sum|standard_deviation|variance|median|
 *                              percentile", "p": 0.5, "assertion": A},
 *       {"type": "uniqueness", "columns": ["c"], "kind": "full_uniqueness|distinctness|unique_value_ratio|primary_key|
 *                               unique_with_nulls", "threshold": 1.0, "assertion": A, "null_handling": "exclude|include|distinct"},
 *       {"type": "format", "column": "c", "format": "regex|email|url|credit_card|phone|postal_code|uuid|ipv4|ipv6|json|
 *                           iso8601_datetime|social_security_number", "pattern": "...", "allow_localhost": false,
 *                           "detect_only": false, "country": "US", "threshold": 1.0,
 *                           "options": {"case_sensitive": true, "trim_before_check": false, "null_is_valid": true}},
 *       {"type": "containment", "column": "c", "allowed_values": ["a", "b"]},
 *       {"type": "length", "column": "c", "kind": "min|max|between|exactly|not_empty", "a": n, "b": m},
 *       {"type": "quantile", "column": "c", "quantile": 0.5, "assertion": A},
 *       {"type": "correlation", "column1": "a", "column2": "b", "assertion": A}]}]}
 *   A = {"kind": "equals|not_equals|greater_than|greater_than_or_equal|less_than|less_than_or_equal|between|
 *                 not_between", "args": [x] or [lo, hi]}                          (TG/constraints/assertion.rs:27-46)
 *
 * Strings returned through `char **` are owned by the library: release them with tgx_host_free.
 */
#ifndef TGX_HOST_H
#define TGX_HOST_H

#include "tgx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ValidationSuite::run (TG/core/suite.rs:399): plans every constraint into one tgx_plan, feeds the table's
 * batches once, applies each constraint's verdict.  `columns` holds n_batches x n_columns views, batch-major;
 * column_names[i] names columns[b * n_columns + i].  The table is registered under the suite's table_name. */
tgx_status tgx_host_run_suite_json(const char *suite_json, const char *const *column_names, size_t n_columns,
                                   const tgx_column *columns, size_t n_batches, char **out_json, tgx_error *err);

/* The aggregates one constraint asks for (its half of the fused plan), as a JSON array. */
tgx_status tgx_host_constraint_plan_json(const char *constraint_json, char **out_json, tgx_error *err);

/* `Constraint::evaluate`'s verdict half on given aggregates: results_json is an array of objects with
 * tgx_result's field names (answering the plan in order; a KLL entry may carry {"quantiles": {"0.5": v}}).
 * Returns {"status": "success|failure|skipped", "metric": x|null, "message": s|null}.  Needs no device.
 * The constraint JSON may carry "column_type" (the column's Arrow DataType) and "strict_reference_types" as above. */
tgx_status tgx_host_constraint_verdict_json(const char *constraint_json, const char *results_json, char **out_json,
                                            tgx_error *err);

/* SqlSecurity::validate_identifier (TG/security.rs:103-146) */
tgx_status tgx_host_validate_identifier(const char *identifier, tgx_error *err);

/* Assertion::evaluate + Display (TG/constraints/assertion.rs:48-76) */
tgx_status tgx_host_assertion_json(const char *assertion_json, double value, int32_t *holds, char **description,
                                   tgx_error *err);

/* ---- analyzers (TG/analyzers/traits.rs:65-179, runner.rs:64-202) --------------------------------------------
 * Analysis JSON: {"table_name": "data", "continue_on_error": true, "analyzers": [
 *     {"type": "size"}, {"type": "completeness|distinctness|approx_count_distinct|mean|min|max|sum|standard_deviation", "column": "c"},
 *     {"type": "correlation", "column1": "a", "column2": "b", "method": "pearson|spearman|covariance"}]}
 * AnalysisRunner::run: every analyzer's aggregates planned into ONE tgx_plan, one pass over the table.  Returns
 *   {"metrics": {metric_key: {"type": "Double|Long|Map", "value": ..}},      (MetricValue's serde form, types.rs:11-35)
 *    "states":  {metric_key: {..the AnalyzerState's serde fields..}},
 *    "errors":  [{"analyzer_name": "..", "error": ".."}]}                        (context.rs:118-123)
 * With continue_on_error false the first failing analyzer makes the call fail with "Analyzer {name} failed". */
tgx_status tgx_host_run_analysis_json(const char *analysis_json, const char *const *column_names, size_t n_columns,
                                      const tgx_column *columns, size_t n_batches, char **out_json, tgx_error *err);

/* AnalyzerState::merge of the analyzer's state type over a JSON array of states; needs no device. */
tgx_status tgx_host_merge_states_json(const char *analyzer_json, const char *states_json, char **out_state_json,
                                      tgx_error *err);

/* Analyzer::compute_metric_from_state; needs no device.  An AnalyzerError (e.g. NoData) is returned as
 * TGX_INVALID_ARGUMENT with the reference's Display text in err->msg. */
tgx_status tgx_host_metric_from_state_json(const char *analyzer_json, const char *state_json, char **out_metric_json,
                                           tgx_error *err);

void tgx_host_free(char *s);

#ifdef __cplusplus
}
#endif
#endif /* TGX_HOST_H */
