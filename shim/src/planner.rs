//! Constraints that answer from ONE fused pass over the table.
//!
//! term-guard evaluates a suite constraint by constraint, each with a SQL query and a full scan of its own
//! (core/suite.rs:67-100).  A [`GpuPlanner`] hands out constraints -- ordinary `impl Constraint`s, added to ordinary
//! `Check`s with `CheckBuilder::constraint` (core/check.rs:263) -- that share one plan: [`GpuPlanner::run`] streams
//! the table once through `tgx_update` (every column read once, whatever the number of checks), and then runs the
//! suite's own `ValidationSuite::run` (core/suite.rs:399), in which every such constraint finds its aggregates
//! waiting and only applies ITS verdict: ratio, threshold, `Assertion::evaluate`, message -- the code of the stock
//! constraint's `evaluate`, restated below with the line it follows.  A constraint whose column type, pattern or shape
//! the library does not take (`TGX_UNSUPPORTED`), and every constraint when the planner was not run, evaluates through
//! the stock constraint it wraps: results never depend on which way a constraint went.
//!
//! | builder call of `CheckBuilder`                      | spec(s)                                   | follows |
//! |---|---|---|
//! | `has_size`                                          | COUNT(first column)                       | constraints/size.rs:66-116 |
//! | `completeness`, `any_complete`, `at_least_complete`, `exactly_complete` | COUNT(c) per column | constraints/completeness.rs:137-246, core/unified.rs:41-123 |
//! | `statistic`, `has_min/max/mean/sum/standard_deviation/variance` | NUMERIC_STATS(c) [VARIANCE]   | constraints/statistics.rs:254-322 |
//! | `statistic(Median / Percentile)`, `has_approx_quantile` | KLL(c, k = 200)                       | constraints/quantile.rs:282-345 (t-digest there) |
//! | `uniqueness`, `validates_uniqueness / distinctness / unique_value_ratio / primary_key / uniqueness_with_nulls` | DISTINCT(c or (a, b, ..)) [MULTIPLICITY] | constraints/uniqueness.rs:449-482, 549-718, 730-851 |
//! | `has_format`, `validates_regex / email / url / credit_card / phone / postal_code / uuid / ipv4 / ipv6 / json / iso8601_datetime` (+ `_with_options`) | REGEX_MATCH(c, pattern, TRIM / CASE_INSENSITIVE / NULL_IS_VALID) | constraints/format.rs:740-843 |
//! | `has_min_length / max_length / length_between / exact_length`, `is_not_empty` | LENGTH(c, min, max) | constraints/length.rs:167-196 |
//! | containment (`ContainmentConstraint::new`)          | COUNT(c) + REGEX_MATCH(c, `^(?:a|b|..)$`) | constraints/values.rs:245-291 |
//! | `has_approx_count_distinct`                         | APPROX_DISTINCT(c)                        | constraints/approx_count_distinct.rs:53-120 |
//! | `has_correlation` (Pearson), covariance, independence | COMOMENTS(a, b)                         | constraints/correlation.rs:299-444 |
use crate::column::{column_view, string_typed, unused_column, validity_only_view, ColumnView};
use crate::handles::{Error, Plan, Spec, State};
use crate::sys::*;
use arrow::datatypes::DataType;
use async_trait::async_trait;
use datafusion::prelude::SessionContext;
use futures::StreamExt;
use std::collections::HashMap;
use std::sync::{Arc, Mutex, RwLock};
use term_guard::constraints::{
    ApproxCountDistinctConstraint, Assertion, CompletenessConstraint, ContainmentConstraint, CorrelationConstraint,
    FormatConstraint, FormatOptions, FormatType, LengthAssertion, LengthConstraint, NullHandling, QuantileConstraint,
    SizeConstraint, StatisticType, StatisticalConstraint, UniquenessConstraint, UniquenessOptions, UniquenessType,
};
use term_guard::core::{
    current_validation_context, Constraint, ConstraintMetadata, ConstraintResult, LogicalOperator, ValidationResult,
    ValidationSuite,
};
use term_guard::error::{Result as TermResult, TermError};

const KLL_K: u32 = 200; // rank error bound 1.65 / sqrt(k) (analyzers/advanced/kll_sketch.rs:397-399)

/// What one constraint asks of the pass, by column NAME (indices are assigned when the plan is built).
#[derive(Debug, Clone, Default)]
struct Request {
    kind: i32,
    column: String,
    column2: Option<String>,
    columns: Vec<String>, // DISTINCT over a tuple
    flags: u32,
    pattern: String,
    kll_k: u32,
    length_min: u64,
    length_max: u64,
}

/// The verdict a constraint applies to its aggregates.
#[derive(Debug, Clone)]
enum Verdict {
    Size(Assertion),
    Completeness { columns: Vec<String>, op: LogicalOperator, threshold: f64 },
    Statistic { stat: StatisticType, assertion: Assertion },
    Uniqueness { columns: Vec<String>, kind: UniquenessType },
    Format { format: FormatType, threshold: f64 },
    Length(LengthAssertion),
    Containment,
    ApproxCountDistinct { column: String, assertion: Assertion },
    Quantile { quantile: f64, assertion: Assertion },
    Pearson { a: String, b: String, assertion: Assertion },
    Covariance { a: String, b: String, assertion: Assertion },
    Independence { a: String, b: String, max_correlation: f64 },
}

struct Binding {
    requests: Vec<Request>,
    verdict: Verdict,
}

/// What a run left for the constraints: per binding, the results of its requests (and the sketch handle for quantiles).
struct RunOutput {
    per_binding: Vec<Option<Vec<tgx_result>>>, // None: the binding went unplanned (unsupported): use the stock constraint
    quantiles: HashMap<(usize, u64), f64>,     // (binding, quantile bits) -> value, read before the state went away
}

struct Shared {
    bindings: Mutex<Vec<Binding>>,
    output: RwLock<Option<RunOutput>>,
    /// see [`GpuPlanner::strict_reference_types`]
    strict_types: std::sync::atomic::AtomicBool,
    exact_keys: std::sync::atomic::AtomicBool,
}

/// Can the stock constraint read its aggregate off a column of this type?  `StatisticalConstraint::evaluate` downcasts
/// the result to `Int64Array`, then `Float64Array`, else `Err("Failed to extract statistic value")`
/// (constraints/statistics.rs:277-308); DataFusion's MIN / MAX / APPROX_PERCENTILE_CONT keep the input type, SUM gives
/// Int64 for signed integers, UInt64 for unsigned ones, Float64 for floats, AVG / STDDEV / VARIANCE give Float64.
/// `QuantileConstraint` also reads `Int32Array` (constraints/quantile.rs:308-324).
fn reference_extracts(v: &Verdict, t: &DataType) -> bool {
    use DataType::*;
    let sint = matches!(t, Int8 | Int16 | Int32);
    let uint = matches!(t, UInt8 | UInt16 | UInt32 | UInt64);
    let flt = matches!(t, Float16 | Float32);
    if matches!(t, Int64 | Float64) {
        return true;
    }
    match v {
        Verdict::Statistic { stat, .. } => match stat {
            StatisticType::Min | StatisticType::Max | StatisticType::Median | StatisticType::Percentile(_) => false,
            StatisticType::Sum => sint || flt,
            StatisticType::Mean | StatisticType::StandardDeviation | StatisticType::Variance => sint || uint || flt,
        },
        Verdict::Quantile { .. } => matches!(t, Int32),
        // pattern / length / containment checks are for string columns (a Binary column has the string LAYOUT here, but
        // `~` and LENGTH on it are the reference's to refuse)
        Verdict::Format { .. } | Verdict::Length(_) | Verdict::Containment => string_typed(t),
        _ => true, // counts: COUNT / COUNT(DISTINCT) come back as Int64 whatever the column
    }
}

/// Factory of GPU-answered constraints, registry of what they need, and the runner of the fused pass.
#[derive(Clone)]
pub struct GpuPlanner {
    shared: Arc<Shared>,
}

/// A constraint handed out by a [`GpuPlanner`]: answers from the planner's last run, else through `stock`.
#[derive(Debug)]
pub struct GpuConstraint {
    id: usize,
    stock: Arc<dyn Constraint>,
    shared: Arc<SharedDebug>,
}
// (`Constraint: Debug`; the shared state prints as its address)
struct SharedDebug(Arc<Shared>);
impl std::fmt::Debug for SharedDebug {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "GpuPlanner@{:p}", Arc::as_ptr(&self.0))
    }
}

#[async_trait]
impl Constraint for GpuConstraint {
    async fn evaluate(&self, ctx: &SessionContext) -> TermResult<ConstraintResult> {
        let verdict = {
            let out = self.shared.0.output.read().unwrap();
            match out.as_ref().and_then(|o| o.per_binding.get(self.id).and_then(|r| r.as_ref().map(|r| (r.clone(), o)))) {
                Some((results, o)) => {
                    let bindings = self.shared.0.bindings.lock().unwrap();
                    Some(apply(&bindings[self.id].verdict, &results, |q| o.quantiles.get(&(self.id, q.to_bits())).copied()))
                }
                None => None,
            }
        };
        match verdict {
            Some(v) => Ok(v),
            None => self.stock.evaluate(ctx).await, // not planned, or the planner was not run: the stock SQL
        }
    }
    fn name(&self) -> &str {
        self.stock.name()
    }
    fn column(&self) -> Option<&str> {
        self.stock.column()
    }
    fn description(&self) -> Option<&str> {
        self.stock.description()
    }
    fn metadata(&self) -> ConstraintMetadata {
        self.stock.metadata().with_custom("backend", "tgx")
    }
}

impl Default for GpuPlanner {
    fn default() -> Self {
        Self::new()
    }
}

impl GpuPlanner {
    pub fn new() -> Self {
        GpuPlanner {
            shared: Arc::new(Shared {
                bindings: Mutex::new(Vec::new()),
                output: RwLock::new(None),
                strict_types: std::sync::atomic::AtomicBool::new(true),
                exact_keys: std::sync::atomic::AtomicBool::new(true),
            }),
        }
    }

    /// `true` (the default): a statistic or quantile whose aggregate the stock constraint could not read off the
    /// column's type -- MIN / MAX of an Int32, Date32, Float32 or Timestamp column, SUM of a UInt column, ... -- is NOT
    /// answered from the device: the binding stays unplanned and the stock constraint returns the reference's own
    /// `Err("Failed to extract statistic value")` (a failed check, "Error evaluating constraint: ..").  `false`: the
    /// kernels' value for the widened column answers -- a deviation from the reference, listed in INTEGRATION.md.
    pub fn strict_reference_types(&self, on: bool) -> &Self {
        self.shared.strict_types.store(on, std::sync::atomic::Ordering::Relaxed);
        self
    }

    /// `true` (the default): uniqueness checks over string / binary / tuple keys count by VALUE -- the library keeps
    /// the bytes of every distinct key and confirms equal fingerprints byte by byte (`TGX_FLAG_EXACT_KEYS`), which is
    /// `COUNT(DISTINCT c)` as DataFusion's hash aggregation computes it (constraints/uniqueness.rs:612-617, 709-715).
    /// `false`: by keyed 128-bit fingerprint alone -- faster on big batches; two distinct values count once only if all
    /// 128 bits agree under a key the data's producer does not know (deviations).  Applies to
    /// uniqueness constraints created after the call.
    pub fn exact_string_keys(&self, on: bool) -> &Self {
        self.shared.exact_keys.store(on, std::sync::atomic::Ordering::Relaxed);
        self
    }

    fn bind(&self, stock: Arc<dyn Constraint>, requests: Vec<Request>, verdict: Verdict) -> GpuConstraint {
        let mut b = self.shared.bindings.lock().unwrap();
        b.push(Binding { requests, verdict });
        GpuConstraint { id: b.len() - 1, stock, shared: Arc::new(SharedDebug(self.shared.clone())) }
    }

    // ---- the builder calls of CheckBuilder (core/check.rs), one by one ---------------------------------------------
    /// `has_size` (check.rs:321): COUNT(*) rides on the table's first column.
    pub fn has_size(&self, assertion: Assertion) -> GpuConstraint {
        let req = Request { kind: TGX_CHECK_COUNT, ..Default::default() }; // column "": the first column of the table
        self.bind(Arc::new(SizeConstraint::new(assertion.clone())), vec![req], Verdict::Size(assertion))
    }

    /// `completeness` / `any_complete` / `at_least_complete` / `exactly_complete` (check.rs:1743, 2233-2300).
    pub fn completeness<I, S>(&self, columns: I, op: LogicalOperator, threshold: f64) -> GpuConstraint
    where
        I: IntoIterator<Item = S>,
        S: Into<String>,
    {
        let columns: Vec<String> = columns.into_iter().map(Into::into).collect();
        let stock = CompletenessConstraint::with_operator(columns.clone(), op.clone(), threshold);
        let requests = columns.iter().map(|c| Request { kind: TGX_CHECK_COUNT, column: c.clone(), ..Default::default() }).collect();
        self.bind(Arc::new(stock), requests, Verdict::Completeness { columns, op, threshold })
    }

    /// `statistic` and `has_min / has_max / has_mean / has_sum / has_standard_deviation / has_variance`
    /// (check.rs:1812-1960); Median / Percentile go to the KLL sketch (APPROX_PERCENTILE_CONT in the reference).
    pub fn statistic(&self, column: impl Into<String>, stat: StatisticType, assertion: Assertion) -> TermResult<GpuConstraint> {
        let column = column.into();
        let stock = StatisticalConstraint::new(column.clone(), stat.clone(), assertion.clone())?;
        let mut req = Request { column, ..Default::default() };
        match stat {
            StatisticType::Median | StatisticType::Percentile(_) => {
                req.kind = TGX_CHECK_KLL;
                req.kll_k = KLL_K;
            }
            StatisticType::StandardDeviation | StatisticType::Variance => {
                req.kind = TGX_CHECK_NUMERIC_STATS;
                req.flags = TGX_FLAG_VARIANCE;
            }
            _ => req.kind = TGX_CHECK_NUMERIC_STATS,
        }
        Ok(self.bind(Arc::new(stock), vec![req], Verdict::Statistic { stat, assertion }))
    }

    /// `uniqueness` and the `validates_uniqueness / distinctness / unique_value_ratio / primary_key /
    /// uniqueness_with_nulls` family (check.rs:1480-1740).  `UniqueComposite` stays with the stock constraint.
    /// String / binary / tuple keys are counted BY VALUE (`TGX_FLAG_EXACT_KEYS`: keyed fingerprints confirmed byte by
    /// byte -- `COUNT(DISTINCT c)` as the stock constraint's DataFusion query computes it, uniqueness.rs:612-617) unless
    /// [`GpuPlanner::exact_string_keys`] was turned off.  A run is one state fed by one stream, so nothing of it
    /// crosses a state boundary (where keys would travel as fingerprints: INTEGRATION.md section 2).
    pub fn uniqueness<I, S>(&self, columns: I, kind: UniquenessType) -> TermResult<GpuConstraint>
    where
        I: IntoIterator<Item = S>,
        S: Into<String>,
    {
        let columns: Vec<String> = columns.into_iter().map(Into::into).collect();
        let stock = UniquenessConstraint::new(columns.clone(), kind.clone(), UniquenessOptions::default())?;
        let planned = !matches!(kind, UniquenessType::UniqueComposite { .. }) && columns.len() <= 8;
        let mut req = Request { kind: TGX_CHECK_DISTINCT, column: columns[0].clone(), ..Default::default() };
        if columns.len() >= 2 {
            req.columns = columns.clone(); // COUNT(DISTINCT (a, b)): the tuple is one value (uniqueness.rs:557-562)
        }
        if matches!(kind, UniquenessType::UniqueValueRatio(_)) {
            req.flags = TGX_FLAG_MULTIPLICITY;
        }
        if self.shared.exact_keys.load(std::sync::atomic::Ordering::Relaxed) {
            req.flags |= TGX_FLAG_EXACT_KEYS;
        }
        Ok(self.bind(Arc::new(stock), if planned { vec![req] } else { vec![] }, Verdict::Uniqueness { columns, kind }))
    }

    /// `has_format` and every `validates_*` helper (check.rs:829-1470): the helpers only pick the `FormatType` and the
    /// options -- `validates_email` = Email with trimming and NULLs invalid (builder_extensions.rs:309-318).
    pub fn format(&self, column: impl Into<String>, format: FormatType, threshold: f64, options: FormatOptions) -> TermResult<GpuConstraint> {
        let column = column.into();
        let stock = FormatConstraint::new(column.clone(), format.clone(), threshold, options.clone())?;
        let req = Request {
            kind: TGX_CHECK_REGEX_MATCH,
            column,
            pattern: crate::patterns::of(&format),
            flags: (if options.case_sensitive { 0 } else { TGX_FLAG_CASE_INSENSITIVE })
                | (if options.trim_before_check { TGX_FLAG_TRIM } else { 0 })
                | (if options.null_is_valid { TGX_FLAG_NULL_IS_VALID } else { 0 }),
            ..Default::default()
        };
        Ok(self.bind(Arc::new(stock), vec![req], Verdict::Format { format, threshold }))
    }

    /// `has_min_length / has_max_length / has_length_between / has_exact_length / is_not_empty` (check.rs:518-640).
    pub fn length(&self, column: impl Into<String>, assertion: LengthAssertion) -> GpuConstraint {
        let column = column.into();
        let stock = LengthConstraint::new(column.clone(), assertion.clone());
        let (lo, hi) = match assertion {
            LengthAssertion::Min(n) => (n as u64, u64::MAX),
            LengthAssertion::Max(n) => (0, n as u64),
            LengthAssertion::Between(a, b) => (a as u64, b as u64),
            LengthAssertion::Exactly(n) => (n as u64, n as u64),
            LengthAssertion::NotEmpty => (1, u64::MAX),
        };
        let req = Request { kind: TGX_CHECK_LENGTH, column, length_min: lo, length_max: hi, ..Default::default() };
        self.bind(Arc::new(stock), vec![req], Verdict::Length(assertion))
    }

    /// `ContainmentConstraint` (constraints/values.rs:200-330): the IN-list as the anchored alternation of the escaped
    /// literals (Rust's `$` is the end of the text: exact string equality); NULL rows are outside the WHERE clause.
    pub fn containment<I, S>(&self, column: impl Into<String>, allowed: I) -> GpuConstraint
    where
        I: IntoIterator<Item = S>,
        S: Into<String>,
    {
        let column = column.into();
        let allowed: Vec<String> = allowed.into_iter().map(Into::into).collect();
        let stock = ContainmentConstraint::new(column.clone(), allowed.clone());
        let escape = |lit: &str| -> String {
            let mut o = String::new();
            for ch in lit.chars() {
                if "\\.+*?()|[]{}^$#&-~".contains(ch) {
                    o.push('\\');
                }
                o.push(ch);
            }
            o
        };
        let pattern = format!("^(?:{})$", allowed.iter().map(|a| escape(a)).collect::<Vec<_>>().join("|"));
        let requests = vec![
            Request { kind: TGX_CHECK_COUNT, column: column.clone(), ..Default::default() },
            Request { kind: TGX_CHECK_REGEX_MATCH, column, pattern, ..Default::default() },
        ];
        self.bind(Arc::new(stock), if allowed.is_empty() { vec![] } else { requests }, Verdict::Containment)
    }

    /// `has_approx_count_distinct` (check.rs:379): a HyperLogLog lane of the column's scan (2^14 registers, as
    /// DataFusion's); the exact count on string columns.
    pub fn approx_count_distinct(&self, column: impl Into<String>, assertion: Assertion) -> GpuConstraint {
        let column = column.into();
        let stock = ApproxCountDistinctConstraint::new(column.clone(), assertion.clone());
        let req = Request { kind: TGX_CHECK_APPROX_DISTINCT, column: column.clone(), ..Default::default() };
        self.bind(Arc::new(stock), vec![req], Verdict::ApproxCountDistinct { column, assertion })
    }

    /// `has_approx_quantile` (check.rs:414): one KLL sketch per column answers every quantile asked of it.
    pub fn approx_quantile(&self, column: impl Into<String>, quantile: f64, assertion: Assertion) -> TermResult<GpuConstraint> {
        let column = column.into();
        let stock = QuantileConstraint::percentile(column.clone(), quantile, assertion.clone())?;
        let req = Request { kind: TGX_CHECK_KLL, column, kll_k: KLL_K, ..Default::default() };
        Ok(self.bind(Arc::new(stock), vec![req], Verdict::Quantile { quantile, assertion }))
    }

    /// `has_correlation` (check.rs:478): `CORR(a, b)` from the centred co-moments of one pass.
    pub fn pearson(&self, a: impl Into<String>, b: impl Into<String>, assertion: Assertion) -> TermResult<GpuConstraint> {
        let (a, b) = (a.into(), b.into());
        let stock = CorrelationConstraint::pearson(a.clone(), b.clone(), assertion.clone())?;
        let req = Request { kind: TGX_CHECK_COMOMENTS, column: a.clone(), column2: Some(b.clone()), ..Default::default() };
        Ok(self.bind(Arc::new(stock), vec![req], Verdict::Pearson { a, b, assertion }))
    }

    /// `CorrelationConstraint::independence` (constraints/correlation.rs:242): `ABS(CORR(a, b)) <= max`.
    pub fn independence(&self, a: impl Into<String>, b: impl Into<String>, max_correlation: f64) -> TermResult<GpuConstraint> {
        let (a, b) = (a.into(), b.into());
        let stock = CorrelationConstraint::independence(a.clone(), b.clone(), max_correlation)?;
        let req = Request { kind: TGX_CHECK_COMOMENTS, column: a.clone(), column2: Some(b.clone()), ..Default::default() };
        Ok(self.bind(Arc::new(stock), vec![req], Verdict::Independence { a, b, max_correlation }))
    }

    // ---- the pass -------------------------------------------------------------------------------------------------
    /// One scan of the suite's table through the fused plan, then `suite.run(ctx)` (core/suite.rs:399): the tally
    /// loop, the issues and the report are the reference's own.
    pub async fn run(&self, suite: &ValidationSuite, ctx: &SessionContext) -> TermResult<ValidationResult> {
        self.scan(ctx, suite_table(suite)).await?;
        let result = suite.run(ctx).await;
        *self.shared.output.write().unwrap() = None; // the aggregates belong to this run's data
        result
    }

    /// The pass alone: afterwards the planner's constraints answer from it until [`GpuPlanner::clear`].
    pub async fn scan(&self, ctx: &SessionContext, table: String) -> TermResult<()> {
        let df = ctx.table(&table).await?;
        let schema = df.schema().clone();
        let names: Vec<String> = schema.fields().iter().map(|f| f.name().clone()).collect();
        let index = |name: &str| -> Option<i32> {
            if name.is_empty() {
                return if names.is_empty() { None } else { Some(0) };
            }
            names.iter().position(|n| n == name).map(|i| i as i32)
        };
        // 1. constraints -> specs.  A binding whose columns are missing, or that the library refuses, stays unplanned.
        let (mut specs, mut owner): (Vec<Spec>, Vec<(usize, usize)>) = (Vec::new(), Vec::new());
        let mut planned: Vec<bool> = Vec::new();
        {
            let bindings = self.shared.bindings.lock().unwrap();
            let strict = self.shared.strict_types.load(std::sync::atomic::Ordering::Relaxed);
            for (id, b) in bindings.iter().enumerate() {
                let mut mine = Vec::new();
                for r in &b.requests {
                    // the reference's result-type rule: what it would fail to read is left to it
                    if strict {
                        if let Some(c) = index(&r.column) {
                            if !reference_extracts(&b.verdict, schema.field(c as usize).data_type()) {
                                break;
                            }
                        }
                    }
                    let (Some(c), c2) = (index(&r.column), r.column2.as_deref().map(index)) else { break };
                    if matches!(c2, Some(None)) {
                        break;
                    }
                    let tuple: Option<Vec<i32>> = r.columns.iter().map(|n| index(n)).collect();
                    let Some(tuple) = tuple else { break };
                    let spec = Spec {
                        kind: r.kind,
                        column: c,
                        column2: c2.flatten().unwrap_or(-1),
                        flags: r.flags,
                        pattern: r.pattern.clone().into_bytes(),
                        kll_k: r.kll_k,
                        columns: tuple,
                        length_min: r.length_min,
                        length_max: r.length_max,
                    };
                    // (a spec of its own is planned alone first: TGX_UNSUPPORTED must not take the suite's plan down)
                    match Plan::new(std::slice::from_ref(&spec)) {
                        Ok(_) => mine.push(spec),
                        Err(e) if e.is_unsupported() || e.status == TGX_INVALID_ARGUMENT => break,
                        Err(e) => return Err(internal(e)),
                    }
                }
                let ok = !b.requests.is_empty() && mine.len() == b.requests.len();
                planned.push(ok);
                if ok {
                    for (k, s) in mine.into_iter().enumerate() {
                        // identical requests of several constraints share one spec
                        let at = specs.iter().position(|have| *have == s).unwrap_or_else(|| {
                            specs.push(s);
                            specs.len() - 1
                        });
                        owner.push((id, at));
                        let _ = k;
                    }
                }
            }
        }
        if specs.is_empty() {
            *self.shared.output.write().unwrap() = None;
            return Ok(());
        }
        // `held` is declared BEFORE the state on purpose: locals drop in reverse order, so on every exit of this
        // function (an error from the stream, a refused column, the future being cancelled at an `.await`) the state
        // is destroyed first -- tgx_state_destroy waits for the library's copy threads -- and only then are the
        // retained RecordBatches released.  The other order frees Arrow buffers the copy threads may still be reading.
        let mut held: std::collections::VecDeque<(datafusion::arrow::record_batch::RecordBatch, Vec<Option<ColumnView>>)> =
            std::collections::VecDeque::new();
        let plan = Plan::new(&specs).map_err(internal)?;
        let mut state = State::new(&plan).map_err(internal)?;
        let used: Vec<bool> = (0..names.len() as i32)
            .map(|i| specs.iter().any(|s| s.column == i || s.column2 == i || s.columns.contains(&i)))
            .collect();
        // columns only completeness / size look at: any Arrow type will do (validity + length)
        let reads_values: Vec<bool> = (0..names.len() as i32)
            .map(|i| specs.iter().any(|s| s.kind != TGX_CHECK_COUNT && (s.column == i || s.column2 == i || s.columns.contains(&i))))
            .collect();
        // 2. one scan of the table instead of one per constraint; batches as DataFusion makes them (8192 rows,
        //    core/context.rs:28-38): the library coalesces them
        //    The batches are handed over as TGX_MEM_HOST_RETAINED: a RecordBatch is a set of Arc'd buffers, so holding
        //    the ones the library has only NOTED (`State::pending`: the last few hundred at most -- it flushes every few
        //    tens of MB) costs nothing, and their copy into pinned memory then happens at the flush, on the library's
        //    copy threads, instead of window by window inside `update` on this thread.
        let mut stream = df.execute_stream().await?;
        while let Some(batch) = stream.next().await {
            let batch = batch?;
            // DataFusion streams interleave empty batches (filters, repartitions).  tgx_update returns early for them
            // and notes nothing, so they must not enter `held` either: `held` mirrors the library's NOTED batches one
            // to one, and an extra entry would push the oldest still-pending batch out of the window below.
            if batch.num_rows() == 0 {
                continue;
            }
            let mut views: Vec<Option<ColumnView>> = Vec::with_capacity(names.len());
            for (i, col) in batch.columns().iter().enumerate() {
                views.push(if !used[i] {
                    None
                } else if reads_values[i] {
                    column_view(col)
                } else {
                    Some(column_view(col).unwrap_or_else(|| validity_only_view(col)))
                });
            }
            if views.iter().zip(&used).any(|(v, u)| *u && v.is_none()) {
                // a column type outside the path appeared: nothing of this run is answered from the device
                *self.shared.output.write().unwrap() = None;
                return Ok(());
            }
            let raw: Vec<tgx_column> = views
                .iter()
                .map(|v| v.as_ref().map(|v| v.retained()).unwrap_or_else(unused_column))
                .collect();
            state.update(&raw).map_err(internal)?;
            held.push_back((batch, views)); // (the views own what `raw` pointed into: re-aligned bitmaps, buffer tables)
            let (pending, _) = state.pending();
            while held.len() as u64 > pending {
                held.pop_front();
            }
        }
        // 3. finalize; the quantiles are read while the sketches exist
        let results = state.finalize().map_err(internal)?;
        let mut out = RunOutput { per_binding: planned.iter().map(|p| if *p { Some(Vec::new()) } else { None }).collect(), quantiles: HashMap::new() };
        for (id, at) in &owner {
            out.per_binding[*id].as_mut().unwrap().push(results[*at]);
        }
        {
            let bindings = self.shared.bindings.lock().unwrap();
            let mut seen = vec![0usize; bindings.len()];
            for (id, at) in &owner {
                let k = seen[*id];
                seen[*id] += 1;
                let qs: Vec<f64> = match &bindings[*id].verdict {
                    Verdict::Quantile { quantile, .. } => vec![*quantile],
                    Verdict::Statistic { stat: StatisticType::Median, .. } => vec![0.5],
                    Verdict::Statistic { stat: StatisticType::Percentile(p), .. } => vec![*p],
                    _ => vec![],
                };
                if k == 0 && specs[*at].kind == TGX_CHECK_KLL && results[*at].kll_n > 0 {
                    for q in qs {
                        out.quantiles.insert((*id, q.to_bits()), state.kll_quantile(*at, q).map_err(internal)?);
                    }
                }
            }
        }
        *self.shared.output.write().unwrap() = Some(out);
        Ok(())
    }

    pub fn clear(&self) {
        *self.shared.output.write().unwrap() = None;
    }
}

fn internal(e: Error) -> TermError {
    TermError::Internal(e.to_string()) // run_sequential turns it into "Error evaluating constraint: {e}" (suite.rs:231-256)
}

fn suite_table(_suite: &ValidationSuite) -> String {
    // ValidationSuite::run sets the validation context from its own table name (suite.rs:582, default "data"); outside
    // a run the context's default is the same name
    current_validation_context().table_name().to_string()
}

// ---- verdicts: the post-processing of each stock constraint's evaluate(), on shared aggregates --------------------
fn apply(v: &Verdict, r: &[tgx_result], quantile: impl Fn(f64) -> Option<f64>) -> ConstraintResult {
    match v {
        // constraints/size.rs:66-116: an empty table evaluates on 0, it is not skipped
        Verdict::Size(a) => {
            let rows = r[0].total as f64;
            if a.evaluate(rows) {
                ConstraintResult::success_with_metric(rows)
            } else {
                ConstraintResult::failure_with_metric(rows, format!("Size {rows} does not {}", a.description()))
            }
        }
        // constraints/completeness.rs:170-246 per column, core/unified.rs:50-121 across columns
        Verdict::Completeness { columns, op, threshold } => {
            let one = |col: &str, r: &tgx_result| -> ConstraintResult {
                if r.total == 0 {
                    return ConstraintResult::skipped("No data to validate");
                }
                let c = r.non_null as f64 / r.total as f64;
                if c >= *threshold {
                    ConstraintResult::success_with_metric(c)
                } else {
                    ConstraintResult::failure_with_metric(
                        c,
                        format!("Column '{col}' completeness {:.2}% is below threshold {:.2}%", c * 100.0, threshold * 100.0),
                    )
                }
            };
            if columns.is_empty() {
                return ConstraintResult::skipped("No columns specified");
            }
            if columns.len() == 1 {
                return one(&columns[0], &r[0]);
            }
            let each: Vec<ConstraintResult> = columns.iter().zip(r).map(|(c, r)| one(c, r)).collect();
            let oks: Vec<bool> = each.iter().map(|e| e.status.is_success()).collect();
            let metrics: Vec<f64> = each.iter().filter_map(|e| e.metric).collect();
            let metric = if metrics.is_empty() { None } else { Some(metrics.iter().sum::<f64>() / metrics.len() as f64) };
            let names = |want: bool| -> String {
                columns.iter().zip(&oks).filter(|(_, ok)| **ok == want).map(|(c, _)| c.as_str()).collect::<Vec<_>>().join(", ")
            };
            let mut out = if op.evaluate(&oks) {
                let mut s = ConstraintResult::success();
                s.message = match op {
                    LogicalOperator::All => Some(format!("All {} columns satisfy the constraint", columns.len())),
                    LogicalOperator::Any => Some(format!("Columns {} satisfy the constraint", names(true))),
                    _ => None,
                };
                s
            } else {
                ConstraintResult::failure(format!("Constraint failed for columns: {}. Required: {}", names(false), op.description()))
            };
            out.metric = metric;
            out
        }
        // constraints/statistics.rs:278-320: Float64, else Int64 cast to f64; NULL aggregate => Failure
        Verdict::Statistic { stat, assertion } => {
            let a = &r[0];
            let value = match stat {
                StatisticType::Min if a.has_value != 0 => Some(a.min_f),
                StatisticType::Max if a.has_value != 0 => Some(a.max_f),
                StatisticType::Mean if a.has_value != 0 => Some(a.mean),
                StatisticType::Sum if a.has_value != 0 => Some(if a.is_float != 0 { a.sum_f } else { a.sum_i as f64 }),
                StatisticType::StandardDeviation if a.has_variance != 0 => Some(a.stddev_samp),
                StatisticType::Variance if a.has_variance != 0 => Some(a.var_samp),
                StatisticType::Median if a.kll_n > 0 => quantile(0.5),
                StatisticType::Percentile(p) if a.kll_n > 0 => quantile(*p),
                _ => None,
            };
            let name = stat.name(); // "minimum", "standard deviation", .. (statistics.rs:77-94)
            match value {
                None => ConstraintResult::failure(format!("{name} is null (no non-null values)")),
                Some(v) if assertion.evaluate(v) => ConstraintResult::success_with_metric(v),
                Some(v) => ConstraintResult::failure_with_metric(v, format!("{name} {v} does not {}", assertion.description())),
            }
        }
        // constraints/uniqueness.rs:730-851
        Verdict::Uniqueness { columns, kind } => {
            let a = &r[0];
            let total = a.total as f64;
            if total == 0.0 {
                return ConstraintResult::skipped("No data to validate");
            }
            let nulls = (a.total - a.non_null) as f64;
            let cols = columns.join(", ");
            match kind {
                UniquenessType::FullUniqueness { threshold } | UniquenessType::UniqueWithNulls { threshold, .. } => {
                    let mut unique = a.distinct as f64;
                    if let UniquenessType::UniqueWithNulls { null_handling, .. } = kind {
                        if columns.len() == 1 {
                            // Include: COUNT(DISTINCT COALESCE(c, '<NULL>')) (:572-577); Distinct: + (COUNT(*) - COUNT(c)) (:594-598)
                            match null_handling {
                                NullHandling::Include => unique += if nulls > 0.0 { 1.0 } else { 0.0 },
                                NullHandling::Distinct => unique += nulls,
                                _ => {}
                            }
                        }
                    }
                    let ratio = unique / total;
                    if ratio >= *threshold {
                        ConstraintResult::success_with_metric(ratio)
                    } else {
                        ConstraintResult::failure_with_metric(
                            ratio,
                            format!("Uniqueness ratio {ratio:.3} is below threshold {threshold:.3} for columns: {cols}"),
                        )
                    }
                }
                UniquenessType::Distinctness(assertion) | UniquenessType::UniqueValueRatio(assertion) => {
                    let count = if matches!(kind, UniquenessType::Distinctness(_)) { a.distinct } else { a.groups_once } as f64;
                    let ratio = count / total;
                    if assertion.evaluate(ratio) {
                        ConstraintResult::success_with_metric(ratio)
                    } else {
                        ConstraintResult::failure_with_metric(
                            ratio,
                            format!("{} ratio {ratio:.3} does not satisfy {} for columns: {cols}", kind.name(), assertion.description()),
                        )
                    }
                }
                UniquenessType::PrimaryKey => {
                    let unique = a.distinct as f64;
                    if nulls > 0.0 {
                        ConstraintResult::failure_with_metric(nulls / total, format!("Primary key columns contain {nulls} NULL values: {cols}"))
                    } else if unique != total {
                        ConstraintResult::failure_with_metric(
                            (total - unique) / total,
                            format!("Primary key columns contain {} duplicate values: {cols}", total - unique),
                        )
                    } else {
                        ConstraintResult::success_with_metric(1.0)
                    }
                }
                _ => ConstraintResult::success(),
            }
        }
        // constraints/format.rs:780-843
        Verdict::Format { format, threshold } => {
            let a = &r[0];
            let total = a.total as f64;
            if total == 0.0 {
                return ConstraintResult::skipped("No data to validate");
            }
            let ratio = a.matches as f64 / total;
            let detect = matches!(format, FormatType::CreditCard { detect_only: true });
            if (detect && ratio <= *threshold) || (!detect && ratio >= *threshold) {
                ConstraintResult::success_with_metric(ratio)
            } else if detect {
                ConstraintResult::failure_with_metric(ratio, format!("Credit card detection ratio {ratio:.3} exceeds threshold {threshold:.3}"))
            } else {
                ConstraintResult::failure_with_metric(
                    ratio,
                    format!("Format validation ratio {ratio:.3} is below threshold {threshold:.3} - values that {}", format.description()),
                )
            }
        }
        // constraints/length.rs:167-196: the ratio must reach 1.0
        Verdict::Length(assertion) => {
            let a = &r[0];
            if a.total == 0 {
                return ConstraintResult::skipped("No data to validate");
            }
            let ratio = a.matches as f64 / a.total as f64;
            if ratio >= 1.0 {
                ConstraintResult::success_with_metric(ratio)
            } else {
                ConstraintResult::failure_with_metric(
                    ratio,
                    format!("Length constraint failed: {:.2}% of values are {}", ratio * 100.0, assertion.description()),
                )
            }
        }
        // constraints/values.rs:245-291: denominator = non-NULL rows; must be exactly 1.0
        Verdict::Containment => {
            let (total, valid) = (r[0].non_null as f64, r[1].matches as f64);
            if total == 0.0 {
                return ConstraintResult::skipped("No non-null data to validate");
            }
            let ratio = valid / total;
            if ratio == 1.0 {
                ConstraintResult::success_with_metric(ratio)
            } else {
                ConstraintResult::failure_with_metric(ratio, format!("{} values are not in the allowed set", total - valid))
            }
        }
        // constraints/approx_count_distinct.rs:75-120: an empty column gives 0, it is not skipped
        Verdict::ApproxCountDistinct { column, assertion } => {
            let count = r[0].distinct as f64;
            if assertion.evaluate(count) {
                ConstraintResult::success_with_metric(count)
            } else {
                ConstraintResult::failure_with_metric(
                    count,
                    format!("Approximate distinct count {count} does not satisfy assertion {} for column '{column}'", assertion.description()),
                )
            }
        }
        // constraints/quantile.rs:287-345
        Verdict::Quantile { quantile: q, assertion } => {
            if r[0].kll_n == 0 {
                return ConstraintResult::skipped("No data to validate");
            }
            let value = quantile(*q).unwrap_or(f64::NAN);
            if assertion.evaluate(value) {
                ConstraintResult::success_with_metric(value)
            } else {
                ConstraintResult::failure_with_metric(value, format!("Quantile {q} is {value} which does not {}", assertion.description()))
            }
        }
        // constraints/correlation.rs:343-396: CORR = population covariance over the population deviations (DataFusion's
        // online accumulators end on the centred moments the library returns), 0 when a deviation is 0
        Verdict::Pearson { a, b, assertion } | Verdict::Covariance { a, b, assertion } => {
            if r[0].total == 0 {
                return ConstraintResult::skipped("No data to validate");
            }
            let covariance = matches!(v, Verdict::Covariance { .. });
            let value = if covariance { covar_samp(&r[0]) } else { pearson(&r[0]) };
            let label = if covariance { "Covariance" } else { "Pearson correlation" };
            if assertion.evaluate(value) {
                ConstraintResult::success_with_metric(value)
            } else {
                ConstraintResult::failure_with_metric(value, format!("{label} between {a} and {b} is {value} which does not {}", assertion.description()))
            }
        }
        // constraints/correlation.rs:397-439
        Verdict::Independence { a, b, max_correlation } => {
            if r[0].total == 0 {
                return ConstraintResult::skipped("No data to validate");
            }
            let abs = pearson(&r[0]).abs();
            if abs <= *max_correlation {
                ConstraintResult::success_with_metric(abs)
            } else {
                ConstraintResult::failure_with_metric(
                    abs,
                    format!("Columns {a} and {b} have correlation {abs} exceeding independence threshold {max_correlation}"),
                )
            }
        }
    }
}

fn pearson(r: &tgx_result) -> f64 {
    let n = r.non_null as f64;
    if n < 1.0 {
        return 0.0;
    }
    let cov = r.co_c_xy / n;
    let sx = if r.co_m2_x > 0.0 { (r.co_m2_x / n).sqrt() } else { 0.0 };
    let sy = if r.co_m2_y > 0.0 { (r.co_m2_y / n).sqrt() } else { 0.0 };
    if sx == 0.0 || sy == 0.0 {
        0.0
    } else {
        cov / sx / sy
    }
}

fn covar_samp(r: &tgx_result) -> f64 {
    let n = r.non_null as f64;
    if n < 2.0 {
        0.0 // SQL NULL; the reference reads the raw slot (constraints/correlation.rs:355-362)
    } else {
        r.co_c_xy / (n - 1.0)
    }
}
