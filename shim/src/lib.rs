//! tgx: term-guard's check evaluator on an MI355X.
//!
//! * [`sys`] -- the C ABI of `libtgx.so`, generated from `include/tgx.h`;
//! * [`handles`] -- `Plan` / `State` / `Comm` as owning Rust types, statuses as `Result`;
//! * [`column`] -- an Arrow array as the `tgx_column` view the kernels read in place;
//! * [`planner`] (feature `term-guard`) -- constraints that answer from ONE fused pass over the table, for use in an
//!   ordinary `ValidationSuite` (`term-guard/src/core/suite.rs:399`).
//!
//! ```ignore
//! let gpu = tgx::planner::GpuPlanner::new();
//! let suite = ValidationSuite::builder("orders")
//!     .check(Check::builder("keys").level(Level::Error)
//!         .constraint(gpu.completeness(["o_orderkey"], LogicalOperator::All, 1.0)?)
//!         .constraint(gpu.uniqueness(["o_orderkey"], UniquenessType::PrimaryKey)?)
//!         .constraint(gpu.statistic("o_totalprice", StatisticType::Min, Assertion::GreaterThanOrEqual(0.0))?)
//!         .build())
//!     .build();
//! let report = gpu.run(&suite, &ctx).await?;      // one scan of the table, then the suite's own tally
//! ```
pub mod column;
pub mod handles;
pub mod patterns;
#[cfg(feature = "term-guard")]
pub mod planner;
pub mod sys;

pub use handles::{init, Comm, Error, Plan, Spec, State};
