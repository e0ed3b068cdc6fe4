//! term-guard's CPU path on the synthetic table of bench.py: the 16-column
//! "null + range + unique" suite -- completeness x16, has_min / has_max / has_mean x16, validates_uniqueness x2 --
//! run through `ValidationSuite::run` (term-guard/src/core/suite.rs:399) over a DataFusion MemTable whose batches
//! are spread over `target_partitions = nproc` partitions, like term-guard/benches/comprehensive_benchmarks.rs:49-104
//! builds its table.  Prints one JSON line: rows, cores, seconds, rows_per_s, and the check tally.
//!
//!     term_guard_cpu <rows> [seed] [batch_rows]
//!
//! The table is the one term_amd/synth.py generates on the device (COLUMNS_16): value(col, row) =
//! f(mix64(seed ^ (col + 1) * PHI ^ row)); the same seed gives the same table, so the tallies can be compared with
//! the GPU run's.  NOT compiled in the builder's image (no cargo, no vendored crates): see Cargo.toml.
use std::sync::Arc;
use std::time::Instant;

use arrow::array::{ArrayRef, Float64Array, Int64Array};
use arrow::datatypes::{DataType, Field, Schema};
use arrow::record_batch::RecordBatch;
use datafusion::datasource::MemTable;
use datafusion::prelude::{SessionConfig, SessionContext};
use term_guard::constraints::Assertion;
use term_guard::core::{Check, ConstraintOptions, Level, ValidationSuite};

const PHI: u64 = 0x9E37_79B9_7F4A_7C15;
const NULL_RATE: f64 = 0.05;

#[derive(Clone, Copy, PartialEq)]
enum Kind {
    IdPerm,
    KMod10,
    IWide,
    ISmall,
    FUniform,
    FNormal,
    FExpo,
}

// term_amd/synth.py COLUMNS_16: (kind, has_validity)
const COLUMNS: [(Kind, bool); 16] = [
    (Kind::IdPerm, false),
    (Kind::KMod10, true),
    (Kind::IWide, true),
    (Kind::IWide, true),
    (Kind::IWide, true),
    (Kind::IWide, true),
    (Kind::ISmall, true),
    (Kind::IWide, false),
    (Kind::FUniform, true),
    (Kind::FNormal, true),
    (Kind::FUniform, true),
    (Kind::FExpo, true),
    (Kind::FUniform, true),
    (Kind::FNormal, true),
    (Kind::FUniform, false),
    (Kind::FNormal, false),
];

fn mix64(mut x: u64) -> u64 {
    x ^= x >> 30;
    x = x.wrapping_mul(0xBF58_476D_1CE4_E5B9);
    x ^= x >> 27;
    x = x.wrapping_mul(0x94D0_49BB_1331_11EB);
    x ^ (x >> 31)
}

fn gcd(a: u64, b: u64) -> u64 {
    if b == 0 {
        a
    } else {
        gcd(b, a % b)
    }
}

fn perm_multiplier(n_total: u64) -> u64 {
    let mut a = 6_364_136_223u64;
    while gcd(a, n_total) != 1 {
        a += 2;
    }
    a
}

fn is_valid(col: usize, row: u64, seed: u64) -> bool {
    let salt = seed ^ ((col as u64 + 101).wrapping_mul(PHI));
    let thresh = (NULL_RATE * (1u64 << 53) as f64) as u64;
    (mix64(row ^ salt) >> 11) >= thresh
}

fn column(col: usize, kind: Kind, has_validity: bool, row0: u64, n: usize, n_total: u64, seed: u64) -> ArrayRef {
    let salt = seed ^ ((col as u64 + 1).wrapping_mul(PHI));
    let a = perm_multiplier(n_total);
    let valid = |row: u64| !has_validity || is_valid(col, row, seed);
    let unit = |h: u64| (h >> 11) as f64 * (1.0 / (1u64 << 53) as f64);
    match kind {
        Kind::IdPerm | Kind::KMod10 | Kind::IWide | Kind::ISmall => {
            let it = (0..n as u64).map(|i| {
                let row = row0 + i;
                if !valid(row) {
                    return None;
                }
                let h = mix64(row ^ salt);
                Some(match kind {
                    Kind::IdPerm => ((row.wrapping_mul(a).wrapping_add(12345)) % n_total) as i64,
                    Kind::KMod10 => ((h >> 1) % std::cmp::max(1, n_total / 10)) as i64,
                    Kind::IWide => (h as i64) >> 23,
                    _ => ((h >> 1) % 1000) as i64 - 500,
                })
            });
            Arc::new(Int64Array::from_iter(it))
        }
        _ => {
            let it = (0..n as u64).map(|i| {
                let row = row0 + i;
                if !valid(row) {
                    return None;
                }
                let h = mix64(row ^ salt);
                let u = unit(h);
                Some(match kind {
                    Kind::FUniform => u * 1000.0,
                    Kind::FExpo => -(-u).ln_1p() * 50.0,
                    _ => {
                        let u2 = unit(mix64(h ^ 0xD1B5_4A32_D192_ED03));
                        (-2.0 * (-u).ln_1p()).sqrt() * (6.283185307179586 * u2).cos()
                    }
                })
            });
            Arc::new(Float64Array::from_iter(it))
        }
    }
}

fn build_suite() -> ValidationSuite {
    // one check per column, as a term-guard user writes it; every constraint is one SQL query over the table
    // (core/suite.rs:67-100): 16 x 4 + 2 = 66 scans
    let mut suite = ValidationSuite::builder("null+range+unique x16");
    for c in 0..COLUMNS.len() {
        let name = format!("c{c}");
        let mut check = Check::builder(format!("col_{c}"))
            .level(Level::Error)
            .completeness(name.as_str(), ConstraintOptions::new().with_threshold(0.9))
            .has_min(name.as_str(), Assertion::GreaterThan(f64::MIN))
            .has_max(name.as_str(), Assertion::LessThan(f64::MAX))
            .has_mean(name.as_str(), Assertion::Between(-1e18, 1e18));
        if c < 2 {
            // the two key columns: FullUniqueness (constraints/uniqueness.rs:612-617); c1 has ~10 rows per key
            check = check.validates_uniqueness(vec![name.as_str()], if c == 0 { 1.0 } else { 0.0 });
        }
        suite = suite.check(check.build());
    }
    suite.build()
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    let rows: u64 = args.get(1).and_then(|s| s.parse().ok()).unwrap_or(1 << 24);
    let seed: u64 = args.get(2).and_then(|s| s.parse().ok()).unwrap_or(0x7E57_0004);
    let batch_rows: usize = args.get(3).and_then(|s| s.parse().ok()).unwrap_or(8192 * 64);
    let cores = num_cpus::get();

    let fields: Vec<Field> = (0..COLUMNS.len())
        .map(|c| {
            let float = matches!(COLUMNS[c].0, Kind::FUniform | Kind::FNormal | Kind::FExpo);
            Field::new(format!("c{c}"), if float { DataType::Float64 } else { DataType::Int64 }, COLUMNS[c].1)
        })
        .collect();
    let schema = Arc::new(Schema::new(fields));
    // batches dealt round-robin over `cores` partitions: DataFusion scans the partitions of a MemTable in parallel
    let mut partitions: Vec<Vec<RecordBatch>> = vec![Vec::new(); cores];
    let mut row0 = 0u64;
    let mut k = 0usize;
    while row0 < rows {
        let n = std::cmp::min(batch_rows as u64, rows - row0) as usize;
        let cols: Vec<ArrayRef> = (0..COLUMNS.len())
            .map(|c| column(c, COLUMNS[c].0, COLUMNS[c].1, row0, n, rows, seed))
            .collect();
        partitions[k % cores].push(RecordBatch::try_new(schema.clone(), cols).expect("batch"));
        row0 += n as u64;
        k += 1;
    }
    let config = SessionConfig::new().with_target_partitions(cores).with_batch_size(8192); // core/context.rs:28-38
    let ctx = SessionContext::new_with_config(config);
    let table = MemTable::try_new(schema, partitions).expect("table");
    ctx.register_table("data", Arc::new(table)).expect("register");

    let suite = build_suite();
    let rt = tokio::runtime::Builder::new_multi_thread().enable_all().build().expect("runtime");
    // one untimed pass (plans, allocator), then the timed one
    let _ = rt.block_on(suite.run(&ctx)).expect("warm-up run");
    let t0 = Instant::now();
    let result = rt.block_on(suite.run(&ctx)).expect("run");
    let secs = t0.elapsed().as_secs_f64();
    let m = &result.report().metrics;
    println!(
        "{{\"rows\": {rows}, \"cores\": {cores}, \"seconds\": {secs:.6}, \"rows_per_s\": {:.1}, \"total_checks\": {}, \
         \"passed_checks\": {}, \"failed_checks\": {}, \"success\": {}}}",
        rows as f64 / secs,
        m.total_checks,
        m.passed_checks,
        m.failed_checks,
        result.is_success()
    );
}
