//! Owning handles over the C ABI.  Nothing here panics across the boundary and the library never unwinds into Rust:
//! every status comes back as `Err(Error)`.
use crate::sys::*;
use std::ffi::CStr;
use std::os::raw::c_char;
use std::ptr;

/// A non-zero `tgx_status` with the library's message.
#[derive(Debug, Clone)]
pub struct Error {
    pub status: i32,
    pub message: String,
}
impl Error {
    /// `TGX_UNSUPPORTED`: the type / shape / pattern lies outside the path -- fall back to the stock SQL constraint.
    pub fn is_unsupported(&self) -> bool {
        self.status == TGX_UNSUPPORTED
    }
    pub fn status_name(&self) -> String {
        unsafe { CStr::from_ptr(tgx_status_name(self.status)) }.to_string_lossy().into_owned()
    }
}
impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "{}: {}", self.status_name(), self.message)
    }
}
impl std::error::Error for Error {}

fn new_err() -> tgx_error {
    tgx_error { code: 0, msg: [0 as c_char; 256] }
}
fn check(status: i32, err: &tgx_error) -> Result<(), Error> {
    if status == TGX_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(err.msg.as_ptr()) }.to_string_lossy().into_owned();
    Err(Error { status, message })
}

/// Once per process, before any other call: selects the device (`-1` = the current HIP device) and checks that it
/// is a gfx950, and that the header these bindings were generated from is the library's.
pub fn init(device_id: i32, coalesce_small_batches: bool) -> Result<(), Error> {
    let abi = unsafe { tgx_abi_version() };
    if abi != TGX_ABI_VERSION {
        return Err(Error { status: TGX_INTERNAL, message: format!("libtgx has ABI {abi}, these bindings {TGX_ABI_VERSION}") });
    }
    let opts = tgx_options {
        device_id,
        flags: if coalesce_small_batches { 0 } else { TGX_OPT_NO_COALESCE },
        distinct_capacity_hint: 0,
    };
    let mut err = new_err();
    check(unsafe { tgx_init(&opts, &mut err) }, &err)
}

/// One aggregate of the fused plan (a `tgx_check_spec` that owns its pattern and column list).
#[derive(Debug, Clone, Default, PartialEq)]
pub struct Spec {
    pub kind: i32,
    pub column: i32,
    pub column2: i32,
    pub flags: u32,
    pub pattern: Vec<u8>,
    pub kll_k: u32,
    pub columns: Vec<i32>,
    pub length_min: u64,
    pub length_max: u64,
}

/// The fused set of aggregates of a suite: every column buffer is read once.  Immutable, shareable between threads.
pub struct Plan {
    raw: *mut tgx_plan,
    n_specs: usize,
}
unsafe impl Send for Plan {}
unsafe impl Sync for Plan {}
impl Plan {
    pub fn new(specs: &[Spec]) -> Result<Plan, Error> {
        let raw_specs: Vec<tgx_check_spec> = specs
            .iter()
            .map(|s| tgx_check_spec {
                kind: s.kind,
                column: s.column,
                column2: s.column2,
                flags: s.flags,
                pattern: if s.pattern.is_empty() { ptr::null() } else { s.pattern.as_ptr() as *const c_char },
                pattern_len: s.pattern.len() as u64,
                kll_k: s.kll_k,
                reserved: 0,
                columns: if s.columns.is_empty() { ptr::null() } else { s.columns.as_ptr() },
                n_columns: s.columns.len() as u32,
                reserved2: 0,
                length_min: s.length_min,
                length_max: s.length_max,
            })
            .collect();
        let mut raw = ptr::null_mut();
        let mut err = new_err();
        check(unsafe { tgx_plan_create(raw_specs.as_ptr(), raw_specs.len(), &mut raw, &mut err) }, &err)?;
        Ok(Plan { raw, n_specs: specs.len() })
    }
    pub fn num_specs(&self) -> usize {
        self.n_specs
    }
    pub(crate) fn raw(&self) -> *const tgx_plan {
        self.raw
    }
}
impl Drop for Plan {
    fn drop(&mut self) {
        unsafe { tgx_plan_destroy(self.raw) }
    }
}

/// The partial state of every aggregate of a plan (`AnalyzerState`, term-guard/src/analyzers/traits.rs:154-179).
/// One per concurrent run / partition stream; callable from any thread, one thread at a time.
pub struct State<'p> {
    plan: &'p Plan,
    raw: *mut tgx_state,
}
unsafe impl<'p> Send for State<'p> {}
impl<'p> State<'p> {
    pub fn new(plan: &'p Plan) -> Result<State<'p>, Error> {
        let mut raw = ptr::null_mut();
        let mut err = new_err();
        check(unsafe { tgx_state_create(plan.raw(), ptr::null_mut(), &mut raw, &mut err) }, &err)?;
        Ok(State { plan, raw })
    }
    /// One call per RecordBatch (`Analyzer::compute_state_from_data`).  The views are borrowed until the call returns
    /// (HOST memory; DEVICE buffers stay alive and unmodified until `finalize` / `sync`, include/tgx.h).
    pub fn update(&mut self, columns: &[tgx_column]) -> Result<(), Error> {
        let mut err = new_err();
        check(unsafe { tgx_update(self.plan.raw(), self.raw, columns.as_ptr(), columns.len(), &mut err) }, &err)
    }
    /// How many of the batches fed so far are only noted (coalesced, not yet run): the last `.0` ones (`.1` rows).
    /// Whoever feeds `TGX_MEM_HOST_RETAINED` buffers keeps exactly those alive.
    pub fn pending(&self) -> (u64, u64) {
        let (mut b, mut r) = (0u64, 0u64);
        unsafe { tgx_state_pending(self.raw, &mut b, &mut r) };
        (b, r)
    }
    /// `AnalyzerState::merge` (traits.rs:160-170): exact, DISTINCT included (set union).
    pub fn merge(&mut self, others: &mut [State<'p>]) -> Result<(), Error> {
        let raws: Vec<*mut tgx_state> = others.iter().map(|s| s.raw).collect();
        let mut err = new_err();
        check(unsafe { tgx_merge(self.plan.raw(), self.raw, raws.as_ptr(), raws.len(), &mut err) }, &err)
    }
    /// The same merge across the row shards of several GPUs (one process per GPU): afterwards `finalize` returns the
    /// whole table's results on every rank.
    pub fn allreduce(&mut self, comm: &mut Comm) -> Result<(), Error> {
        let mut err = new_err();
        check(unsafe { tgx_allreduce(self.plan.raw(), self.raw, comm.raw, &mut err) }, &err)
    }
    /// One result per spec (`Analyzer::compute_metric_from_state` inputs).
    pub fn finalize(&mut self) -> Result<Vec<tgx_result>, Error> {
        let mut out: Vec<tgx_result> = vec![unsafe { std::mem::zeroed() }; self.plan.num_specs()];
        let mut err = new_err();
        check(unsafe { tgx_finalize(self.plan.raw(), self.raw, out.as_mut_ptr(), out.len(), &mut err) }, &err)?;
        Ok(out)
    }
    /// `KllSketch::get_quantile` (analyzers/advanced/kll_sketch.rs:246-322) of a KLL spec.
    pub fn kll_quantile(&mut self, spec_index: usize, phi: f64) -> Result<f64, Error> {
        let mut v = 0.0f64;
        let mut err = new_err();
        check(unsafe { tgx_kll_quantile(self.plan.raw(), self.raw, spec_index, phi, &mut v, &mut err) }, &err)?;
        Ok(v)
    }
    pub fn reset(&mut self) -> Result<(), Error> {
        let mut err = new_err();
        check(unsafe { tgx_state_reset(self.plan.raw(), self.raw, &mut err) }, &err)
    }
    /// The bytes an `AnalyzerState` implementation can persist (analyzers/incremental/runner.rs:71-111).
    pub fn serialize(&mut self) -> Result<Vec<u8>, Error> {
        let (mut len, mut err) = (0usize, new_err());
        check(unsafe { tgx_state_serialize(self.plan.raw(), self.raw, ptr::null_mut(), 0, &mut len, &mut err) }, &err)?;
        let mut buf = vec![0u8; len];
        check(unsafe { tgx_state_serialize(self.plan.raw(), self.raw, buf.as_mut_ptr(), buf.len(), &mut len, &mut err) }, &err)?;
        buf.truncate(len);
        Ok(buf)
    }
    pub fn deserialize(plan: &'p Plan, bytes: &[u8]) -> Result<State<'p>, Error> {
        let mut raw = ptr::null_mut();
        let mut err = new_err();
        check(unsafe { tgx_state_deserialize(plan.raw(), bytes.as_ptr(), bytes.len(), &mut raw, &mut err) }, &err)?;
        Ok(State { plan, raw })
    }
}
impl<'p> Drop for State<'p> {
    fn drop(&mut self) {
        unsafe { tgx_state_destroy(self.raw) }
    }
}

/// The transport of the cross-rank step: RCCL over xGMI (`ncclCommInitRank` inside the library).
pub struct Comm {
    raw: *mut tgx_comm,
}
unsafe impl Send for Comm {}
impl Comm {
    /// Rank 0 draws the id and ships its 128 bytes to the other ranks over the host's own control plane.
    pub fn rccl_unique_id() -> Result<[u8; TGX_RCCL_UNIQUE_ID_BYTES as usize], Error> {
        let mut id = [0u8; TGX_RCCL_UNIQUE_ID_BYTES as usize];
        let mut err = new_err();
        check(unsafe { tgx_comm_rccl_unique_id(id.as_mut_ptr(), &mut err) }, &err)?;
        Ok(id)
    }
    pub fn rccl(id: &[u8; TGX_RCCL_UNIQUE_ID_BYTES as usize], rank: i32, world: i32) -> Result<Comm, Error> {
        let mut raw = ptr::null_mut();
        let mut err = new_err();
        check(unsafe { tgx_comm_create_rccl(id.as_ptr(), rank, world, &mut raw, &mut err) }, &err)?;
        Ok(Comm { raw })
    }
    /// Any transport of the caller's: three collectives (`tgx_comm_ops`).
    pub fn custom(ops: &tgx_comm_ops) -> Result<Comm, Error> {
        let mut raw = ptr::null_mut();
        let mut err = new_err();
        check(unsafe { tgx_comm_create(ops, &mut raw, &mut err) }, &err)?;
        Ok(Comm { raw })
    }
}
impl Drop for Comm {
    fn drop(&mut self) {
        unsafe { tgx_comm_destroy(self.raw) }
    }
}
