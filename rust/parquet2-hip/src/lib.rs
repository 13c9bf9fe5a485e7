//! The public names of parquet2's hybrid-RLE / bit-unpack path (`src/encoding/{hybrid_rle,bitpacked,uleb128}`,
//! `src/deserialize/utils.rs`, `src/page/mod.rs:350-431`, `src/read/levels.rs`) with the decode done by
//! `libpq2hip.so` on an MI355X.  NOT COMPILED in the build image (no rustc); see ../README.md.
//!
//! The iterators decode eagerly in `try_new` (one call through the C ABI) and then hand out what they hold:
//! an error that the reference would return from a later `next()` comes from `try_new`.
pub mod deserialize;
pub mod encoding;
pub mod error;
pub mod page;
pub mod read;

use pq2hip_sys as sys;
use std::cell::RefCell;

/// One `pq2_ctx` per (thread, GPU): a context is driven by one thread at a time.
pub struct Context(pub(crate) *mut sys::pq2_ctx);

impl Context {
    pub fn new(device_id: i32) -> Result<Self, error::Error> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { sys::pq2_ctx_create(device_id, &mut ctx) };
        if rc != sys::PQ2_OK {
            return Err(error::from_status(rc, std::ptr::null()));
        }
        Ok(Self(ctx))
    }
    pub fn as_ptr(&self) -> *mut sys::pq2_ctx {
        self.0
    }
    /// `pq2_decode_batch_host`: many pages, one call (what a reader of column chunks uses instead of per-page
    /// `try_new`).  Returns the per-page statuses.
    pub fn decode_batch_host(&self, input: &[u8], descs: &[sys::pq2_page_desc], out: &mut [u32]) -> Result<Vec<i32>, error::Error> {
        let mut status = vec![0i32; descs.len()];
        let rc = unsafe {
            sys::pq2_decode_batch_host(self.0, input.as_ptr() as _, input.len() as u64, descs.as_ptr(), descs.len() as u64,
                                       out.as_mut_ptr(), out.len() as u64, status.as_mut_ptr())
        };
        if rc != sys::PQ2_OK {
            return Err(error::from_status(rc, self.0));
        }
        Ok(status)
    }
    /// `pq2_partition_pages`: contiguous page ranges for `n_parts` GPUs, balanced on the bytes a page moves.
    pub fn partition_pages(descs: &[sys::pq2_page_desc], n_parts: u32) -> Vec<u64> {
        let mut bounds = vec![0u64; n_parts as usize + 1];
        unsafe { sys::pq2_partition_pages(descs.as_ptr(), descs.len() as u64, n_parts, bounds.as_mut_ptr()) };
        bounds
    }
    /// `pq2_page_costs` + `pq2_partition_by_cost`: the same with the streams at hand (`input`, host memory), balanced
    /// on bytes and on the run headers the pages are estimated to hold.
    pub fn partition_pages_of(input: &[u8], descs: &[sys::pq2_page_desc], n_parts: u32) -> Vec<u64> {
        let mut costs = vec![0u64; descs.len()];
        let mut bounds = vec![0u64; n_parts as usize + 1];
        unsafe {
            sys::pq2_page_costs(input.as_ptr() as *const _, input.len() as u64, descs.as_ptr(), descs.len() as u64,
                                costs.as_mut_ptr());
            sys::pq2_partition_by_cost(costs.as_ptr(), descs.len() as u64, n_parts, bounds.as_mut_ptr());
        }
        bounds
    }
}
impl Drop for Context {
    fn drop(&mut self) {
        unsafe { sys::pq2_ctx_destroy(self.0) };
    }
}

thread_local! {
    static CTX: RefCell<Option<Context>> = RefCell::new(None);
}
/// The calling thread's context on the GPU that `PQ2_DEVICE` names (default 0), created at first use.
pub(crate) fn with_ctx<T>(f: impl FnOnce(&Context) -> Result<T, error::Error>) -> Result<T, error::Error> {
    CTX.with(|slot| {
        let mut slot = slot.borrow_mut();
        if slot.is_none() {
            let device = std::env::var("PQ2_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
            *slot = Some(Context::new(device)?);
        }
        f(slot.as_ref().unwrap())
    })
}
