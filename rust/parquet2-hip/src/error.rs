//! `parquet2::error::Error` (src/error.rs:23-34) and the mapping of the C ABI's statuses onto it.
use pq2hip_sys as sys;

#[derive(Debug, Clone, PartialEq, Eq)]
pub enum Error {
    OutOfSpec(String),
    FeatureNotActive(String),
    FeatureNotSupported(String),
    InvalidParameter(String),
    WouldOverAllocate,
}

pub(crate) fn from_status(status: i32, ctx: *const sys::pq2_ctx) -> Error {
    let text = |p: *const std::os::raw::c_char| {
        if p.is_null() { String::new() } else { unsafe { std::ffi::CStr::from_ptr(p) }.to_string_lossy().into_owned() }
    };
    let mut msg = if ctx.is_null() { String::new() } else { text(unsafe { sys::pq2_last_error(ctx) }) };
    if msg.is_empty() {
        msg = text(unsafe { sys::pq2_strerror(status) });
    }
    match status {
        sys::PQ2_OUT_OF_SPEC => Error::OutOfSpec(msg),
        sys::PQ2_INVALID_PARAMETER => Error::InvalidParameter(msg),
        sys::PQ2_FEATURE_NOT_SUPPORTED => Error::FeatureNotSupported(msg),
        sys::PQ2_WOULD_OVER_ALLOCATE => Error::WouldOverAllocate,
        // the reference panics on these inputs; the drop-in reports them
        sys::PQ2_REFERENCE_WOULD_PANIC => Error::OutOfSpec(format!("malformed stream (parquet2 would panic): {msg}")),
        _ => Error::OutOfSpec(format!("device error: {msg}")),
    }
}
