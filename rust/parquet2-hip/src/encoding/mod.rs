//! `parquet2::encoding` (src/encoding/mod.rs): the root helpers and the sub-modules on the path.
pub mod bitpacked;
pub mod delta_bitpacked;
pub mod hybrid_rle;
pub mod uleb128;
pub mod zigzag_leb128;

use pq2hip_sys as sys;

/// src/encoding/mod.rs:14-21
pub fn get_length(values: &[u8]) -> Option<usize> {
    let mut out = 0u64;
    let rc = unsafe { sys::pq2_get_length(values.as_ptr(), values.len() as u64, &mut out) };
    (rc == sys::PQ2_OK).then_some(out as usize)
}

/// src/encoding/mod.rs:23-27
pub fn ceil8(value: usize) -> usize {
    unsafe { sys::pq2_ceil8(value as u64) as usize }
}
