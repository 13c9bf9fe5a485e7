//! src/encoding/zigzag_leb128.rs:5-15
use crate::error::Error;
use pq2hip_sys as sys;

pub fn decode(values: &[u8]) -> Result<(i64, usize), Error> {
    let (mut value, mut consumed) = (0i64, 0u64);
    let rc = unsafe { sys::pq2_zigzag_leb128_decode(values.as_ptr(), values.len() as u64, &mut value, &mut consumed) };
    if rc != sys::PQ2_OK {
        return Err(crate::error::from_status(rc, std::ptr::null()));
    }
    Ok((value, consumed as usize))
}

pub fn encode(value: i64) -> ([u8; 10], usize) {
    let mut container = [0u8; 10];
    let n = unsafe { sys::pq2_zigzag_leb128_encode(value, container.as_mut_ptr()) } as usize;
    (container, n)
}
