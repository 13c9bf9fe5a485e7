//! src/encoding/uleb128.rs:3-46
use crate::error::Error;
use pq2hip_sys as sys;

pub fn decode(values: &[u8]) -> Result<(u64, usize), Error> {
    let (mut value, mut consumed) = (0u64, 0u64);
    let rc = unsafe { sys::pq2_uleb128_decode(values.as_ptr(), values.len() as u64, &mut value, &mut consumed) };
    if rc != sys::PQ2_OK {
        return Err(crate::error::from_status(rc, std::ptr::null()));
    }
    Ok((value, consumed as usize))
}

pub fn encode(value: u64, container: &mut [u8; 10]) -> usize {
    unsafe { sys::pq2_uleb128_encode(value, container.as_mut_ptr()) as usize }
}
