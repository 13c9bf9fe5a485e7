//! `parquet2::encoding::bitpacked` (src/encoding/bitpacked/{mod,decode,encode}.rs): `Decoder<u32>` and
//! `Decoder<u64>` over `pq2_bitunpack_u32_host` / `pq2_bitunpack_u64_host`, `encode` over `pq2_bitpacked_encode_u32`.
use crate::error::Error;
use pq2hip_sys as sys;

/// The element types the device decodes (the reference also has u8 and u16, which nothing on the path uses).
pub trait Unpackable: Copy + Default {
    #[doc(hidden)]
    unsafe fn unpack(ctx: *mut sys::pq2_ctx, packed: &[u8], num_bits: usize, out: &mut [Self]) -> i32;
}
impl Unpackable for u32 {
    unsafe fn unpack(ctx: *mut sys::pq2_ctx, packed: &[u8], num_bits: usize, out: &mut [Self]) -> i32 {
        sys::pq2_bitunpack_u32_host(ctx, packed.as_ptr() as _, packed.len() as u64, num_bits as u32, out.len() as u64, out.as_mut_ptr())
    }
}
impl Unpackable for u64 {
    unsafe fn unpack(ctx: *mut sys::pq2_ctx, packed: &[u8], num_bits: usize, out: &mut [Self]) -> i32 {
        sys::pq2_bitunpack_u64_host(ctx, packed.as_ptr() as _, packed.len() as u64, num_bits as u32, out.len() as u64, out.as_mut_ptr())
    }
}

/// src/encoding/bitpacked/decode.rs:9-86
#[derive(Debug, Clone)]
pub struct Decoder<'a, T: Unpackable> {
    values: std::vec::IntoIter<T>,
    _borrow: std::marker::PhantomData<&'a [u8]>,
}

impl<'a, T: Unpackable> Decoder<'a, T> {
    /// Errors as the reference's, in its order: `num_bits == 0`, then too few bytes for `length` values.
    pub fn try_new(packed: &'a [u8], num_bits: usize, length: usize) -> Result<Self, Error> {
        let mut out = vec![T::default(); length];
        crate::with_ctx(|ctx| {
            let rc = unsafe { T::unpack(ctx.as_ptr(), packed, num_bits, &mut out) };
            if rc != sys::PQ2_OK {
                return Err(crate::error::from_status(rc, ctx.as_ptr()));
            }
            Ok(())
        })?;
        // (an empty `packed` yields nothing, as decode.rs:46-50 has it)
        if packed.is_empty() {
            out.clear();
        }
        Ok(Self { values: out.into_iter(), _borrow: Default::default() })
    }
}
impl<'a, T: Unpackable> Iterator for Decoder<'a, T> {
    type Item = T;
    fn next(&mut self) -> Option<T> {
        self.values.next()
    }
    fn size_hint(&self) -> (usize, Option<usize>) {
        self.values.size_hint()
    }
}

/// src/encoding/bitpacked/encode.rs:11-54 (u32): `packed` must hold `ceil8(unpacked.len() * num_bits)` bytes.
pub fn encode(unpacked: &[u32], num_bits: usize, packed: &mut [u8]) {
    let rc = unsafe {
        sys::pq2_bitpacked_encode_u32(unpacked.as_ptr(), unpacked.len() as u64, num_bits as u32, packed.as_mut_ptr(), packed.len() as u64)
    };
    assert_eq!(rc, sys::PQ2_OK, "packed is too short");
}
